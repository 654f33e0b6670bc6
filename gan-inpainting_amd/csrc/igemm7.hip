// fp16 implicit GEMM for the small-M layers (the bottleneck of the U-Net: 16x16 maps and below at the headline batch,
// everything in small test nets): few output tiles, long K (4..16 taps x 512..1024 channels), 8..17 MB of weights per
// layer. Two things bound such a layer: how deep each workgroup's K pipeline is (a 2-stage register-staged loop pays the
// full global-load latency per K tile: 1.4 us per tile measured on igemm.hip) and how many launches it takes.
//
//   * 128 x BN output tile (BN = 128 or 64), 4 waves (2 x 2; wave tile 64 x BN/2), K tile = 64 halves;
//   * FOUR-stage LDS ring filled by LDS-DMA (global_load_lds_dwordx4), counted vmcnt, one raw s_barrier per K tile: two
//     tiles are in flight while one is multiplied (48..64 KB per CU outstanding), the DMA instructions of tile t+3 are
//     issued between the MFMAs of tile t (same LDS image, swizzle and gather scheme as igemm3.hip);
//   * split-K over the workgroups of a tile; every split stores its fp32 tile in thread-linear order with the sc1 bit
//     (write-through to the level all XCDs share), waits for the stores, takes a ticket; the LAST arriver adds the splits
//     up in split order (sc1 loads: miss-always) - a fixed order whoever is last - and runs the epilogue (bias,
//     activation, BatchNorm column statistics, 16-byte coalesced rows). No finish launch, no agent-scope fence (a fence is
//     buffer_wbl2 + buffer_inv of the whole L2 under the other workgroups' weight streams: measured 1.6x slower).
#include <stdlib.h>

#include "common.h"
#include "stat_acc.h"
#include "bn_acc.h"

namespace {

struct KP7 {
  const char* in;
  const char* w;
  char* out;
  const char* zero;   // >= 4 KiB of zeros (padding taps)
  const float* bias;
  float* partials;
  unsigned long long* stat_acc; int stat_pg, stat_reps;
  float* ws;          // split-K tiles: [split][tile][BM * BN] fp32
  unsigned* tickets;
  int M, Hs, Ws;
  int cin, ldin, coffin;
  int cout, ldout, coffout;
  int Ktot, nk, splitk, kt_per_split;
  int relu_in, act_out, relu_cend;
  int Hin, Win, Hout, Wout;
  int mtiles, ntiles;
  int dbg;            // ablation build only (GI_IGEMM7_DBG): 1 no MFMA, 2 no LDS-DMA (timing experiments, wrong results)
  // folded normalisation (IgemmFold, common.h): fold != 0 -> the last finisher of a channel column normalises the column
  int fold;
  unsigned* col_tickets;             // [ntiles], zero before the launch and left zero
  BnAccP fa;
  char* fdst; int flddst, fcoffdst, fact;
  uint8_t* fdrop; float fdrop_scale; uint64_t fdrop_seed; uint32_t fdrop_thresh;
};

// max(v, floor) on the halves as 16-bit integers: floor = 0 is the ReLU (negative halves, sign bit set, are negative integers), floor =
// -32768 leaves v as it is. A wave-uniform floor instead of a wave-uniform branch around the ReLU: a branch between the fragment reads
// and the MFMAs makes hipcc wait for ALL outstanding LDS reads at the join (`s_waitcnt lgkmcnt(0)`), i.e. also for the second K
// half's reads that are meant to run under the first half's MFMAs.
__device__ __forceinline__ h8_t relu7(h8_t v, short floor) {
  typedef short s8_t __attribute__((ext_vector_type(8)));
  s8_t h = __builtin_bit_cast(s8_t, v);
  const s8_t z = {floor, floor, floor, floor, floor, floor, floor, floor};
  h = __builtin_elementwise_max(h, z);
  return __builtin_bit_cast(h8_t, h);
}
__device__ __forceinline__ void glds16_7(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base,
                                   16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm7() {   // counted wait: all but the N youngest LDS-DMA pieces have landed
  static_assert(N >= 0 && N <= 63, "unexpected DMA count");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else static_assert(N < 0, "add the immediate");
}

// NSTG: ring stages; PF: split-K tiles requested together in the last arriver's tail. Shipped: GI7_NSTG / GI7_PF below; the
// ablation build (build.sh -DGI_ABLATION) instantiates the alternatives, selected by GI_IGEMM7_NSTG / GI_IGEMM7_PF.
// NW: waves per workgroup. 4 (2 x 2 waves, 64 x BN/2 each): ONE wave per SIMD - its LDS fragment reads (512 cycles per K tile at the
// LDS' 128 bytes per clock), its 32 MFMAs (512) and the issue of its 8 LDS-DMA pieces (~460) run one after the other, nothing else
// is resident to fill the gaps: 0.83 us per K tile on d5, of which 0.45 remain with the MFMAs and the DMA switched off (ablation,
// DESIGN.md 4.1j). 8 (2 x 4 waves, 64 x BN/4 each): two waves per SIMD, one's reads and DMA issue under the other's MFMAs; half
// the DMA pieces and MFMAs per wave, 1.5 x the fragment-read bytes. The register cap of two waves per SIMD (256) also keeps hipcc
// from parking accumulators in AGPRs (with 512 registers allowed it copied four of them around every MFMA).
template <int PHASE, int BN, int NSTG_, int PF_, int NW>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) igemm7_kernel(KP7 p) {
  static_assert(NW == 4 || NW == 8, "four or eight waves");
  constexpr int BM = 128, BK = 64, NTHR = NW * 64;
  constexpr int NWM = (NW == 8 && BN == 64) ? 4 : 2, NWN = NW / NWM;   // waves along M / N: wave tiles 64 x BN/2 | 64 x 32 (BN 128) | 32 x 32 (BN 64)
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;   // 32 / 24 KiB
  // ring depth (K tiles of 64: 32 KiB per stage with 128-column tiles, 24 KiB with 64-column ones)
  constexpr int NSTG = NSTG_;
  constexpr int AH = NSTG - 1;               // K tiles issued ahead of the one being multiplied
  constexpr int AJ = (BM / 8) / NW, BJ = (BN / 8) / NW;   // 8-row blocks per wave per tile: 4 + (4 | 2)
  constexpr int NPC = AJ + BJ;
  constexpr int WM = BM / NWM, WN = BN / NWN, MT = WM / 16, NT = WN / 16;
  static_assert(MT >= 1 && NT >= 1 && MT * NT >= 4, "wave tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_last;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;

  // ---- work decode: (M tile, N tile, phase, split). With >= 8 M tiles the workgroups of one XCD (bid & 7) share M tiles
  //      (gathered rows from that XCD's L2); with fewer, consecutive workgroups walk the N tiles / splits so that all XCDs work
  const int nph = PHASE == 1 ? 4 : 1;
  const int nyz = p.ntiles * nph * p.splitk;
  const int bid = blockIdx.x;
  int mt_idx, yz;
  if (p.mtiles >= 8) {
    const int xcd = bid & 7, local = bid >> 3;
    mt_idx = (local / nyz) * 8 + xcd;
    yz = local % nyz;
    if (mt_idx >= p.mtiles) return;
  } else {
    mt_idx = bid % p.mtiles;
    yz = bid / p.mtiles;
  }
  const int ks = yz % p.splitk;
  const int t2 = yz / p.splitk;
  const int nt_idx = t2 % p.ntiles;
  const int ph = t2 / p.ntiles;
  const int py = ph >> 1, px = ph & 1;
  const int m0 = mt_idx * BM, n0 = nt_idx * BN;
  const char* wptr = p.w + (PHASE == 1 ? (int64_t)ph * p.cout * p.Ktot * 2 : 0);
  const int kt0 = ks * p.kt_per_split;
  const int kt1 = min(p.nk, kt0 + p.kt_per_split);
#ifdef GI_ABLATION
  const int nk = (p.dbg & 8) ? 1 : kt1 - kt0;   // 8: one K tile only
#else
  const int nk = kt1 - kt0;
#endif

  // ---- per-lane gather rows (igemm3.hip) -----------------------------------------------------------
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ (lrow & 7);
  int abase[AJ];
  unsigned amask[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int m = m0 + (wave * AJ + j) * 8 + lrow;
    abase[j] = 0;
    amask[j] = 0;
    if (m < p.M) {
      const int x = m % p.Ws;
      const int t = m / p.Ws;
      const int y = t % p.Hs;
      const int n = t / p.Hs;
      if (PHASE) {
        const int y0 = y + py, x0 = x + px;
        abase[j] = ((n * p.Hs + y0) * p.Ws + x0) * p.ldin + p.coffin;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int iy = y0 - (tt >> 1), ix = x0 - (tt & 1);
          if (iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws) amask[j] |= 1u << tt;
        }
      } else {
        const int y0 = 2 * y - 1, x0 = 2 * x - 1;
        abase[j] = ((n * p.Hin + y0) * p.Win + x0) * p.ldin + p.coffin;
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) {
          const int iy = y0 + (tt >> 2), ix = x0 + (tt & 3);
          if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) amask[j] |= 1u << tt;
        }
      }
    }
  }
  const char* pa[AJ];
  const char* pb[BJ];
  int tap = (kt0 * BK) / p.cin, c0 = (kt0 * BK) % p.cin;
  auto set_tap = [&]() {
    int toff;
    if (PHASE) toff = -((tap >> 1) * p.Win + (tap & 1)) * p.ldin;
    else toff = ((tap >> 2) * p.Win + (tap & 3)) * p.ldin;
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      pa[j] = ((amask[j] >> tap) & 1u) ? p.in + (int64_t)(abase[j] + toff) * 2 + lchunk * 16 : p.zero + lchunk * 16;
  };
#pragma unroll
  for (int j = 0; j < BJ; ++j) pb[j] = wptr + (int64_t)(n0 + (wave * BJ + j) * 8 + lrow) * p.Ktot * 2 + lchunk * 16;
  set_tap();

  int kt_issue = kt0;
  auto issue_piece = [&](auto STG, auto PIECE) {
    constexpr int stage = decltype(STG)::value;
    constexpr int j = decltype(PIECE)::value;
#ifdef GI_ABLATION
    if (p.dbg & 2) return;
#endif
    if constexpr (j < AJ) glds16_7(pa[j] + c0 * 2, smem + stage * STAGE + wave * (AJ * 1024) + j * 1024);
    else glds16_7(pb[j - AJ] + (int64_t)kt_issue * (BK * 2), smem + stage * STAGE + A_BYTES + wave * (BJ * 1024) + (j - AJ) * 1024);
  };
  auto issue_done = [&]() {
    ++kt_issue;
    c0 += BK;
    if (c0 >= p.cin) { c0 = 0; ++tap; set_tap(); }
  };
  auto issue = [&](auto STG) {
    static_for<NPC>([&](auto J) { issue_piece(STG, J); });
    issue_done();
  };

  f4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 15, lq = lane >> 4;
  const int rdA0 = (wm * WM + lr) * 128 + (((0 + lq) ^ (lr & 7)) << 4);
  const int rdA1 = (wm * WM + lr) * 128 + (((4 + lq) ^ (lr & 7)) << 4);
  const int rdB0 = A_BYTES + (wn * WN + lr) * 128 + (((0 + lq) ^ (lr & 7)) << 4);
  const int rdB1 = A_BYTES + (wn * WN + lr) * 128 + (((4 + lq) ^ (lr & 7)) << 4);
  const int relu_cend = p.relu_in ? p.relu_cend : 0;
  int cc0 = c0;   // channel offset of the tile being computed

  // multiply the tile in stage CUR; with ISS the NPC DMA pieces of the tile three ahead go out one at a time between the MFMAs
  auto compute = [&](auto STG, auto NXT, auto ISS) {
    constexpr int stage = decltype(STG)::value;
    constexpr bool iss = decltype(ISS)::value;
    constexpr int NMF = 2 * MT * NT;
    const short rfloor = cc0 < relu_cend ? (short)0 : (short)-32768;   // wave-uniform: ReLU on / off (relu7)
    const char* s = smem + stage * STAGE;
    // both K halves' fragments are requested before the first MFMA: the second half's reads (LDS bandwidth: 32 KiB per half for the
    // workgroup) run under the first half's MFMAs. (Read per half, hipcc issued the second half's reads only after ten MFMAs of the
    // first and the wave - the only one on its SIMD - waited for LDS twice per tile: 0.45 of the 0.83 us per tile, measured with the
    // MFMAs and the DMA switched off.)
    h8_t af[2][MT], bf[2][NT];
    // fragments of the second K half: read r (r < MT: pixel fragment r, else column fragment r - MT)
    auto read1 = [&](int r) {
      if (r < MT) af[1][r] = *(const h8_t*)(s + rdA1 + r * 2048);
      else bf[1][r - MT] = *(const h8_t*)(s + rdB1 + (r - MT) * 2048);
    };
    constexpr int NRD = MT + NT, PS = NRD < MT * NT ? NRD : MT * NT;   // PS of them go behind the first MFMAs, the rest up front
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[0][mt] = *(const h8_t*)(s + rdA0 + mt * 2048);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf[0][nt] = *(const h8_t*)(s + rdB0 + nt * 2048);
#pragma unroll
    for (int r = 0; r < NRD - PS; ++r) read1(r);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[k2][mt] = relu7(af[k2][mt], rfloor);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int idx = k2 * MT * NT + mt * NT + nt;
#ifdef GI_ABLATION
          if (!(p.dbg & 1))
#endif
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[k2][nt], af[k2][mt], acc[mt][nt], 0, 0, 0);   // D^T: rows = channels
          if (k2 == 0 && idx < PS) {               // one read of the second half behind each of the first MFMAs, pinned
            read1(NRD - PS + idx);
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (iss) {
            static_for<NPC>([&](auto Q) {
              constexpr int q = decltype(Q)::value;
              if (idx == ((q + 1) * NMF) / (NPC + 1) - 1) issue_piece(NXT, Q);
            });
          }
        }
    }
    if constexpr (iss) issue_done();
    cc0 += BK;
    if (cc0 >= p.cin) cc0 = 0;
  };

  using T1 = std::integral_constant<bool, true>;
  using T0 = std::integral_constant<bool, false>;
  // ---- NSTG-stage ring: tiles t+1 .. t+AH-1 stay in flight while tile t is multiplied -----------------------------------------
  static_for<AH>([&](auto I) { if (nk > decltype(I)::value) issue(I); });
  auto wait_tile = [&](int newer) {   // `newer` tiles issued after tile t may stay in flight (newer <= AH - 1)
    static_for<AH>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if (newer == i || (i == AH - 1 && newer > i)) wait_vm7<i * NPC>();
    });
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // tile t is multiplied from stage t % NSTG while tile t + AH is issued into the stage tile t - 1 was read from (free once every
  // wave has passed the barrier of step t); the steady state is unrolled over the ring so that stage indices are compile-time
  int t = 0;
  for (; t + NSTG - 1 < nk - AH; t += NSTG)
    static_for<NSTG>([&](auto S) {
      constexpr int st = decltype(S)::value;
      wait_tile(AH - 1);
      compute(S, std::integral_constant<int, (st + AH) % NSTG>{}, T1{});
    });
  for (; t < nk; ++t) {
    wait_tile(min(AH - 1, nk - 1 - t));
    const bool is = t + AH < nk;
    const int cur = t % NSTG;
    static_for<NSTG>([&](auto S) {
      constexpr int st = decltype(S)::value;
      if (cur == st) {
        if (is) compute(S, std::integral_constant<int, (st + AH) % NSTG>{}, T1{});
        else compute(S, std::integral_constant<int, (st + AH) % NSTG>{}, T0{});
      }
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- split-K fix-up by the last arriver of the tile ---------------------------------------------------------------------
  // Hand-off form: guides/MI355X_MICROARCH.md, "Valid forms", first row of the table of hand-offs measured with sc1 loads in
  // place of the acquire: EVERY payload store is a 16-byte sc1 (write-through) buffer store, every storing wave drains them
  // (s_waitcnt vmcnt(0)), a workgroup barrier, then ONE lane's agent-scope atomic add on the tile's ticket; the workgroup
  // whose add returned splitk - 1 is the consumer: its other waves pass a workgroup barrier the adding lane then joins, and
  // EVERY load of the handed-off bytes is a 16-byte sc1 buffer load to registers (one workgroup per CU: 96 - 128 KiB of
  // LDS). That row is a measured property of gfx950 / ROCm 7.2, not an architectural guarantee of the LLVM memory model;
  // the alternative (release fence in every workgroup + acquire in the last arriver) measured 1.6x slower on these layers.
  // The tickets are zero at bind time and every launch leaves them zero (the last arriver resets its tile's ticket); a launch
  // that FAILS leaves the context unusable anyway (GI_ERR_HIP), so a stale ticket never meets a later launch of a live context.
#ifdef GI_ABLATION
  if (p.dbg & 4) {   // no split-K hand-off, no epilogue: keep the accumulators alive
    float tsum = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) tsum += acc[i][j][0] + acc[i][j][3];
    if (tsum == 12345.678f) p.out[0] = 1;
    return;
  }
#endif
  if (p.splitk > 1) {
    constexpr int NF = MT * NT;
    constexpr int TILE_BYTES = BM * BN * 4;
    const int tile = (ph * p.ntiles + nt_idx) * p.mtiles + mt_idx;
    const int ntile = p.mtiles * p.ntiles * nph;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.ws, 0, 0x7FFFFF00u, 0x00020000);
    const unsigned mine = (unsigned)(ks * ntile + tile) * TILE_BYTES + tid * 16;
#pragma unroll
    for (int i = 0; i < NF; ++i)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, acc[i / NT][i % NT]), rsW, mine + i * (NTHR * 16), 0, 16);   // sc1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned tk = __hip_atomic_fetch_add(p.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (tk == (unsigned)p.splitk - 1u) ? 1 : 0;
      if (s_last) __hip_atomic_store(p.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[i / NT][i % NT] = f4_t{0.f, 0.f, 0.f, 0.f};
    // split order, whoever is last. The tiles of FOUR splits are requested before the first is added (splits beyond the last one
    // with an out-of-range offset: zeros, not added): as one split per iteration - load its tile, wait, add - the tail was a chain
    // of `splitk` dependent memory round trips of ~1.5 us each on ONE CU while the others still stream (8 splits on d6 / d7).
    constexpr int PF = PF_;
    for (int k0 = 0; k0 < p.splitk; k0 += PF) {
      f4_t v[PF][NF];
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const unsigned src = k0 + j < p.splitk ? (unsigned)((k0 + j) * ntile + tile) * TILE_BYTES + tid * 16 : 0x80000000u;
#pragma unroll
        for (int i = 0; i < NF; ++i) v[j][i] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rsW, src + i * (NTHR * 16), 0, 16));
      }
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        if (k0 + j < p.splitk) {
#pragma unroll
          for (int i = 0; i < NF; ++i) acc[i / NT][i % NT] += v[j][i];
        }
      }
    }
  }

  // ---- epilogue (contract of igemm3.hip) ----------------------------------------------------------------------------------
  auto out_pixel = [&](int m) -> int {
    if (PHASE != 1) return m;
    const int x = m % p.Ws;
    const int tt = m / p.Ws;
    const int y = tt % p.Hs;
    const int n = tt / p.Hs;
    return (n * p.Hout + 2 * y + py) * p.Wout + 2 * x + px;
  };
  constexpr int SLD = BN + 8;
  half_t* stg = (half_t*)smem;
  float* red = (float*)(smem + (int64_t)BM * SLD * 2);   // [NWM][BN][2]
  const bool stats = p.partials || p.stat_acc;
  gi_with_act(p.act_out, [&](auto ACTc) {                // the activation, statistics and bias as compile-time constants (common.h)
  gi_with_bool(stats, [&](auto STc) {
  gi_with_bool(p.bias != nullptr, [&](auto BIc) {
    constexpr int ACT = decltype(ACTc)::value;
    constexpr bool ST = decltype(STc)::value, BI = decltype(BIc)::value;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ch = wn * WN + nt * 16 + 4 * lq;         // this lane's 4 consecutive channels
      float bs[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (BI) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[r] = p.bias[n0 + ch + r];
      }
      float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        h4_t o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[mt][nt][r];
          if constexpr (BI) v += bs[r];
          if constexpr (ST) { s[r] += v; q[r] += v * v; }
          o[r] = (half_t)gi_act_c<ACT>(v);
        }
        *(h4_t*)(stg + (wm * WM + mt * 16 + lr) * SLD + ch) = o;
      }
      if constexpr (ST) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[r] = gi_row16_sum(s[r]); q[r] = gi_row16_sum(q[r]); }
        if (lr == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { red[(wm * BN + ch + r) * 2] = s[r]; red[(wm * BN + ch + r) * 2 + 1] = q[r]; }
        }
      }
    }
  }); }); });
  __syncthreads();
  if ((p.partials || p.stat_acc) && tid < BN) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < NWM; ++w) { s += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }   // (two addends: the same sum as ever)
    if (p.stat_acc) {
      const int grp = (p.stat_pg > 0 && m0 >= p.stat_pg) ? 1 : 0, rep = (mt_idx + ph) & (p.stat_reps - 1);
      gi_stat_add(p.stat_acc, p.cout, rep, grp, 0, n0 + tid, s);
      gi_stat_add(p.stat_acc, p.cout, rep, grp, 1, n0 + tid, q);
    } else {
      const int64_t trow = (int64_t)mt_idx + (int64_t)p.mtiles * ph;
      p.partials[(trow * 2 + 0) * p.cout + n0 + tid] = s;
      p.partials[(trow * 2 + 1) * p.cout + n0 + tid] = q;
    }
  }
  constexpr int CPRO = BN / 8;   // 16-byte chunks per output row
  const int oc = tid % CPRO;
  if (NW != 4 || !p.fold) {      // (the folded normalisation below is the four-wave kernel's: op_igemm7 never asks the other for it)
#pragma unroll 1
    for (int r = tid / CPRO; r < BM; r += NTHR / CPRO) {
      const int m = m0 + r;
      if (m < p.M) {
        const int64_t o = (int64_t)out_pixel(m) * p.ldout + p.coffout + n0 + oc * 8;
        *(u4_t*)(p.out + o * 2) = *(const u4_t*)((const char*)stg + ((int64_t)r * SLD + oc * 8) * 2);
      }
    }
    return;
  }

  // ---- folded normalisation: BatchNorm + activation (+ dropout) of this channel column by the workgroup that completes it ------
  // Hand-off, the form of the split-K fix-up above (guides/MI355X_MICROARCH.md, hand-off table, rows 1 and 3): every raw row is a
  // 16-byte sc1 (write-through) buffer store, the column sums are agent-scope integer atomics (gi_stat_add); every wave drains its
  // stores and atomics (s_waitcnt vmcnt(0)), a workgroup barrier, then ONE lane's agent-scope atomic add on the column's ticket.
  // The workgroup whose add returned (tiles of the column) - 1 reads the accumulators with agent-scope atomic loads and every raw
  // row with 16-byte sc1 buffer loads to registers. No workgroup waits for another one: nothing spins.
  const int out_pixels = (PHASE == 1 ? 4 : 1) * p.M;
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)((int64_t)out_pixels * p.ldout * 2), 0x00020000);
#pragma unroll 1
  for (int r = tid / CPRO; r < BM; r += 256 / CPRO) {
    const int m = m0 + r;
    if (m < p.M) {
      const unsigned o = (unsigned)(out_pixel(m) * p.ldout + p.coffout + n0 + oc * 8) * 2u;
      __builtin_amdgcn_raw_buffer_store_b128(*(const u4_t*)((const char*)stg + ((int64_t)r * SLD + oc * 8) * 2), rsO, o, 0, 16);   // sc1
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned need = (unsigned)(p.mtiles * nph);
    const unsigned tk = __hip_atomic_fetch_add(p.col_tickets + nt_idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (tk == need - 1u) ? 1 : 0;
    if (s_last) __hip_atomic_store(p.col_tickets + nt_idx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  // Everything the column finisher needs from memory is requested at once - its first 16 raw rows, the accumulator words and the
  // layer's parameters are independent loads - because each dependent round trip costs ~2 us while the other CUs stream weights
  // (first version: accumulators, then parameters, then two batches of eight rows, one after the other: +12 us per layer).
  constexpr int RPP = 256 / CPRO, NR = 16;   // rows per pass of the workgroup; loads in flight per thread
  const int ch0 = n0 + oc * 8;
  auto row_off = [&](int r) -> unsigned { return r < out_pixels ? (unsigned)(r * p.ldout + p.coffout + ch0) * 2u : 0x80000000u; };   // beyond: zeros
  u4_t raw[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) raw[k] = __builtin_amdgcn_raw_buffer_load_b128(rsO, row_off(tid / CPRO + k * RPP), 0, 16);   // sc1
  float* aff = (float*)smem;   // [2][BN]: scale, shift of this column (the staged tile has been stored)
  if (tid < BN) {
    float sc, sh;
    bn_from_acc<true>(p.fa, p.cout, n0 + tid, 0, true, sc, sh);
    aff[tid] = sc;
    aff[BN + tid] = sh;
  }
  if (p.fa.zero_next) {        // this column's words of the layer's other accumulator region (rows of cout words: stat_acc.h)
    const int rows = p.fa.zero_words / p.cout;
    for (int i = tid; i < rows * BN; i += 256) p.fa.zero_next[(int64_t)(i / BN) * p.cout + n0 + (i % BN)] = 0ull;
  }
  __syncthreads();
  float fsc[8], fsh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { fsc[e] = aff[oc * 8 + e]; fsh[e] = aff[BN + oc * 8 + e]; }
#pragma unroll 1
  for (int r0 = tid / CPRO; r0 < out_pixels; r0 += RPP * NR) {
    u4_t nxt[NR];
    const bool more = r0 + RPP * NR < out_pixels;
    if (more) {
#pragma unroll
      for (int k = 0; k < NR; ++k) nxt[k] = __builtin_amdgcn_raw_buffer_load_b128(rsO, row_off(r0 + RPP * NR + k * RPP), 0, 16);
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {   // (no `break` in here: it would keep the loop rolled and send raw[] to scratch memory)
      const int r = r0 + k * RPP;
      if (r < out_pixels) {
      const h8_t h = __builtin_bit_cast(h8_t, raw[k]);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {        // bn_apply_kernel's arithmetic (elementwise.hip)
        float t = fmaf((float)h[e], fsc[e], fsh[e]);
        if (p.fact == GI_ACT_RELU) t = t > 0.f ? t : 0.f;
        else if (p.fact == GI_ACT_LRELU) t = t > 0.f ? t : 0.2f * t;
        v[e] = t;
      }
      if (p.fdrop) {
        const int64_t e0 = (int64_t)r * p.cout + ch0;
        if (p.fdrop_thresh) {
          uint8_t kp[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) kp[e] = dropout_keep(p.fdrop_seed, e0 + e, p.fdrop_thresh);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = kp[e] ? v[e] * p.fdrop_scale : 0.f;
          *(uint2*)(p.fdrop + e0) = uint2{(unsigned)kp[0] | ((unsigned)kp[1] << 8) | ((unsigned)kp[2] << 16) | ((unsigned)kp[3] << 24),
                                          (unsigned)kp[4] | ((unsigned)kp[5] << 8) | ((unsigned)kp[6] << 16) | ((unsigned)kp[7] << 24)};
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = p.fdrop[e0 + e] ? v[e] * p.fdrop_scale : 0.f;
        }
      }
      h8_t o8;
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = (half_t)v[e];
      *(h8_t*)(p.fdst + ((int64_t)r * p.flddst + p.fcoffdst + ch0) * 2) = o8;
      }
    }
    if (more) {
#pragma unroll
      for (int k = 0; k < NR; ++k) raw[k] = nxt[k];
    }
  }
}

}  // namespace

const char* gi_igemm3_zero_page(int dev);   // igemm3.hip

// modes 0 (Conv2d 4x4/s2/p1 gather) and 1 (sub-pixel phases), fp16, layers with fewer than 256 tiles of 128 x 128.
// Returns GI_ERR_UNSUPPORTED for what it does not serve (the caller falls back to igemm.hip).
int op_igemm7(hipStream_t st, int mode, IgemmArgs& a) {
  if (mode != 0 && mode != 1) return GI_ERR_UNSUPPORTED;
  if (a.cin % 64 != 0 || a.cout % 64 != 0 || a.force_splitk != 0) return GI_ERR_UNSUPPORTED;
  if (!gi_opt(GI_OPT_IGEMM7)) return GI_ERR_UNSUPPORTED;   // GI_IGEMM7=0: igemm.hip serves these layers
  int max_split = gi_tune("GI_IGEMM7_MAXSPLIT", 8);   // (measured: d7 20.6 us with 16 splits, 18.6 us with 8: the last arriver's tail)
  if (max_split < 1) max_split = 8;
  const int M = a.n * a.Hs * a.Ws;
  const int nph = mode == 1 ? 4 : 1;
  const int mtiles = (M + 127) / 128;
  int BN = (a.cout % 128 == 0 && mtiles * (a.cout / 128) * nph >= 64) ? 128 : 64;
  {   // GI_IGEMM7_BN (ablation build): force the N tile
    const int force_bn = gi_tune("GI_IGEMM7_BN", 0);
    if (force_bn == 64 || (force_bn == 128 && a.cout % 128 == 0)) BN = force_bn;
  }
  const int ntiles = a.cout / BN;
  const int tiles = mtiles * ntiles * nph;
  const int Ktot = (mode == 1 ? 4 : 16) * a.cin, nk = Ktot / 64;
  int splitk = (256 + tiles - 1) / tiles;
  if (splitk > max_split) splitk = max_split;
  if (splitk > nk / 8) splitk = nk / 8;   // at least 8 K tiles per split (u7: 21.7 us with 4 tiles per split, 17.9 us with 8)
  if (splitk < 1) splitk = 1;
  int kps = (nk + splitk - 1) / splitk;
  splitk = (nk + kps - 1) / kps;
  if (splitk > 1 && (!a.tickets || !a.ws || tiles > GI_IGEMM_TICKETS || a.ws_bytes < (int64_t)splitk * tiles * 128 * BN * 4)) return GI_ERR_UNSUPPORTED;
  // Folded normalisation (IgemmFold): taken when ONE workgroup can normalise a channel column in about the time the separate pass
  // spends before its first byte moves (a dependent launch + the accumulator reads: ~5 us): the column finisher reads and writes
  // out_pixels x BN halves at 60 - 100 GB/s (one CU, other workgroups' rows: guides/MI355X_MICROARCH.md "handoff-payload"), i.e.
  // ~1.5 us per 64 KiB each way. The column tickets are the last 32 of the GI_IGEMM_TICKETS words.
  constexpr int COL_TICKETS = 32;
  const int64_t out_pixels = (int64_t)M * nph;
  const int64_t col_bytes = out_pixels * BN * 2;
  const bool fold = a.fold && gi_opt(GI_OPT_BN_FOLD) && a.stat_acc && a.fold->bn.groups == 1 && a.fold->bn.acc == a.stat_acc && a.tickets &&
                    ntiles <= COL_TICKETS && tiles <= GI_IGEMM_TICKETS - COL_TICKETS && col_bytes <= (int64_t)gi_tune("GI_FOLD_MAX_KB", 256) * 1024 &&
                    out_pixels * a.ldout * 2 < (1ll << 31) && out_pixels * a.fold->lddst * 2 < (1ll << 31) && a.ldout % 8 == 0 && a.coffout % 8 == 0 &&
                    a.fold->lddst % 8 == 0 && a.fold->coffdst % 8 == 0 && (a.fold->act == GI_ACT_RELU || a.fold->act == GI_ACT_LRELU || a.fold->act == GI_ACT_NONE);
  int dev = 0;
  GI_HIP(hipGetDevice(&dev));
  const char* zero = gi_igemm3_zero_page(dev);
  if (!zero) return GI_ERR_HIP;
  KP7 kp;
  kp.in = (const char*)a.in; kp.w = (const char*)a.w; kp.out = (char*)a.out; kp.zero = zero;
  kp.bias = a.bias; kp.partials = a.stat_acc ? nullptr : a.partials;
  kp.stat_acc = a.stat_acc; kp.stat_pg = a.stat_pg; kp.stat_reps = a.stat_reps > 0 ? a.stat_reps : 1;
  a.stat_used = a.stat_acc ? 1 : 0;
  GI_REQUIRE(!a.stat_acc || a.stat_pg == 0 || a.stat_pg % 128 == 0, "igemm7: stat_pg=%d must be a multiple of 128", a.stat_pg);
  kp.ws = a.ws; kp.tickets = a.tickets;
  kp.M = M; kp.Hs = a.Hs; kp.Ws = a.Ws;
  kp.cin = a.cin; kp.ldin = a.ldin; kp.coffin = a.coffin;
  kp.cout = a.cout; kp.ldout = a.ldout; kp.coffout = a.coffout;
  kp.Ktot = Ktot; kp.nk = nk; kp.splitk = splitk; kp.kt_per_split = kps;
  kp.relu_in = a.relu_in; kp.act_out = a.act_out;
  kp.relu_cend = a.relu_cend > 0 ? a.relu_cend : a.cin;
  if (mode == 1) { kp.Hin = a.Hs; kp.Win = a.Ws; kp.Hout = 2 * a.Hs; kp.Wout = 2 * a.Ws; }
  else { kp.Hin = 2 * a.Hs; kp.Win = 2 * a.Ws; kp.Hout = a.Hs; kp.Wout = a.Ws; }
  GI_REQUIRE((int64_t)a.n * kp.Hin * kp.Win * a.ldin < (1ll << 31) && (int64_t)a.n * kp.Hout * kp.Wout * a.ldout < (1ll << 31),
             "igemm7: tensor too large for 32-bit offsets");
  kp.mtiles = mtiles; kp.ntiles = ntiles;
  kp.dbg = gi_tune("GI_IGEMM7_DBG", 0);
  kp.fold = fold ? 1 : 0;
  a.fold_applied = kp.fold;
  if (fold) {
    gi_note_fold();
    const IgemmFold& f = *a.fold;
    gi_fill_acc_params(kp.fa, f.bn);
    kp.col_tickets = a.tickets + (GI_IGEMM_TICKETS - COL_TICKETS);
    kp.fdst = (char*)f.dst; kp.flddst = f.lddst; kp.fcoffdst = f.coffdst; kp.fact = f.act;
    kp.fdrop = f.drop_mask; kp.fdrop_scale = f.drop_scale; kp.fdrop_seed = f.drop_seed;
    kp.fdrop_thresh = f.drop_p > 0.f ? gi_dropout_thresh(f.drop_p) : 0u;
  } else {
    kp.col_tickets = nullptr; kp.fa = BnAccP{}; kp.fdst = nullptr; kp.flddst = kp.fcoffdst = kp.fact = 0;
    kp.fdrop = nullptr; kp.fdrop_scale = 1.f; kp.fdrop_seed = 0; kp.fdrop_thresh = 0;
  }
  const int nyz = ntiles * nph * splitk;
  const int grid = mtiles >= 8 ? ((mtiles + 7) / 8) * 8 * nyz : mtiles * nyz;
  // shipped choice, measured on d5 / d6 / d7 / u7 / u6 at the headline batch (tools/r4_small.sh, profiles/r04_igemm7_variants.txt):
  // four stages and one split per tail iteration. A six-stage ring on the 64-column tiles (four K tiles in flight) ran 3 - 5 us
  // SLOWER per layer (the five-tile prologue burst of every CU delays the first tile by more than the deeper ring gains), two or
  // four splits of the tail in flight changed nothing (+-0.5 us): the tail's round trips are not what these launches wait for.
  constexpr int GI7_NSTG = 4, GI7_PF = 1;
  int nstg = GI7_NSTG, pf = GI7_PF;
#ifdef GI_ABLATION
  nstg = gi_tune("GI_IGEMM7_NSTG", GI7_NSTG);
  pf = gi_tune("GI_IGEMM7_PF", GI7_PF);
  if (BN == 128 || nstg != 6) nstg = 4;
  if (pf != 2 && pf != 4) pf = 1;
#endif
  // waves per workgroup (kernel header): eight unless the folded normalisation is asked for (the four-wave kernel's) or GI_IGEMM7_WAVES=4
  const int nw = (fold || gi_opt(GI_OPT_IGEMM7_WAVES) == 4) ? 4 : 8;
  const int ring = nstg * (128 + BN) * 128, epi = 128 * (BN + 8) * 2 + 4 * BN * 8;
  const int LDS = ring > epi ? ring : epi;
  const int vi = (BN == 64 ? 2 : 0) + mode;
  auto launch = [&](auto NS, auto PFc, auto NWc) -> int {
    constexpr int ns = decltype(NS)::value, pfc = decltype(PFc)::value, nwc = decltype(NWc)::value;
    static GiDevOnce attr_set[4];
    const void* fn[4] = {(const void*)igemm7_kernel<0, 128, ns, pfc, nwc>, (const void*)igemm7_kernel<1, 128, ns, pfc, nwc>,
                         (const void*)igemm7_kernel<0, 64, ns, pfc, nwc>, (const void*)igemm7_kernel<1, 64, ns, pfc, nwc>};
    if (attr_set[vi].first()) { GI_HIP(hipFuncSetAttribute(fn[vi], hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)); }
    switch (vi) {
      case 0: hipLaunchKernelGGL((igemm7_kernel<0, 128, ns, pfc, nwc>), dim3(grid), dim3(nwc * 64), LDS, st, kp); break;
      case 1: hipLaunchKernelGGL((igemm7_kernel<1, 128, ns, pfc, nwc>), dim3(grid), dim3(nwc * 64), LDS, st, kp); break;
      case 2: hipLaunchKernelGGL((igemm7_kernel<0, 64, ns, pfc, nwc>), dim3(grid), dim3(nwc * 64), LDS, st, kp); break;
      default: hipLaunchKernelGGL((igemm7_kernel<1, 64, ns, pfc, nwc>), dim3(grid), dim3(nwc * 64), LDS, st, kp); break;
    }
    return GI_OK;
  };
  using I1_ = std::integral_constant<int, 1>;
  using I2_ = std::integral_constant<int, 2>;
  using I4_ = std::integral_constant<int, 4>;
  using I6_ = std::integral_constant<int, 6>;
  using I8_ = std::integral_constant<int, 8>;
#ifdef GI_ABLATION
  if (nw == 8) GI_TRY(launch(I4_{}, I1_{}, I8_{}));
  else if (nstg == 6) { if (pf == 4) GI_TRY(launch(I6_{}, I4_{}, I4_{})); else if (pf == 2) GI_TRY(launch(I6_{}, I2_{}, I4_{})); else GI_TRY(launch(I6_{}, I1_{}, I4_{})); }
  else { if (pf == 4) GI_TRY(launch(I4_{}, I4_{}, I4_{})); else if (pf == 2) GI_TRY(launch(I4_{}, I2_{}, I4_{})); else GI_TRY(launch(I4_{}, I1_{}, I4_{})); }
#else
  (void)nstg; (void)pf;
  if (nw == 8) GI_TRY(launch(std::integral_constant<int, GI7_NSTG>{}, std::integral_constant<int, GI7_PF>{}, I8_{}));
  else GI_TRY(launch(std::integral_constant<int, GI7_NSTG>{}, std::integral_constant<int, GI7_PF>{}, I4_{}));
#endif
  { static const char* nm[8] = {"igemm7<0,128>", "igemm7<1,128>", "igemm7<0,64>", "igemm7<1,64>",
                                "igemm7<0,128>+bn", "igemm7<1,128>+bn", "igemm7<0,64>+bn", "igemm7<1,64>+bn"}; gi_note_kernel(nm[vi + (fold ? 4 : 0)]); }
  GI_LAUNCH_CHECK();
  a.ntiles_out = mtiles * nph;
  return GI_OK;
}
