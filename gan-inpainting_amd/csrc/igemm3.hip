// fp16 implicit-GEMM, second generation (gfx950): 256x128 output tile, 8 waves (4x2, 64x64 each),
// K tile = 64 halves, THREE-stage LDS ring filled by LDS-DMA (global_load_lds_dwordx4: no VGPR
// staging, no ds_write), counted vmcnt + one raw s_barrier per K tile, MFMA 16x16x32 f16.
//
// Same algebra and LDS image as igemm.hip (rows of 128 bytes, 16-byte chunk c stored at physical
// chunk c ^ (row & 7)); because an LDS-DMA wave instruction writes 1 KiB linearly (lane l -> slot l)
// the swizzle is applied to the per-lane SOURCE address: lane l fetches chunk (l&7)^((l>>3)&7) of row
// l>>3 of its 8-row block. Padding taps read a zero page instead (the DMA cannot zero-fill), the
// decoder's ReLU is applied to the A fragments after the LDS read (integer pk_max).
// Block order is XCD-aware: the (N tile, phase) blocks that share an M tile run back to back on one
// XCD so the gathered input rows are served by that XCD's L2.
#include <stdlib.h>

#include "common.h"
#include "stat_acc.h"

namespace {

struct KP3 {
  const char* in;
  const char* w;
  char* out;
  const char* zero;   // >= 4 KiB of zeros
  const float* bias;
  float* partials;
  unsigned long long* stat_acc; int stat_pg, stat_reps;   // IgemmArgs::stat_acc
  int M, Hs, Ws;
  int cin, ldin, coffin;
  int cout, ldout, coffout;
  int Ktot, nk;
  int relu_in, act_out;
  int relu_cend;      // ReLU is applied to input channels [0, relu_cend) only (the skip half of a concat buffer)
  int Hin, Win, Hout, Wout;
  int mtiles, ntiles;
  int dbg;            // timing-only ablations (GI_IGEMM3_DBG): 1 no LDS-DMA, 2 no MFMA, 4 no fragment reads
};


__device__ __forceinline__ h8_t relu_h8(h8_t v) {
  typedef short s8_t __attribute__((ext_vector_type(8)));
  s8_t h = __builtin_bit_cast(s8_t, v);
  const s8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
  h = __builtin_elementwise_max(h, z);
  return __builtin_bit_cast(h8_t, h);
}

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// BN = 128, or 64 for the layers with 64 output channels. NW = waves per workgroup: 8 (4x2, two per
// SIMD) or 4 (4x1, ONE per SIMD, each wave owns 64 rows x BN columns so that every A fragment feeds
// BN/16 MFMAs and the per-K-tile bookkeeping is issued once per SIMD instead of twice).
template <int PHASE, int BN, int NW, int DBG = 0>   // DBG: compile-time timing ablations (tools only)
__global__ void __launch_bounds__(NW * 64, NW / 4) igemm3_kernel(KP3 p) {
  constexpr int BM = 256, BK = 64;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;   // 64 / 48 / 40 KiB
  constexpr int NSTG = BN == 256 ? 2 : 3;   // 256x256 tiles: two 64 KiB stages (half the loaded bytes per FLOP)
  constexpr int AJ = 32 / NW, BJ = (BN / 8) / NW;    // 8-row blocks per wave per tile
  constexpr int WNW = NW / 4;            // waves along N
  constexpr int WN = BN / WNW;           // columns per wave
  constexpr int MT = 4, NT = WN / 16;
  constexpr int NTHREADS = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;

  // ---- XCD-aware tile order ------------------------------------------------------------------
  const int nyz = p.ntiles * (PHASE == 1 ? 4 : 1);
  const int bid = blockIdx.x;
  const int xcd = bid & 7, local = bid >> 3;
  const int mt_idx = (local / nyz) * 8 + xcd;
  if (mt_idx >= p.mtiles) return;
  const int yz = local % nyz;
  const int nt_idx = yz % p.ntiles;
  const int ph = yz / p.ntiles;
  const int py = ph >> 1, px = ph & 1;
  const int m0 = mt_idx * BM, n0 = nt_idx * BN;
  const char* wptr = p.w + (PHASE == 1 ? (int64_t)ph * p.cout * p.Ktot * 2 : 0);

  // ---- per-lane gather rows ------------------------------------------------------------------
  const int lrow = lane >> 3;                       // row inside the 8-row block
  const int lchunk = (lane & 7) ^ (lrow & 7);       // logical 16-byte chunk this lane fetches
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
      if (PHASE == 2) {          // 3x3 / stride 1 / pad 1 (VGG features): tap (ky,kx) reads (y-1+ky, x-1+kx)
        const int y0 = y - 1, x0 = x - 1;
        abase[j] = ((n * p.Hin + y0) * p.Win + x0) * p.ldin + p.coffin;
#pragma unroll
        for (int tt = 0; tt < 9; ++tt) {
          const int iy = y0 + tt / 3, ix = x0 + tt % 3;
          if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) amask[j] |= 1u << tt;
        }
      } else if (PHASE) {
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
  int tap = 0, c0 = 0;
  auto set_tap = [&]() {
    int toff;
    if (PHASE == 2) toff = ((tap / 3) * p.Win + (tap % 3)) * p.ldin;
    else if (PHASE) toff = -((tap >> 1) * p.Win + (tap & 1)) * p.ldin;
    else toff = ((tap >> 2) * p.Win + (tap & 3)) * p.ldin;
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      pa[j] = ((amask[j] >> tap) & 1u) ? p.in + (int64_t)(abase[j] + toff) * 2 + lchunk * 16 : p.zero + lchunk * 16;
  };
#pragma unroll
  for (int j = 0; j < BJ; ++j) pb[j] = wptr + (int64_t)(n0 + (wave * BJ + j) * 8 + lrow) * p.Ktot * 2 + lchunk * 16;
  set_tap();

  int kt_issue = 0;
  // one LDS-DMA piece (1 KiB) of the tile being issued: pieces 0..AJ-1 = A blocks, AJ..AJ+BJ-1 = B blocks
  auto issue_piece = [&](auto STG, auto PIECE) {
    constexpr int stage = decltype(STG)::value;
    constexpr int j = decltype(PIECE)::value;
    if constexpr (!(DBG & 1)) {
      if constexpr (j < AJ) glds16(pa[j] + c0 * 2, smem + stage * STAGE + wave * (AJ * 1024) + j * 1024);
      else glds16(pb[j - AJ] + (int64_t)kt_issue * (BK * 2), smem + stage * STAGE + A_BYTES + wave * (BJ * 1024) + (j - AJ) * 1024);
    }
  };
  auto issue_done = [&]() {
    ++kt_issue;
    c0 += BK;
    if (c0 >= p.cin) { c0 = 0; ++tap; set_tap(); }
  };
  auto issue = [&](auto STG) {
    static_for<AJ + BJ>([&](auto J) { issue_piece(STG, J); });
    issue_done();
  };

  f4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 15, lq = lane >> 4;
  const int rdA0 = (wm * 64 + lr) * 128 + (((0 + lq) ^ (lr & 7)) << 4);
  const int rdA1 = (wm * 64 + lr) * 128 + (((4 + lq) ^ (lr & 7)) << 4);
  const int rdB0 = A_BYTES + (wn * WN + lr) * 128 + (((0 + lq) ^ (lr & 7)) << 4);
  const int rdB1 = A_BYTES + (wn * WN + lr) * 128 + (((4 + lq) ^ (lr & 7)) << 4);
  const int relu_cend = p.relu_in ? p.relu_cend : 0;
  int cc0 = 0;   // channel offset of the tile being computed

  // compute tile from stage CUR; when ISS is set, the AJ+BJ LDS-DMA pieces of the tile two ahead are
  // issued one at a time BETWEEN the MFMAs (one piece per NMF/(AJ+BJ) MFMAs): a piece occupies the CU's
  // TA/L1 path for >= 16 cycles (1 KiB at 64 B/clk), and a burst of 8 waves x 6 pieces leaves the matrix
  // pipe idle for ~800 cycles per K tile and the TA idle during the MFMA phase (ablation in DESIGN.md).
  auto compute = [&](auto STG, auto NXT, auto ISS) {
    constexpr int stage = decltype(STG)::value;
    constexpr bool iss = decltype(ISS)::value;
    constexpr int NMF = 2 * MT * NT, NPC = AJ + BJ;
    const bool relu_in = cc0 < relu_cend;   // wave-uniform
    const char* s = smem + stage * STAGE;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      h8_t af[MT], bf[NT];
      const int oa = k2 ? rdA1 : rdA0, ob = k2 ? rdB1 : rdB0;
      if constexpr ((DBG & 4) != 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { af[mt] = h8_t{1, 1, 1, 1, 1, 1, 1, 1}; asm volatile("" : "+v"(af[mt])); }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { bf[nt] = h8_t{1, 1, 1, 1, 1, 1, 1, 1}; asm volatile("" : "+v"(bf[nt])); }
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *(const h8_t*)(s + oa + mt * 2048);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = *(const h8_t*)(s + ob + nt * 2048);
      }
      if (relu_in) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = relu_h8(af[mt]);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int idx = k2 * MT * NT + mt * NT + nt;
          if constexpr ((DBG & 2) != 0) {
            asm volatile("" :: "v"(af[mt]));
            asm volatile("" :: "v"(bf[nt]));
          } else {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[nt], af[mt], acc[mt][nt], 0, 0, 0);   // D^T: rows = channels
          }
          if constexpr (iss) {
            // piece q goes after MFMA number ((q+1)*NMF)/(NPC+1) - 1  (idx is a constant after unrolling)
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

  // ---- 3-stage ring. Per tile each wave issues AJ+BJ = 6 LDS-DMA instructions; before computing tile
  //      t its own 6 must have landed (vmcnt(6) leaves tile t+1 in flight), the barrier then makes every
  //      wave's share visible and proves stage (t+2)%3 == (t-1)%3 is no longer being read. --------------
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  const int nk = p.nk;
  using T1 = std::integral_constant<bool, true>;
  using T0 = std::integral_constant<bool, false>;
  if constexpr (NSTG == 3) {
    issue(S0{});
    if (nk > 1) issue(S1{});
    auto wait_tile = [&](bool newer_in_flight) {
      if (newer_in_flight) {
        if constexpr (AJ + BJ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (AJ + BJ == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if constexpr (AJ + BJ == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else { static_assert(AJ + BJ == 10 || NSTG != 3, "unexpected DMA count"); asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    int t = 0;
    const int nfull = (DBG & 16) ? 0 : nk - 2;      // tiles whose step also issues tile t+2
    for (; t + 2 < nfull; t += 3) {                 // steady state, stage indices are compile-time
      wait_tile(true); compute(S0{}, S2{}, T1{});
      wait_tile(true); compute(S1{}, S0{}, T1{});
      wait_tile(true); compute(S2{}, S1{}, T1{});
    }
    // remaining tiles (at most 4), generic in the stage
    for (; t < ((DBG & 16) ? 0 : nk); ++t) {
      wait_tile(t + 1 < nk);
      const bool is = t + 2 < nk;
      switch (t % 3) {
        case 0: if (is) compute(S0{}, S2{}, T1{}); else compute(S0{}, S2{}, T0{}); break;
        case 1: if (is) compute(S1{}, S0{}, T1{}); else compute(S1{}, S0{}, T0{}); break;
        default: if (is) compute(S2{}, S1{}, T1{}); else compute(S2{}, S1{}, T0{}); break;
      }
    }
  } else {
    // two stages: tile t+1 streams in while tile t is multiplied
    issue(S0{});
    for (int t = 0; t < nk; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const bool is = t + 1 < nk;
      if (t & 1) { if (is) compute(S1{}, S0{}, T1{}); else compute(S1{}, S0{}, T0{}); }
      else { if (is) compute(S0{}, S1{}, T1{}); else compute(S0{}, S1{}, T0{}); }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue (same contract as igemm.hip) --------------------------------------------------------
  auto out_pixel = [&](int m) -> int {
    if (PHASE != 1) return m;
    const int x = m % p.Ws;
    const int t = m / p.Ws;
    const int y = t % p.Hs;
    const int n = t / p.Hs;
    return (n * p.Hout + 2 * y + py) * p.Wout + 2 * x + px;
  };
  constexpr int SLD = BN + 8;
  half_t* stg = (half_t*)smem;
  float* red = (float*)(smem + (int64_t)BM * SLD * 2);   // [4][BN][2]
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
        *(h4_t*)(stg + (wm * 64 + mt * 16 + lr) * SLD + ch) = o;
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
  if ((p.partials || p.stat_acc) && tid < BN) {   // (NTHREADS >= 256 >= BN)
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { s += red[(i * BN + tid) * 2]; q += red[(i * BN + tid) * 2 + 1]; }
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
#pragma unroll 1
  for (int r = tid / CPRO; r < BM; r += NTHREADS / CPRO) {
    const int m = m0 + r;
    if (m < p.M && !(DBG & 8)) {
      const int64_t o = (int64_t)out_pixel(m) * p.ldout + p.coffout + n0 + oc * 8;
      *(u4_t*)(p.out + o * 2) = *(const u4_t*)((const char*)stg + ((int64_t)r * SLD + oc * 8) * 2);
    }
  }
}

char* g_zero_page[16] = {nullptr};

}  // namespace

// the zero page padding taps read (shared with igemm5.hip); allocated on first use, one per device
const char* gi_igemm3_zero_page(int dev) {
  if (!g_zero_page[dev & 15]) {
    if (hipMalloc((void**)&g_zero_page[dev & 15], 8192) != hipSuccess) return nullptr;
    if (hipMemset(g_zero_page[dev & 15], 0, 8192) != hipSuccess) return nullptr;
  }
  return g_zero_page[dev & 15];
}

int op_igemm5(hipStream_t st, int mode, IgemmArgs& a);   // igemm5.hip: halo-resident kernels (modes 0 and 1)

// mode: 0 = Conv2d 4x4/s2/p1 gather, 1 = sub-pixel phases (ConvTranspose2d forward / Conv2d dgrad),
//       2 = Conv2d 3x3/s1/p1 (VGG features; weights [cout][9*cin], tap-major).
// returns GI_ERR_UNSUPPORTED when the shape is not served by this kernel (caller falls back)
int op_igemm3(hipStream_t st, int mode, IgemmArgs& a) {
  if (a.cin % 64 != 0 || a.cout % 64 != 0 || a.cin > 2048) return GI_ERR_UNSUPPORTED;
  const int use5 = gi_opt(GI_OPT_IGEMM5);   // GI_IGEMM5: bit 0 / 1 / 2 = halo-resident kernel for mode 1 / 0 / 2 (default all)
  if ((mode == 1 && (use5 & 1)) || (mode == 0 && (use5 & 2)) || (mode == 2 && (use5 & 4))) {
    const int rc = op_igemm5(st, mode, a);
    if (rc != GI_ERR_UNSUPPORTED) return rc;
  }
  const int M = a.n * a.Hs * a.Ws;
  const int nph = mode == 1 ? 4 : 1;
  int BN = (a.cout % 128 == 0) ? 128 : 64;
  if (mode != 2) {   // too few tiles to fill 256 CUs: the split-K path of igemm.hip serves those layers
    const int tiles = ((M + 255) / 256) * (a.cout / BN) * nph;
    if (tiles < 128) return GI_ERR_UNSUPPORTED;
    // 128..255 tiles leave CUs idle (one 8-wave workgroup per CU): 64-wide N tiles double the workgroups
    const int narrow = gi_tune("GI_IGEMM3_NARROW", 1);
    if (narrow && BN == 128 && tiles < 256) BN = 64;
  }
  int dev = 0;
  GI_HIP(hipGetDevice(&dev));
  if (!g_zero_page[dev & 15]) {
    GI_HIP(hipMalloc((void**)&g_zero_page[dev & 15], 8192));
    GI_HIP(hipMemset(g_zero_page[dev & 15], 0, 8192));
  }
  KP3 kp;
  kp.in = (const char*)a.in; kp.w = (const char*)a.w; kp.out = (char*)a.out; kp.zero = g_zero_page[dev & 15];
  kp.bias = a.bias; kp.partials = a.stat_acc ? nullptr : a.partials;
  kp.stat_acc = a.stat_acc; kp.stat_pg = a.stat_pg; kp.stat_reps = a.stat_reps > 0 ? a.stat_reps : 1;
  a.stat_used = a.stat_acc ? 1 : 0;
  GI_REQUIRE(!a.stat_acc || a.stat_pg == 0 || a.stat_pg % 256 == 0, "igemm3: stat_pg=%d must be a multiple of 256", a.stat_pg);
  kp.M = M; kp.Hs = a.Hs; kp.Ws = a.Ws;
  kp.cin = a.cin; kp.ldin = a.ldin; kp.coffin = a.coffin;
  kp.cout = a.cout; kp.ldout = a.ldout; kp.coffout = a.coffout;
  kp.Ktot = (mode == 1 ? 4 : (mode == 2 ? 9 : 16)) * a.cin;
  kp.nk = kp.Ktot / 64;
  kp.relu_in = a.relu_in; kp.act_out = a.act_out;
  kp.relu_cend = a.relu_cend > 0 ? a.relu_cend : a.cin;
  kp.dbg = 0;
#ifdef GI_ABLATION   // timing-only ablation kernels compute wrong results: compiled only with `build.sh -DGI_ABLATION`
  { const char* e = getenv("GI_IGEMM3_DBG"); kp.dbg = e ? atoi(e) : 0; }
#endif
  if (mode == 1) { kp.Hin = a.Hs; kp.Win = a.Ws; kp.Hout = 2 * a.Hs; kp.Wout = 2 * a.Ws; }
  else if (mode == 2) { kp.Hin = a.Hs; kp.Win = a.Ws; kp.Hout = a.Hs; kp.Wout = a.Ws; }
  else { kp.Hin = 2 * a.Hs; kp.Win = 2 * a.Ws; kp.Hout = a.Hs; kp.Wout = a.Ws; }
  GI_REQUIRE((int64_t)a.n * kp.Hin * kp.Win * a.ldin < (1ll << 31) && (int64_t)a.n * kp.Hout * kp.Wout * a.ldout < (1ll << 31),
             "igemm3: tensor too large for 32-bit offsets");
  kp.mtiles = (M + 255) / 256;
  kp.ntiles = a.cout / BN;
  const int nyz = kp.ntiles * nph;
  const int grid = ((kp.mtiles + 7) / 8) * 8 * nyz;
  constexpr int LDS_MAX = 3 * (256 + 128) * 128;
  const int ring = 3 * (256 + BN) * 128, epi = 256 * (BN + 8) * 2 + 4 * BN * 8;
  const int LDS = ring > epi ? ring : epi;
  static GiDevOnce attr_set[6];
  const void* fn[6] = {(const void*)igemm3_kernel<0, 128, 8>, (const void*)igemm3_kernel<1, 128, 8>, (const void*)igemm3_kernel<2, 128, 8>,
                       (const void*)igemm3_kernel<0, 64, 8>,  (const void*)igemm3_kernel<1, 64, 8>,  (const void*)igemm3_kernel<2, 64, 8>};
  const int vi = (BN == 64 ? 3 : 0) + mode;
  if (attr_set[vi].first()) { GI_HIP(hipFuncSetAttribute(fn[vi], hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX)); }
  const dim3 g(grid), b(512);
#ifdef GI_ABLATION
  if (kp.dbg && vi == 1) {   // timing-only ablation builds of the PHASE / 128 kernel (GI_IGEMM3_DBG, tools only)
    static GiDevOnce dbg_attr;
    if (dbg_attr.first()) {
      GI_HIP(hipFuncSetAttribute((const void*)igemm3_kernel<1, 128, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
      GI_HIP(hipFuncSetAttribute((const void*)igemm3_kernel<1, 128, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
      GI_HIP(hipFuncSetAttribute((const void*)igemm3_kernel<1, 128, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
      GI_HIP(hipFuncSetAttribute((const void*)igemm3_kernel<1, 128, 8, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
      GI_HIP(hipFuncSetAttribute((const void*)igemm3_kernel<1, 128, 8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
      GI_HIP(hipFuncSetAttribute((const void*)igemm3_kernel<1, 128, 8, 15>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
      GI_HIP(hipFuncSetAttribute((const void*)igemm3_kernel<1, 128, 8, 31>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
    }
    switch (kp.dbg) {
      case 1: hipLaunchKernelGGL((igemm3_kernel<1, 128, 8, 1>), g, b, LDS, st, kp); break;
      case 2: hipLaunchKernelGGL((igemm3_kernel<1, 128, 8, 2>), g, b, LDS, st, kp); break;
      case 4: hipLaunchKernelGGL((igemm3_kernel<1, 128, 8, 4>), g, b, LDS, st, kp); break;
      case 7: hipLaunchKernelGGL((igemm3_kernel<1, 128, 8, 7>), g, b, LDS, st, kp); break;
      case 8: hipLaunchKernelGGL((igemm3_kernel<1, 128, 8, 8>), g, b, LDS, st, kp); break;
      case 15: hipLaunchKernelGGL((igemm3_kernel<1, 128, 8, 15>), g, b, LDS, st, kp); break;
      default: hipLaunchKernelGGL((igemm3_kernel<1, 128, 8, 31>), g, b, LDS, st, kp); break;
    }
    GI_LAUNCH_CHECK();
    a.ntiles_out = kp.mtiles * nph;
    return GI_OK;
  }
#endif
  switch (vi) {
    case 0: hipLaunchKernelGGL((igemm3_kernel<0, 128, 8>), g, b, LDS, st, kp); break;
    case 1: hipLaunchKernelGGL((igemm3_kernel<1, 128, 8>), g, b, LDS, st, kp); break;
    case 2: hipLaunchKernelGGL((igemm3_kernel<2, 128, 8>), g, b, LDS, st, kp); break;
    case 3: hipLaunchKernelGGL((igemm3_kernel<0, 64, 8>), g, b, LDS, st, kp); break;
    case 4: hipLaunchKernelGGL((igemm3_kernel<1, 64, 8>), g, b, LDS, st, kp); break;
    default: hipLaunchKernelGGL((igemm3_kernel<2, 64, 8>), g, b, LDS, st, kp); break;
  }
  { static const char* nm[6] = {"igemm3<0,128>", "igemm3<1,128>", "igemm3<2,128>", "igemm3<0,64>", "igemm3<1,64>", "igemm3<2,64>"}; gi_note_kernel(nm[vi]); }
  GI_LAUNCH_CHECK();
  a.ntiles_out = kp.mtiles * nph;
  return GI_OK;
}
