// fp16 implicit GEMMs of the 4x4 / stride-2 / pad-1 family with an LDS-RESIDENT INPUT HALO (gfx950), same contract
// as igemm3_kernel: MODE 1 = sub-pixel phases (ConvTranspose2d forward / Conv2d input gradient), MODE 0 = the
// stride-2 gather (Conv2d forward / ConvTranspose2d input gradient), MODE 2 = Conv2d 3x3 / stride 1 / pad 1 (VGG
// features: a (TH+2) x (TW+2) halo serves nine taps), MODE 3 = MODE 1 for layers with 64 output channels per
// N tile: BOTH px phases of a patch in one workgroup (the 128 columns are px 0 | px 1 of 64 channels, one halo of
// TW+2 columns feeds both), so that those layers run on 64 x 64 wave tiles too.
//
// igemm3 gathers the A operand tap by tap: every input row is fetched once per tap (4x in this mode), and the
// hardware counters / ablations show the kernel bound by LDS-DMA issue, i.e. by loaded bytes per MAC
// (48 KiB per 256x128x64 MAC step). Here a workgroup owns a TH x TW patch of the small grid (TH*TW = 256 output
// pixels of ONE sub-pixel phase). Per 64-channel chunk it loads the (TH+1) x (TW+1) input halo ONCE (<= 40 KiB,
// double buffered) and runs the four taps from it: the A fragment of tap (ty,tx) for pixel (y,x) is LDS row
// (y+1-ty)*(TW+1) + (x+1-tx). Only the 16 KiB weight slice of each (chunk, tap) step still streams through a
// three-stage ring: 25.5 instead of 48 KiB of LDS-DMA per step.
// MODE 0 uses the polyphase view of a stride-2 convolution: restricted to the input pixels of one parity class
// q = (qy,qx) it is a 2x2 / stride-1 convolution (input row 2y-1+ky has parity qy for ky = 2*ty + 1 - qy... i.e.
// ky = qy ? 2*ty : 2*ty+1), so per channel chunk FOUR halos (one per parity class, rows picked with stride 2 from
// the large tensor) are loaded in turn and each serves four taps: the same pipeline with 4x as many halo groups.
//   LDS: A halo 2 x 40 KiB | B ring 3 x BN*128 B   (128 KiB for BN = 128); rows of 128 B, 16-byte chunk c of row
//   r at physical chunk c ^ (r & 7) (applied to the per-lane DMA source address, as in igemm3).
//   Step s = (chunk c, tap): counted vmcnt + ONE barrier; during its 2x16 MFMAs per wave the pieces of the next
//   chunk's halo (5 per wave and chunk, split 2,1,1,1 over the four taps) and of B(s+2) are issued.
#include <stdlib.h>

#include "common.h"
#include "stat_acc.h"
#include "halo_args.h"

const char* gi_igemm3_zero_page(int dev);   // igemm3.hip
int op_igemm8_launch(hipStream_t st, int mode, bool dual, bool relu, int grid, const KP5& kp, int bn);   // igemm8.hip

namespace {

__device__ __forceinline__ h8_t relu5(h8_t v) {
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
// one LDS-DMA piece through a buffer descriptor: lane l writes 16 bytes at lds_wave_base + 16 l, read from
// base + voff (per lane) + soff (wave-uniform); out-of-range offsets write zeros
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void wait_vm(int n) {   // n is wave-uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
  }
}

// ---- epilogue shared by igemm5 / igemm6 (contract of igemm3): bias / activation, per-tile column statistics, the tile
//      staged through LDS and stored with 16-byte rows; optional fused activation backward (IgemmArgs::mask) -------------
template <int MODE, int BN>
__device__ __forceinline__ void epilogue5(const KP5& p, f4_t (&acc)[4][BN / 32], char* smem, int tid, int lane, int wm, int wn,
                                          int mt_idx, int nt_idx, int ph, int py, int px, int n0, int img, int y0, int x0, int lgTW) {
  constexpr bool DUAL = MODE == 3;
  constexpr bool PH = MODE == 1 || MODE == 3;
  constexpr int BM = 256, WN = BN / 2, MT = 4, NT = WN / 16;
  const int lr = lane & 15, lq = lane >> 4;
  auto out_pixel = [&](int m) -> int {
    const int ty_l = m >> lgTW, tx_l = m & (p.TW - 1);
    if constexpr (PH) return (img * 2 * p.Hs + 2 * (y0 + ty_l) + py) * (2 * p.Ws) + 2 * (x0 + tx_l) + px;   // DUAL: + 1 for px 1, by the caller
    else return (img * p.Hs + y0 + ty_l) * p.Ws + x0 + tx_l;   // MODE 0 / 2: the tile's own pixels
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
      const int ch = wn * WN + nt * 16 + 4 * lq;         // column of the tile (DUAL: px * 64 + channel)
      float bs[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (BI) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[r] = p.bias[n0 + (DUAL ? (ch & 63) : ch) + r];
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
  if ((p.partials || p.stat_acc) && tid < BN) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { s += red[(i * BN + tid) * 2]; q += red[(i * BN + tid) * 2 + 1]; }
    const int phr = MODE == 1 ? ph : (DUAL ? py * 2 + (tid >> 6) : 0);       // sub-pixel phase of this column
    const int col = DUAL ? (tid & 63) : tid;
    if (p.stat_acc) {   // a patch lies inside one image, i.e. inside one BatchNorm population
      const int grp = (p.stat_pg > 0 && mt_idx * BM >= p.stat_pg) ? 1 : 0, rep = (mt_idx + phr) & (p.stat_reps - 1);
      gi_stat_add(p.stat_acc, p.cout, rep, grp, 0, n0 + col, s);
      gi_stat_add(p.stat_acc, p.cout, rep, grp, 1, n0 + col, q);
    } else {
      const int64_t trow = (int64_t)mt_idx + (int64_t)p.mtiles * phr;
      p.partials[(trow * 2 + 0) * p.cout + n0 + col] = s;
      p.partials[(trow * 2 + 1) * p.cout + n0 + col] = q;
    }
  }
  constexpr int CPRO = BN / 8;
  const int oc = tid % CPRO;
  // fused BatchNorm-backward reduction: this thread's 8 channels are fixed (oc), its rows vary
  const bool bwd = !DUAL && p.bwd_acc != nullptr;
  float bsc[8], bsh[8], bmu[8], biv[8], bs[8], bsx[8];
  if (bwd) {
    const int go = (p.bwd_pg_tiles > 0 && mt_idx >= p.bwd_pg_tiles) ? p.bwd_stride : 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = go + n0 + oc * 8 + e;
      bsc[e] = p.bwd_scale[ch]; bsh[e] = p.bwd_shift[ch]; bmu[e] = p.bwd_mean[ch]; biv[e] = p.bwd_inv[ch];
      bs[e] = bsx[e] = 0.f;
    }
  }
  // A thread copies NR rows (one 16-byte chunk each). With a fused mask / add / BatchNorm reduction every row also needs global
  // loads; issued row by row behind the previous row's store they run one latency at a time (the pointers may alias for the
  // compiler) - 150 us for the critic's conv2 input gradient, whose epilogue moves 268 MB. All loads of the NR rows go first.
  constexpr int NR = BM / (512 / CPRO);
  int opxs[NR];
  u4_t mk[NR], ad[NR], xs[NR];
  const int och = n0 + (DUAL ? (oc & 7) : oc) * 8;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int r = tid / CPRO + k * (512 / CPRO);
    opxs[k] = out_pixel(r) + (DUAL ? (oc >> 3) : 0);
    const int64_t opx = opxs[k];
    if (p.mask) mk[k] = *(const u4_t*)(p.mask + (opx * p.ldmask + p.coffmask + och) * 2);
    if (p.mask && p.add) ad[k] = *(const u4_t*)(p.add + (opx * p.ldadd + p.coffadd + och) * 2);
    if (bwd) xs[k] = *(const u4_t*)(p.bwd_x + (opx * p.bwd_ldx + och) * 2);
  }
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int r = tid / CPRO + k * (512 / CPRO);
    const int64_t opx = opxs[k];
    const int64_t o = opx * p.ldout + p.coffout + och;
    u4_t v = *(const u4_t*)((const char*)stg + ((int64_t)r * SLD + oc * 8) * 2);
    if (bwd) {
      const h8_t xv = __builtin_bit_cast(h8_t, xs[k]);
      const h8_t gv = __builtin_bit_cast(h8_t, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xf = (float)xv[e];
        const float dz = (float)gv[e] * (fmaf(xf, bsc[e], bsh[e]) > 0.f ? 1.f : p.bwd_slope);
        bs[e] += dz;
        bsx[e] = fmaf(dz, (xf - bmu[e]) * biv[e], bsx[e]);
      }
    }
    if (p.mask) {   // same arithmetic as the separate pass: fp16 value -> fp32 * slope -> fp16
      const h8_t m = __builtin_bit_cast(h8_t, mk[k]);
      h8_t hv = __builtin_bit_cast(h8_t, v);
      if (p.add) {
        const h8_t a8 = __builtin_bit_cast(h8_t, ad[k]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool pos = (float)m[e] > 0.f;
          const float g = (float)hv[e] + (pos ? (float)a8[e] : 0.f);
          hv[e] = (half_t)(pos ? g : g * p.mask_slope);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[e] = (float)m[e] > 0.f ? hv[e] : (half_t)((float)hv[e] * p.mask_slope);
      }
      v = __builtin_bit_cast(u4_t, hv);
    }
#ifdef GI_ABLATION
    if (p.dbg_epi & 1) { *(u4_t*)(p.out + ((o * 2) & 0xFFF0)) = v; continue; }
#endif
    *(u4_t*)(p.out + o * 2) = v;
  }
  if (bwd) {
    // lanes oc, oc + 16, oc + 32, oc + 48 of a wave hold the same channels (CPRO = 16): fold them, then the 8 waves through LDS
    static_assert(DUAL || CPRO == 16 || CPRO == 8, "CPRO");
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int off = CPRO; off < 64; off <<= 1) { bs[e] += __shfl_xor(bs[e], off); bsx[e] += __shfl_xor(bsx[e], off); }
    }
    __syncthreads();      // the staged tile has been read by every thread
    float* fold = (float*)smem;   // [8 waves][BN][2]
    if (lane < CPRO) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { fold[((tid >> 6) * BN + oc * 8 + e) * 2] = bs[e]; fold[((tid >> 6) * BN + oc * 8 + e) * 2 + 1] = bsx[e]; }
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { s += fold[(w * BN + tid) * 2]; q += fold[(w * BN + tid) * 2 + 1]; }
      const int grp = (p.bwd_pg_tiles > 0 && mt_idx >= p.bwd_pg_tiles) ? 1 : 0, rep = (mt_idx + (MODE == 1 ? ph : 0)) & (p.bwd_reps - 1);
      gi_stat_add(p.bwd_acc, p.cout, rep, grp, 0, n0 + tid, s);
      gi_stat_add(p.bwd_acc, p.cout, rep, grp, 1, n0 + tid, q);
    }
  }
}

template <int MODE, int BN>
__global__ void __launch_bounds__(512, 2) igemm5_kernel(KP5 p) {
  constexpr bool DUAL = MODE == 3;                      // both px phases in this workgroup (BN = 2 x 64)
  constexpr bool PH = MODE == 1 || MODE == 3;           // sub-pixel-phase algebra
  constexpr int NQ = MODE == 0 ? 4 : 1;                 // halo groups per channel chunk
  constexpr int NTAP = MODE == 2 ? 9 : 4;               // taps (steps) per halo group
  constexpr int PAD = MODE == 2 ? 2 : 1;                // halo = (TH + PAD) x (TW + PADX)
  constexpr int PADX = DUAL ? 2 : PAD;
  static_assert(!DUAL || BN == 128, "dual-px mode: 2 x 64 columns");
  constexpr int BM = 256, BK = 64, NW = 8;
  constexpr int AJ = MODE == 2 ? 6 : 5;                 // halo pieces (8 rows x 128 B) per wave and group
  constexpr int A_ROWS = AJ * 64, A_BYTES = A_ROWS * 128;   // 40 / 48 KiB per buffer
  constexpr int B_BYTES = BN * 128;
  constexpr int A_OFF = 0, B_OFF = 2 * A_BYTES;
  constexpr int BJ = (BN / 8) / NW;                     // weight-slice pieces per wave and step
  constexpr int WN = BN / 2, MT = 4, NT = WN / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware tile order (as igemm3): the (N tile, phase) blocks of one patch run back to back on one XCD
  const int nyz = p.ntiles * (DUAL ? 2 : (PH ? 4 : 1));
  const int bid = blockIdx.x;
  const int xcd = bid & 7, local = bid >> 3;
  const int mt_idx = (local / nyz) * 8 + xcd;
  if (mt_idx >= p.mtiles) return;
  const int yz = local % nyz;
  const int nt_idx = yz % p.ntiles;
  const int ph = yz / p.ntiles;
  const int py = DUAL ? ph : (ph >> 1), px = DUAL ? 0 : (ph & 1);   // DUAL: px = wn, per wave
  const int n0 = nt_idx * (DUAL ? 64 : BN);
  const int img = mt_idx / p.tiles_per_img, trem = mt_idx % p.tiles_per_img;
  const int y0 = (trem / p.tiles_x) * p.TH, x0 = (trem % p.tiles_x) * p.TW;
  const int HC = p.TW + PADX, HR = p.TH + PAD;
  const int Ktot2 = (PH ? 4 : (MODE == 2 ? 9 : 16)) * p.cin * 2;   // bytes per weight row
  const int64_t phase_bytes = (int64_t)p.cout * Ktot2;             // one sub-pixel phase of the packed weights
  const char* wptr = p.w + (MODE == 1 ? ph * phase_bytes : (DUAL ? (py * 2) * phase_bytes : 0));
  const int Win = 2 * p.Ws, Hin = 2 * p.Hs;             // MODE 0: the large (input) grid

  // ---- per-lane DMA sources: halo rows (fixed for the whole K loop) and weight rows ------------------------
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ (lrow & 7);
  const char* pa[AJ];   // MODE 1: final source (or the zero page); MODE 0: source for parity class (0,0)
  unsigned amask[AJ];   // MODE 0: bit q set = this halo row exists in parity class q
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int r = (wave * AJ + j) * 8 + lrow;
    const int hr = r / HC, hc = r - hr * HC;
    amask[j] = 0;
    if constexpr (MODE != 0) {
      const int iy = y0 + (PH ? py : 0) - 1 + hr, ix = x0 + (MODE == 1 ? px : 0) - 1 + hc;   // DUAL: columns x0-1 .. x0+TW
      const bool ok = hr < HR && iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws;
      pa[j] = ok ? p.in + ((int64_t)((img * p.Hs + iy) * p.Ws + ix) * p.ldin + p.coffin) * 2 + lchunk * 16 : p.zero + lchunk * 16;
    } else {
      // halo (hr,hc) of parity class (qy,qx) is input pixel (2*(y0+hr) - qy, 2*(x0+hc) - qx)
      const int iy0 = 2 * (y0 + hr), ix0 = 2 * (x0 + hc);
      pa[j] = p.in + ((int64_t)((img * Hin + iy0) * Win + ix0) * p.ldin + p.coffin) * 2 + lchunk * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int iy = iy0 - (q >> 1), ix = ix0 - (q & 1);
        if (hr < HR && iy >= 0 && iy < Hin && ix >= 0 && ix < Win) amask[j] |= 1u << q;
      }
    }
  }
  const char* pb[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int rb = (wave * BJ + j) * 8 + lrow;                     // row of the B tile
    if constexpr (DUAL) pb[j] = wptr + (rb >> 6) * phase_bytes + (int64_t)(n0 + (rb & 63)) * Ktot2 + lchunk * 16;   // rows 64.. = px 1
    else pb[j] = wptr + (int64_t)(n0 + rb) * Ktot2 + lchunk * 16;
  }

  // halo piece J of group g = chunk * NQ + q into A buffer `buf`
  auto issue_a = [&](int g, int buf, auto J) {
    constexpr int j = decltype(J)::value;
    char* dst = smem + A_OFF + buf * A_BYTES + (wave * AJ + j) * 1024;
    if constexpr (MODE != 0) {
      glds16(pa[j] + g * (BK * 2), dst);
    } else {
      const int chunk = g >> 2, q = g & 3;
      const int64_t back = (int64_t)((q >> 1) * Win + (q & 1)) * p.ldin * 2;      // wave-uniform
      const char* src = ((amask[j] >> q) & 1u) ? pa[j] - back + chunk * (BK * 2) : p.zero + lchunk * 16;
      glds16(src, dst);
    }
  };
  // weight slice of step s = g * NTAP + tap into ring stage `stage`
  auto issue_b = [&](int s, int stage, auto J) {
    constexpr int j = decltype(J)::value;
    const int g = s / NTAP, tap = s - g * NTAP;
    int koff;
    if constexpr (MODE != 0) {
      koff = tap * p.cin + g * BK;
    } else {
      const int chunk = g >> 2, q = g & 3;
      const int ky = (q >> 1) ? 2 * (tap >> 1) : 2 * (tap >> 1) + 1, kx = (q & 1) ? 2 * (tap & 1) : 2 * (tap & 1) + 1;
      koff = (ky * 4 + kx) * p.cin + chunk * BK;
    }
    glds16(pb[j] + koff * 2, smem + B_OFF + stage * B_BYTES + (wave * BJ + j) * 1024);
  };

  // ---- fragment read addresses -----------------------------------------------------------------------------
  const int lr = lane & 15, lq = lane >> 4;
  const int lgTW = 31 - __builtin_clz(p.TW);
  int rdA[MT][NTAP];   // byte offset inside an A buffer of (pixel row of tile mt, tap), k-half 0; k-half 1 = ^ 64
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = wm * 64 + mt * 16 + lr;
    const int ty_l = m >> lgTW, tx_l = m & (p.TW - 1);
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      const int R = MODE == 1 ? (ty_l + 1 - (tap >> 1)) * HC + (tx_l + 1 - (tap & 1))
                  : MODE == 3 ? (ty_l + 1 - (tap >> 1)) * HC + (tx_l + wn + 1 - (tap & 1))
                  : MODE == 0 ? (ty_l + (tap >> 1)) * HC + (tx_l + (tap & 1))
                              : (ty_l + tap / 3) * HC + (tx_l + tap % 3);
      rdA[mt][tap] = R * 128 + ((lq ^ (R & 7)) << 4);
    }
  }
  const int rdB = (wn * WN + lr) * 128 + ((lq ^ (lr & 7)) << 4);

  f4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  const int ngroups = p.nchunk * NQ, nsteps = ngroups * NTAP;
  const int relu_cend = p.relu_in ? p.relu_cend : 0;
  // halo pieces of the NEXT group issued during tap t of the current one (halo pieces first, then the weight
  // slice of step s+2). Rule: before step s everything but the issues of step s-1 must have landed (B(s) was the
  // last issue of step s-2), see nwait below.
  // 4-tap modes: 2,1,1,1 (measured 3 % faster than 2,2,1,0); 9-tap mode: one piece in each of the first six taps
  auto a_pieces = [](int tap) -> int { return MODE == 2 ? (tap < 6 ? 1 : 0) : (tap == 0 ? 2 : 1); };
  auto a_first = [](int tap) -> int { return MODE == 2 ? tap : (tap == 0 ? 0 : tap + 1); };   // index of a tap's first piece

  // ---- prologue: halo of group 0, weight slices of steps 0 and 1 ---------------------------------------------
  static_for<AJ>([&](auto J) { issue_a(0, 0, J); });
  static_for<BJ>([&](auto J) { issue_b(0, 0, J); });
  if (nsteps > 1) static_for<BJ>([&](auto J) { issue_b(1, 1, J); });

  int stage = 0;   // B ring stage of the current step
  for (int c = 0; c < ngroups; ++c) {      // c: halo group (channel chunk; MODE 0: chunk * 4 + parity class)
    const int abuf = c & 1;
    const bool relu = (c / NQ) * BK < relu_cend;
    const bool next_a = c + 1 < ngroups;
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      const int s = c * NTAP + tap;
      // outstanding issues that may remain: those of step s-1 (B(s) was the last issue of step s-2)
      int nwait;
      if (s == 0) nwait = nsteps > 1 ? BJ : 0;
      else {
        const int sp = s - 1, gp = sp / NTAP, tp = sp - gp * NTAP;
        nwait = ((sp + 2 < nsteps) ? BJ : 0) + ((gp + 1 < ngroups) ? a_pieces(tp) : 0);
        // 4-tap modes: the halo's last piece is the FIRST issue of step s-1 (tap 3), so at tap 0 only that step's
        // weight pieces may still be in flight
        if (MODE != 2 && tap == 0) nwait = (sp + 2 < nsteps) ? BJ : 0;
      }
      wait_vm(nwait);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");

      const char* sa = smem + A_OFF + abuf * A_BYTES;
      const bool more_b = s + 2 < nsteps;
      int st2 = stage + 2;
      if (st2 >= 3) st2 -= 3;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        h8_t af[MT], bf[NT];
        // k-half 1 = logical chunk 4 + lq: the physical chunk of half 0 with bit 2 flipped, i.e. byte offset ^ 64
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *(const h8_t*)(sa + (rdA[mt][tap] ^ (k2 << 6)));
        const char* sbh = smem + B_OFF + stage * B_BYTES + (rdB ^ (k2 << 6));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = *(const h8_t*)(sbh + nt * 2048);
        if (relu) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) af[mt] = relu5(af[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[nt], af[mt], acc[mt][nt], 0, 0, 0);   // D^T: rows = channels
            const int idx = k2 * MT * NT + mt * NT + nt;
            // halo pieces of the next group first, then the weight slice of step s+2
            if (next_a) {
              static_for<AJ>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const int k = j - a_first(tap);            // position of piece j among this tap's pieces
                if (k >= 0 && k < a_pieces(tap) && idx == 2 + 6 * k) issue_a(c + 1, abuf ^ 1, J);
              });
            }
            if (more_b) {
              static_for<BJ>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                if (idx == 14 + q * (MT * NT * 2 - 16) / (BJ > 1 ? BJ : 1)) issue_b(s + 2, st2, Q);
              });
            }
          }
      }
      ++stage;
      if (stage == 3) stage = 0;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  asm volatile("" ::: "memory");
  epilogue5<MODE, BN>(p, acc, smem, tid, lane, wm, wn, mt_idx, nt_idx, ph, py, px, n0, img, y0, x0, lgTW);
}


// =====================================================================================================================
// igemm6: the 4-tap modes (0, 1, 3) of igemm5 with the issue stream put on a diet. The counters of igemm5
// show 3.1 vector instructions per MFMA - address arithmetic of the LDS-DMA pieces and of the fragment reads, the
// k-half XOR, the ring-stage select - on a 16-cycle MFMA that leaves the SIMD 8 issue cycles (two vector
// instructions), and every step opened with all 8 waves waiting for their first fragment reads behind the barrier.
//   * LDS-DMA by `buffer_load_dwordx4 ... lds`: the per-lane part of a piece's source (row, swizzled chunk) is ONE 32-bit
//     offset register for the whole K loop, the moving part (channel chunk, tap) is the scalar offset; padding rows use an
//     out-of-range offset, for which the buffer unit writes zeros (no zero page, no select). No vector instruction per piece.
//   * the weight ring has FOUR stages, so the stage of a step IS its tap (compile-time); the group loop is unrolled over
//     the halo buffer parity (and over the four parity classes in mode 0): every fragment read is `ds_read_b128 v, vaddr
//     offset:imm` with vaddr one of 2 x 16 + 2 loop-invariant registers.
//   * the first k-half of step s+1's fragments is read DURING step s (the pieces of a step land one step earlier: weight
//     slices are issued three steps ahead, the next halo during taps 0 and 1), so after the barrier the MFMAs start at once.
// DBG (builds with -DGI_ABLATION only, wrong results): 1 no LDS-DMA in the loop, 2 no MFMA, 4 no fragment reads, 8 halo pieces
// all from one 4 KiB window (L2-resident), 16 no per-step barrier
template <int MODE, int BN, bool RELU, int DBG = 0>
__global__ void __launch_bounds__(512, 2) igemm6_kernel(KP5 p) {
  static_assert(MODE == 0 || MODE == 1 || MODE == 3, "4-tap modes");
  constexpr bool DUAL = MODE == 3;
  constexpr bool PH = MODE == 1 || MODE == 3;
  constexpr int NQ = MODE == 0 ? 4 : 1;                 // halo groups (parity classes) per channel chunk
  constexpr int NTAP = 4;
  constexpr int PADX = DUAL ? 2 : 1;
  static_assert(!DUAL || BN == 128, "dual-px mode: 2 x 64 columns");
  constexpr int BM = 256, BK = 64, NW = 8;
  constexpr int AJ = 5;                                 // halo pieces (8 rows x 128 B) per wave and group
  constexpr int A_BYTES = AJ * 64 * 128;                // 40 KiB per buffer
  constexpr int B_BYTES = BN * 128;
  constexpr int A_OFF = 0, B_OFF = 2 * A_BYTES;         // + 4 ring stages
  constexpr int BJ = (BN / 8) / NW;                     // weight-slice pieces per wave and step
  constexpr int WN = BN / 2, MT = 4, NT = WN / 16;
  constexpr unsigned OOB = 0x80000000u;                 // beyond any tensor (sizes are checked < 2^31 bytes): reads as zeros
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware tile order (as igemm5)
  const int nyz = p.ntiles * (DUAL ? 2 : (PH ? 4 : 1));
  const int bid = blockIdx.x;
  const int xcd = bid & 7, local = bid >> 3;
  const int mt_idx = (local / nyz) * 8 + xcd;
  if (mt_idx >= p.mtiles) return;
  const int yz = local % nyz;
  const int nt_idx = yz % p.ntiles;
  const int ph = yz / p.ntiles;
  const int py = DUAL ? ph : (ph >> 1), px = DUAL ? 0 : (ph & 1);
  const int n0 = nt_idx * (DUAL ? 64 : BN);
  const int img = mt_idx / p.tiles_per_img, trem = mt_idx % p.tiles_per_img;
  const int y0 = (trem / p.tiles_x) * p.TH, x0 = (trem % p.tiles_x) * p.TW;
  const int HC = p.TW + PADX, HR = p.TH + 1;
  const int Ktot2 = (PH ? 4 : 16) * p.cin * 2;          // bytes per weight row
  const int64_t phase_bytes = (int64_t)p.cout * Ktot2;
  const int Win = 2 * p.Ws, Hin = 2 * p.Hs;             // mode 0: the large (input) grid

  // ---- buffer descriptors: the input tensor, and the weight rows this workgroup reads
  const int64_t in_bytes = (int64_t)p.n * (MODE == 0 ? 4 : 1) * p.Hs * p.Ws * p.ldin * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)in_bytes, 0x00020000);
  const char* wbase = p.w + (MODE == 1 ? ph * phase_bytes : (DUAL ? (py * 2) * phase_bytes : 0));
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, (int)(DUAL ? 2 * phase_bytes : (MODE == 1 ? phase_bytes : phase_bytes)), 0x00020000);

  // ---- per-lane piece offsets (loop invariant) ---------------------------------------------------------------------
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ (lrow & 7);
  unsigned voffA[NQ][AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int r = (wave * AJ + j) * 8 + lrow;
    const int hr = r / HC, hc = r - hr * HC;
    if constexpr (MODE != 0) {
      const int iy = y0 + py - 1 + hr, ix = x0 + (MODE == 1 ? px : 0) - 1 + hc;   // DUAL: columns x0-1 .. x0+TW
      const bool ok = hr < HR && iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws;
      voffA[0][j] = ok ? (unsigned)((((img * p.Hs + iy) * p.Ws + ix) * p.ldin + p.coffin) * 2 + lchunk * 16) : OOB;
      if constexpr ((DBG & 8) != 0) voffA[0][j] &= 0xFF0u;
    } else {
      // halo (hr,hc) of parity class (qy,qx) is input pixel (2*(y0+hr) - qy, 2*(x0+hc) - qx)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int iy = 2 * (y0 + hr) - (q >> 1), ix = 2 * (x0 + hc) - (q & 1);
        const bool ok = hr < HR && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
        voffA[q][j] = ok ? (unsigned)((((img * Hin + iy) * Win + ix) * p.ldin + p.coffin) * 2 + lchunk * 16) : OOB;
      }
    }
  }
  unsigned voffB[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int rb = (wave * BJ + j) * 8 + lrow;                     // row of the B tile
    if constexpr (DUAL) voffB[j] = (unsigned)((rb >> 6) * phase_bytes + (int64_t)(n0 + (rb & 63)) * Ktot2 + lchunk * 16);   // rows 64.. = px 1
    else voffB[j] = (unsigned)((int64_t)(n0 + rb) * Ktot2 + lchunk * 16);
  }

  // ---- fragment read offsets (loop invariant): [k-half][pixel tile][tap] inside an A buffer, [k-half] inside a stage -----
  const int lr = lane & 15, lq = lane >> 4;
  const int lgTW = 31 - __builtin_clz(p.TW);
  int rdA[MT][NTAP];   // k-half 0; k-half 1 (logical chunk 4 + lq) = bit 2 of the physical chunk flipped = offset ^ 64
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = wm * 64 + mt * 16 + lr;
    const int ty_l = m >> lgTW, tx_l = m & (p.TW - 1);
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      const int R = MODE == 1 ? (ty_l + 1 - (tap >> 1)) * HC + (tx_l + 1 - (tap & 1))
                  : MODE == 3 ? (ty_l + 1 - (tap >> 1)) * HC + (tx_l + wn + 1 - (tap & 1))
                              : (ty_l + (tap >> 1)) * HC + (tx_l + (tap & 1));
      rdA[mt][tap] = A_OFF + R * 128 + ((lq ^ (R & 7)) << 4);
    }
  }
  int rdB[2];
  rdB[0] = B_OFF + (wn * WN + lr) * 128 + ((lq ^ (lr & 7)) << 4);
  rdB[1] = rdB[0] ^ 64;

  f4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  const int ngroups = (DBG & 32) ? 2 : p.nchunk * NQ, nsteps = ngroups * NTAP;   // DBG 32: two groups only (fixed costs)
  const int relu_cend = p.relu_in ? p.relu_cend : 0;

  struct Frag { h8_t a[MT], b[NT]; };
  // One fragment read of k-half KH of the step (halo buffer BUF, tap TAP): I < MT = pixel tile I of A, else column tile
  // I - MT of B. The reads of a k-half are spread over the MFMAs of the previous k-half, one every RSTEP MFMAs (a burst
  // of eight b128 reads per wave, from eight waves at once, fills the LDS queue and blocks the MFMA issue behind it; two
  // waves per SIMD reading one fragment per MFMA each already saturate the LDS array), in the order the next k-half
  // needs them: a0, all of b, a1, a2, ...
  constexpr int RSTEP = (MT * NT) / (MT + NT) >= 2 ? 2 : 1;
  auto rorder = [](int k) constexpr -> int { return k == 0 ? 0 : (k <= NT ? MT + k - 1 : k - NT); };
  auto read_one = [&](auto BUF, auto TAP, auto KH, auto Ic, Frag& f) {
    constexpr int buf = decltype(BUF)::value, tap = decltype(TAP)::value, kh = decltype(KH)::value, i = decltype(Ic)::value;
    if constexpr ((DBG & 4) != 0) {
      if constexpr (i < MT) { f.a[i] = h8_t{1, 1, 1, 1, 1, 1, 1, 1}; asm volatile("" : "+v"(f.a[i])); }
      else { f.b[i - MT] = h8_t{1, 1, 1, 1, 1, 1, 1, 1}; asm volatile("" : "+v"(f.b[i - MT])); }
    } else {
      if constexpr (i < MT) f.a[i] = *(const h8_t*)(smem + (rdA[i][tap] ^ (kh << 6)) + buf * A_BYTES);
      else f.b[i - MT] = *(const h8_t*)(smem + rdB[kh] + tap * B_BYTES + (i - MT) * 2048);
    }
  };
  // RELU kernels: the decoder's ReLU on an A fragment right before its first use. rmin = 0 for the skip half of a concat
  // buffer, 0x8000 (the smallest 16-bit integer: no change) for the rest - no branch either way
  auto relu_a = [&](h8_t& v, int rmin) {
    if constexpr (RELU) {
      typedef short s8_t __attribute__((ext_vector_type(8)));
      const short m = (short)rmin;
      const s8_t lo = {m, m, m, m, m, m, m, m};
      v = __builtin_bit_cast(h8_t, __builtin_elementwise_max(__builtin_bit_cast(s8_t, v), lo));
    }
  };
  auto rmin_of = [&](int chunk) -> int { return chunk * BK < relu_cend ? 0 : -32768; };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // ---- prologue: halo of group 0, weight slices of steps 0, 1, 2 (nsteps >= 4) ------------------------------------------
  if constexpr ((DBG & 128) == 0)
  static_for<AJ>([&](auto J) { blds16(rsA, voffA[0][decltype(J)::value], 0, smem + A_OFF + (wave * AJ + decltype(J)::value) * 1024); });
  auto issue_b = [&](unsigned (&vb)[BJ], int chunk, auto Q, auto TAP, auto J) {
    constexpr int q = decltype(Q)::value, tap = decltype(TAP)::value, j = decltype(J)::value;
    int koff;
    if constexpr (MODE != 0) {
      koff = tap * p.cin + chunk * BK;
    } else {
      constexpr int ky = (q >> 1) ? 2 * (tap >> 1) : 2 * (tap >> 1) + 1, kx = (q & 1) ? 2 * (tap & 1) : 2 * (tap & 1) + 1;
      koff = (ky * 4 + kx) * p.cin + chunk * BK;
    }
    blds16(rsB, vb[j], koff * 2, smem + B_OFF + tap * B_BYTES + (wave * BJ + j) * 1024);
  };
  if constexpr ((DBG & 128) == 0) {
  static_for<BJ>([&](auto J) { issue_b(voffB, 0, I0{}, I0{}, J); });
  static_for<BJ>([&](auto J) { issue_b(voffB, 0, I0{}, I1{}, J); });
  static_for<BJ>([&](auto J) { issue_b(voffB, 0, I0{}, I2{}, J); });
  }
  Frag f0;
  wait_vm(BJ);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  static_for<MT + NT>([&](auto Ic) { read_one(I0{}, I0{}, I0{}, Ic, f0); });

  // One halo group = four steps, straight-line code: every step issues the same number of LDS-DMA pieces (beyond the end
  // of the K loop with out-of-range offsets: zeros into stages nobody reads any more), so the counted waits are immediates
  // and there is no branch between the MFMAs. GI = group index modulo 2 (modes 1, 3) or 4 (mode 0): halo buffer GI & 1,
  // parity class GI (mode 0).
  auto group = [&](int c, auto GIc) {
    constexpr int GI = decltype(GIc)::value;
    constexpr int BUF = GI & 1, Q = MODE == 0 ? GI : 0, QN = MODE == 0 ? ((GI + 1) & 3) : 0;
    using BUFT = std::integral_constant<int, BUF>;
    using BUFN = std::integral_constant<int, BUF ^ 1>;
    using QT = std::integral_constant<int, Q>;
    using QNT = std::integral_constant<int, QN>;
    const int chunk = c / NQ, chunk_n = (c + 1) / NQ;            // channel chunk of this group / of the next one
    const int rmin = rmin_of(chunk), rmin_n = rmin_of(chunk_n);
    const bool next_a = c + 1 < ngroups;
    unsigned va[AJ];                                            // the next group's halo pieces (none after the last group)
#pragma unroll
    for (int j = 0; j < AJ; ++j) va[j] = next_a ? voffA[QN][j] : OOB;
    static_for<NTAP>([&](auto TAPc) {
      constexpr int tap = decltype(TAPc)::value;
      const int s = c * NTAP + tap;
      // in flight may stay: the pieces issued during step s-1 = a weight slice (BJ) + its halo pieces (3 in tap 0, 2 in tap 1)
      constexpr int ptap = (tap + 3) & 3;
      constexpr int nwait = BJ + (ptap == 0 ? 3 : (ptap == 1 ? 2 : 0));
      if constexpr (nwait == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if constexpr (nwait == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if constexpr (nwait == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if constexpr (nwait == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      if constexpr ((DBG & 16) == 0) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      Frag f1;
      // weight slice of step s + 3 = (group c, tap 3) during tap 0, else (group c + 1, tap - 1); out of range past the end
      unsigned vb[BJ];
#pragma unroll
      for (int j = 0; j < BJ; ++j) vb[j] = s + 3 < nsteps ? voffB[j] : OOB;
      auto issue_piece = [&](auto IDX) {
        constexpr int idx = decltype(IDX)::value;      // MFMA counter of this step, 0 .. NM - 1
        constexpr int NM = 2 * MT * NT;                // MFMAs per step and wave (32 or 16)
        if constexpr ((DBG & 1) != 0) return;
        // halo of group c+1: pieces 0,1,2 during tap 0, pieces 3,4 during tap 1
        if constexpr (tap == 0) {
          if constexpr (idx == NM / 4) blds16(rsA, va[0], chunk_n * (BK * 2), smem + A_OFF + (BUF ^ 1) * A_BYTES + (wave * AJ + 0) * 1024);
          if constexpr (idx == NM / 4 + 2) blds16(rsA, va[1], chunk_n * (BK * 2), smem + A_OFF + (BUF ^ 1) * A_BYTES + (wave * AJ + 1) * 1024);
          if constexpr (idx == NM / 4 + 4) blds16(rsA, va[2], chunk_n * (BK * 2), smem + A_OFF + (BUF ^ 1) * A_BYTES + (wave * AJ + 2) * 1024);
        } else if constexpr (tap == 1) {
          if constexpr (idx == NM / 4) blds16(rsA, va[3], chunk_n * (BK * 2), smem + A_OFF + (BUF ^ 1) * A_BYTES + (wave * AJ + 3) * 1024);
          if constexpr (idx == NM / 4 + 3) blds16(rsA, va[4], chunk_n * (BK * 2), smem + A_OFF + (BUF ^ 1) * A_BYTES + (wave * AJ + 4) * 1024);
        }
        static_for<BJ>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          if constexpr (idx == NM - 2 - (BJ - 1 - j) * 2) {
            if constexpr (tap == 0) issue_b(vb, chunk, QT{}, I3{}, Jc);
            else issue_b(vb, chunk_n, QNT{}, std::integral_constant<int, tap - 1>{}, Jc);
          }
        });
      };
      // first k-half (fragments f0 are in registers); behind each of its first MT + NT MFMAs one read of the second k-half
      static_for<MT * NT>([&](auto IDX) {
        constexpr int idx = decltype(IDX)::value, mt = idx / NT, nt = idx % NT;
        if constexpr (nt == 0) relu_a(f0.a[mt], rmin);
        if constexpr ((DBG & 2) != 0) { asm volatile("" :: "v"(f0.a[mt])); asm volatile("" :: "v"(f0.b[nt])); }
        else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f0.b[nt], f0.a[mt], acc[mt][nt], 0, 0, 0);   // D^T: rows = channels
        if constexpr (RSTEP * (idx / RSTEP) == idx && idx / RSTEP < MT + NT)
          read_one(BUFT{}, TAPc, I1{}, std::integral_constant<int, rorder(idx / RSTEP)>{}, f1);
        issue_piece(IDX);
        __builtin_amdgcn_sched_barrier(0);
      });
      // second k-half; behind its first MFMAs the first k-half of the NEXT step (that data landed one step early; after
      // the last step: stale bytes, unused)
      static_for<MT * NT>([&](auto IDX) {
        constexpr int idx = decltype(IDX)::value, mt = idx / NT, nt = idx % NT;
        if constexpr (nt == 0) relu_a(f1.a[mt], rmin);
        if constexpr ((DBG & 2) != 0) { asm volatile("" :: "v"(f1.a[mt])); asm volatile("" :: "v"(f1.b[nt])); }
        else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1.b[nt], f1.a[mt], acc[mt][nt], 0, 0, 0);
        // f0's registers are free once the first k-half's MFMAs have issued: a[i] after MFMA (i, NT-1), b[j] after (MT-1, j)
        if constexpr (RSTEP * (idx / RSTEP) == idx && idx / RSTEP < MT + NT) {
          using RI = std::integral_constant<int, rorder(idx / RSTEP)>;
          if constexpr (tap < 3) read_one(BUFT{}, std::integral_constant<int, tap + 1>{}, I0{}, RI{}, f0);
          else read_one(BUFN{}, I0{}, I0{}, RI{}, f0);
        }
        issue_piece(std::integral_constant<int, idx + MT * NT>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  if constexpr (MODE == 0) {
    for (int c = 0; c < ngroups; c += 4) { group(c, I0{}); group(c + 1, I1{}); group(c + 2, I2{}); group(c + 3, I3{}); }
  } else {
    int c = 0;
    for (; c + 1 < ngroups; c += 2) { group(c, I0{}); group(c + 1, I1{}); }
    if (c < ngroups) group(c, I0{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  asm volatile("" ::: "memory");
  if constexpr ((DBG & 64) != 0) {   // no epilogue: keep the accumulators alive with a store that never happens
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 12345.678f) p.out[0] = 1;
    return;
  }
  epilogue5<MODE, BN>(p, acc, smem, tid, lane, wm, wn, mt_idx, nt_idx, ph, py, px, n0, img, y0, x0, lgTW);
}

}  // namespace

// mode 1 (sub-pixel phases) and mode 0 (stride-2 gather). Returns GI_ERR_UNSUPPORTED for shapes it does not serve
// (the caller falls back to igemm3).
int op_igemm5(hipStream_t st, int mode, IgemmArgs& a) {
  if (mode < 0 || mode > 2) return GI_ERR_UNSUPPORTED;
  if (a.cin % 64 != 0 || a.cout % 64 != 0 || a.cin > 2048) return GI_ERR_UNSUPPORTED;
  if (!gi_is_pow2(a.Ws) || a.Ws < 8) return GI_ERR_UNSUPPORTED;
  const int TW = a.Ws < 32 ? a.Ws : 32, TH = 256 / TW;
  if (a.Hs % TH != 0) return GI_ERR_UNSUPPORTED;
  if (mode == 2 ? (TH + 2) * (TW + 2) > 384 : (TH + 1) * (TW + ((mode == 1 && a.cout % 128 != 0) ? 2 : 1)) > 320) return GI_ERR_UNSUPPORTED;
  int BN = (a.cout % 128 == 0) ? 128 : 64;
  const int nph = mode == 1 ? 4 : 1;
  const int tiles_x = a.Ws / TW, tiles_per_img = tiles_x * (a.Hs / TH);
  const int mtiles = a.n * tiles_per_img;
  if (mode != 2 && mtiles * (a.cout / BN) * nph < 128) return GI_ERR_UNSUPPORTED;
  if (mode == 0 && BN == 128 && mtiles * (a.cout / BN) < 256) {
    // 128..255 workgroups on 256 CUs (generator d4 at 256x256, bs=32): 64-wide N tiles double them
    const int narrow = gi_tune("GI_IGEMM5_NARROW", 1);
    if (narrow) BN = 64;
  }
  const bool dual = mode == 1 && BN == 64;     // 64-channel N tiles: both px phases per workgroup (MODE 3)
  int dev = 0;
  GI_HIP(hipGetDevice(&dev));
  const char* zero = gi_igemm3_zero_page(dev);
  if (!zero) return GI_ERR_HIP;
  const int64_t in_px = (int64_t)a.n * a.Hs * a.Ws * (mode == 0 ? 4 : 1), out_px = (int64_t)a.n * a.Hs * a.Ws * (mode == 1 ? 4 : 1);
  GI_REQUIRE(in_px * a.ldin < (1ll << 31) && out_px * a.ldout < (1ll << 31), "igemm5: tensor too large for 32-bit offsets");
  KP5 kp;
  kp.in = (const char*)a.in; kp.w = (const char*)a.w; kp.out = (char*)a.out; kp.zero = zero;
  kp.bias = a.bias; kp.partials = a.stat_acc ? nullptr : a.partials;
  kp.stat_acc = a.stat_acc; kp.stat_pg = a.stat_pg; kp.stat_reps = a.stat_reps > 0 ? a.stat_reps : 1;
  a.stat_used = a.stat_acc ? 1 : 0;
  GI_REQUIRE(!a.stat_acc || a.stat_pg == 0 || a.stat_pg % (a.Hs * a.Ws) == 0, "igemm5: stat_pg=%d must be whole images", a.stat_pg);
  kp.Hs = a.Hs; kp.Ws = a.Ws; kp.n = a.n; kp.TH = TH; kp.TW = TW;
  kp.tiles_x = tiles_x; kp.tiles_per_img = tiles_per_img; kp.mtiles = mtiles;
  kp.cin = a.cin; kp.ldin = a.ldin; kp.coffin = a.coffin;
  kp.cout = a.cout; kp.ldout = a.ldout; kp.coffout = a.coffout;
  kp.nchunk = a.cin / 64;
  kp.relu_in = a.relu_in; kp.act_out = a.act_out;
  kp.relu_cend = a.relu_cend > 0 ? a.relu_cend : a.cin;
  kp.mask = (const char*)a.mask; kp.ldmask = a.ldmask; kp.coffmask = a.coffmask; kp.mask_slope = a.mask_slope;
  kp.add = a.mask ? (const char*)a.add : nullptr; kp.ldadd = a.ldadd; kp.coffadd = a.coffadd;
  kp.mask_bits = nullptr;
  kp.c1w_img = nullptr; kp.c1w_part = nullptr; kp.c1w_scale = 0.f; kp.c1w_skip_out = 0;
  a.c1w_applied = 0; a.c1w_blocks = 0;
  kp.dbg_epi = 0;
  kp.pool = 0;
  a.pool_applied = 0;
#ifdef GI_ABLATION
  { const char* e = getenv("GI_EPI_DBG"); if (e) kp.dbg_epi = atoi(e); }
#endif
  kp.bwd_acc = nullptr; kp.bwd_c0 = 0; kp.bwd_c = a.cout;
  if (a.mask) {
    GI_REQUIRE(a.ldmask % 8 == 0 && a.coffmask % 8 == 0 && out_px * a.ldmask < (1ll << 31), "igemm5: mask layout");
    GI_REQUIRE(!a.add || (a.ldadd % 8 == 0 && a.coffadd % 8 == 0 && out_px * a.ldadd < (1ll << 31)), "igemm5: add layout");
    a.mask_applied = 1;
  }
  kp.ntiles = a.cout / BN;
  const int nyz = kp.ntiles * (dual ? 2 : nph);
  const int grid = ((mtiles + 7) / 8) * 8 * nyz;
  const int BNk = dual ? 128 : BN;             // columns of the workgroup tile
  const int ring = 2 * (mode == 2 ? 384 : 320) * 128 + 3 * BNk * 128, epi = 256 * (BNk + 8) * 2 + 4 * BNk * 8;
  const int LDS = ring > epi ? ring : epi;
  static GiDevOnce attr[6];
  const void* fn[6] = {(const void*)igemm5_kernel<0, 128>, (const void*)igemm5_kernel<1, 128>, (const void*)igemm5_kernel<2, 128>,
                       (const void*)igemm5_kernel<0, 64>,  (const void*)igemm5_kernel<1, 64>,  (const void*)igemm5_kernel<2, 64>};
  // igemm8 (igemm8.hip): the same tile on four waves, two workgroups per CU. GI_IGEMM8: 0 off, 1 (default) layers whose grid
  // gives every CU at least two workgroups (with one per CU half the wave slots stay empty: measured d3 / u4 / critic conv4,
  // 256 workgroups, 10 - 16 % slower than igemm6; every layer with >= 512 workgroups 3 - 16 % faster), 2 every eligible layer
  // GI_IGEMM6=0 (the first-generation halo kernels: no buffer-descriptor LDS-DMA anywhere) switches igemm8 off as well
  const int use8 = gi_opt(GI_OPT_IGEMM6) ? gi_opt(GI_OPT_IGEMM8) : 0;
  // (the 3x3 mode: 128-column tiles on 32-wide patches, no fused input ReLU; VGG-19 from conv2_1 to conv4_4)
  const bool take8 = use8 && (mode != 2 || (TW == 32 && !a.relu_in)) && (dual || BN == 128 || mode == 2) && a.cin % (mode == 0 ? 64 : 32) == 0 && TW >= 16 && in_px * a.ldin * 2 < (1ll << 31) &&
      (int64_t)a.cout * (mode == 1 ? 4 : (mode == 2 ? 9 : 16)) * a.cin * 2 * (dual ? 2 : 1) < (1ll << 31) && !(mode == 0 && a.relu_in) &&
      (use8 >= 2 || grid >= gi_tune("GI_IGEMM8_MINGRID", 512));
  // fused BatchNorm-backward reduction (the dual-px / 64-column tiles do not take it; nor a launch with a mask; a column range only igemm8)
  const bool bwd_range = a.bwd_c > 0 && (a.bwd_c0 != 0 || a.bwd_c != a.cout);
  if (a.bwd_acc && !a.mask && mode != 2 && a.cout % 128 == 0 && BN == 128 && (!bwd_range || (take8 && a.bwd_c0 % 128 == 0 && a.bwd_c % 128 == 0 && a.bwd_c0 + a.bwd_c <= a.cout))) {
    const int64_t px_per_tile = 256 * (mode == 1 ? 4 : 1);          // output pixels per M tile over all phases
    GI_REQUIRE(a.bwd_ldx % 8 == 0 && out_px * a.bwd_ldx < (1ll << 31) && (a.bwd_pg == 0 || a.bwd_pg % px_per_tile == 0) && a.coffout == 0,
               "igemm5: fused BatchNorm-backward reduction: layout");
    kp.bwd_x = (const char*)a.bwd_x; kp.bwd_ldx = a.bwd_ldx;
    kp.bwd_scale = a.bwd_scale; kp.bwd_shift = a.bwd_shift; kp.bwd_mean = a.bwd_mean; kp.bwd_inv = a.bwd_inv; kp.bwd_stride = a.bwd_stride;
    kp.bwd_slope = a.bwd_slope; kp.bwd_acc = a.bwd_acc; kp.bwd_reps = a.bwd_reps > 0 ? a.bwd_reps : 1;
    kp.bwd_pg_tiles = a.bwd_pg > 0 ? (int)(a.bwd_pg / px_per_tile) : 0;
    if (bwd_range) { kp.bwd_c0 = a.bwd_c0; kp.bwd_c = a.bwd_c; }
    a.bwd_applied = 1;
  }
  if (take8) {
    if (mode == 2 && a.pool2 && !a.mask && !a.stat_acc && !a.partials) {   // the pooled store: igemm8's 3x3 mode only
      GI_REQUIRE(a.coffout == 0 && (int64_t)a.n * (a.Hs / 2) * (a.Ws / 2) * a.ldout < (1ll << 31), "igemm8: pooled output layout");
      kp.pool = 1;
      a.pool_applied = 1;
    }
    if (dual && !a.relu_in && a.mask && a.mask_bits && a.cout == 64 && !a.bias && a.act_out == GI_ACT_NONE && !a.stat_acc && !a.partials) kp.mask_bits = a.mask_bits;
    if (kp.mask_bits && a.c1w_part && a.c1w_img && mtiles % 8 == 0 && a.c1w_part_floats >= (int64_t)grid * 1024 &&
        (int64_t)a.n * 16 * a.Hs * a.Ws < (1ll << 31)) {
      kp.c1w_img = a.c1w_img; kp.c1w_part = a.c1w_part; kp.c1w_scale = a.c1w_scale; kp.c1w_skip_out = a.c1w_skip_out;
      a.c1w_applied = 1; a.c1w_blocks = grid;
    }
    GI_TRY(op_igemm8_launch(st, mode, dual, a.relu_in != 0, grid, kp, BN));
    a.ntiles_out = mtiles * nph;
    return GI_OK;
  }
  const int use6 = gi_opt(GI_OPT_IGEMM6);   // GI_IGEMM6=0: the first-generation halo kernels (also the fallback beyond 2^31-byte tensors)
  if (use6 && mode != 2 && in_px * a.ldin * 2 < (1ll << 31) && (int64_t)a.cout * (mode == 1 ? 4 : 16) * a.cin * 2 * (dual ? 2 : 1) < (1ll << 31) &&
      !(mode == 0 && a.relu_in)) {
    const int lds6 = 2 * 320 * 128 + 4 * BNk * 128;
    const int lds = lds6 > epi ? lds6 : epi;
    const int v6 = (dual ? 4 : (BN == 64 ? 2 : 0) + mode) * 2 + (a.relu_in ? 1 : 0);
    static GiDevOnce attr6[10];
#define GI_K6(MODE_, BN_, RELU_) do { \
      if (attr6[v6].first()) { GI_HIP(hipFuncSetAttribute((const void*)igemm6_kernel<MODE_, BN_, RELU_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); } \
      hipLaunchKernelGGL((igemm6_kernel<MODE_, BN_, RELU_>), dim3(grid), dim3(512), lds, st, kp); } while (0)
#ifdef GI_ABLATION   // timing-only ablation kernels compute wrong results: compiled only with `build.sh -DGI_ABLATION`
    { const char* e = getenv("GI_IGEMM6_DBG"); const int dbg = e ? atoi(e) : 0;
      if (dbg && v6 == 3) {
#define GI_K6D(D_) do { GI_HIP(hipFuncSetAttribute((const void*)igemm6_kernel<1, 128, true, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((igemm6_kernel<1, 128, true, D_>), dim3(grid), dim3(512), lds, st, kp); } while (0)
        switch (dbg) {
          case 1: GI_K6D(1); break; case 2: GI_K6D(2); break; case 4: GI_K6D(4); break; case 8: GI_K6D(8); break;
          case 16: GI_K6D(16); break; case 3: GI_K6D(3); break; case 5: GI_K6D(5); break; case 6: GI_K6D(6); break;
          case 7: GI_K6D(7); break; case 23: GI_K6D(23); break; case 32: GI_K6D(32); break; case 39: GI_K6D(39); break;
          case 64: GI_K6D(64); break; case 96: GI_K6D(96); break; case 225: GI_K6D(225); break; case 231: GI_K6D(231); break; default: GI_K6D(9); break;
        }
#undef GI_K6D
        GI_LAUNCH_CHECK();
        a.ntiles_out = mtiles * nph;
        return GI_OK;
      } }
#endif
    switch (v6) {
      case 0: GI_K6(0, 128, false); gi_note_kernel("igemm6<0,128>"); break;
      case 2: GI_K6(1, 128, false); gi_note_kernel("igemm6<1,128>"); break;
      case 3: GI_K6(1, 128, true); gi_note_kernel("igemm6<1,128,relu>"); break;
      case 4: GI_K6(0, 64, false); gi_note_kernel("igemm6<0,64>"); break;
      case 6: GI_K6(1, 64, false); gi_note_kernel("igemm6<1,64>"); break;
      case 7: GI_K6(1, 64, true); gi_note_kernel("igemm6<1,64,relu>"); break;
      case 8: GI_K6(3, 128, false); gi_note_kernel("igemm6<3,128>"); break;
      default: GI_K6(3, 128, true); gi_note_kernel("igemm6<3,128,relu>"); break;
    }
#undef GI_K6
    GI_LAUNCH_CHECK();
    a.ntiles_out = mtiles * nph;
    return GI_OK;
  }
  if (dual) {
    static GiDevOnce attr_dual;
    if (attr_dual.first()) { GI_HIP(hipFuncSetAttribute((const void*)igemm5_kernel<3, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
    hipLaunchKernelGGL((igemm5_kernel<3, 128>), dim3(grid), dim3(512), LDS, st, kp);
    gi_note_kernel("igemm5<3,128>");
    GI_LAUNCH_CHECK();
    a.ntiles_out = mtiles * nph;
    return GI_OK;
  }
  const int vi = (BN == 64 ? 3 : 0) + mode;
  if (attr[vi].first()) { GI_HIP(hipFuncSetAttribute(fn[vi], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
  switch (vi) {
    case 0: hipLaunchKernelGGL((igemm5_kernel<0, 128>), dim3(grid), dim3(512), LDS, st, kp); break;
    case 1: hipLaunchKernelGGL((igemm5_kernel<1, 128>), dim3(grid), dim3(512), LDS, st, kp); break;
    case 2: hipLaunchKernelGGL((igemm5_kernel<2, 128>), dim3(grid), dim3(512), LDS, st, kp); break;
    case 3: hipLaunchKernelGGL((igemm5_kernel<0, 64>), dim3(grid), dim3(512), LDS, st, kp); break;
    case 4: hipLaunchKernelGGL((igemm5_kernel<1, 64>), dim3(grid), dim3(512), LDS, st, kp); break;
    default: hipLaunchKernelGGL((igemm5_kernel<2, 64>), dim3(grid), dim3(512), LDS, st, kp); break;
  }
  { static const char* nm[6] = {"igemm5<0,128>", "igemm5<1,128>", "igemm5<2,128>", "igemm5<0,64>", "igemm5<1,64>", "igemm5<2,64>"}; gi_note_kernel(nm[vi]); }
  GI_LAUNCH_CHECK();
  a.ntiles_out = mtiles * nph;
  return GI_OK;
}
