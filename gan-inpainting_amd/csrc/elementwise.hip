// HBM-bound elementwise and reduction kernels: BatchNorm2d (statistics finalize / apply / backward),
// activation masks, dropout, mask compositing, reconstruction and adversarial losses, fused
// optimizers, gradient statistics, weight packing. All reductions are two-stage and deterministic
// (per-block partials + fixed-order final sum) except where an accumulate-into-gradient semantic
// makes a float atomic the natural form.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "stat_acc.h"
#include "bn_acc.h"

namespace {

constexpr int kMaxBlocks = 2048;

template <typename T, int EPC>
__device__ __forceinline__ void load_vec(const char* base, int64_t elem_off, float (&v)[EPC]) {
  if constexpr (std::is_same<T, half_t>::value) {
    const h8_t h = *(const h8_t*)(base + elem_off * 2);
#pragma unroll
    for (int e = 0; e < EPC; ++e) v[e] = (float)h[e];
  } else {
    const f4_t f = *(const f4_t*)(base + elem_off * 4);
#pragma unroll
    for (int e = 0; e < EPC; ++e) v[e] = f[e];
  }
}
template <int EPC>
__device__ __forceinline__ void load_f32(const float* src, float (&v)[EPC]) {   // EPC consecutive floats, 16-byte aligned
#pragma unroll
  for (int q = 0; q < EPC / 4; ++q) {
    const f4_t f = *(const f4_t*)(src + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * q + e] = f[e];
  }
}
template <typename T, int EPC>
__device__ __forceinline__ void store_vec(char* base, int64_t elem_off, const float (&v)[EPC]) {
  if constexpr (std::is_same<T, half_t>::value) {
    h8_t h;
#pragma unroll
    for (int e = 0; e < EPC; ++e) h[e] = (half_t)v[e];
    *(h8_t*)(base + elem_off * 2) = h;
  } else {
    *(f4_t*)(base + elem_off * 4) = f4_t{v[0], v[1], v[2], v[3]};
  }
}

// two block-wide sums at once (256 threads): wave shuffles, then one LDS exchange of the four wave totals.
// Fixed order, so results repeat. sh: 8 doubles.
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh) {
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
  __syncthreads();   // sh may still be read from a previous call
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = a; sh[4 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  a = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  b = (sh[4] + sh[5]) + (sh[6] + sh[7]);
}

// ---- BatchNorm statistics --------------------------------------------------------------------
// partials [rows][2][c] -> per-channel scale/shift (+ running-stat update in train mode).
// nn.BatchNorm2d semantics (lib/models/networks.py:38): biased variance for normalisation,
// unbiased for running_var, momentum 0.1, eps 1e-5.
__global__ void __launch_bounds__(256) bn_finalize_kernel(const float* partials, int rows, int c, float count,
                                                          const float* gamma, const float* beta, float* rmean,
                                                          float* rvar, float* scale, float* shift, float* smean,
                                                          float* sinv, int train, float momentum, float eps, int groups,
                                                          int64_t part_stride, int out_stride) {
  // groups > 1: consecutive BatchNorm populations (stacked critic batch); group j's partial rows start at
  // partials + j*part_stride, its outputs at +j*out_stride; running statistics are updated in group order
  __shared__ double sh[8];
  const int ch = blockIdx.x;
  for (int j = 0; j < groups; ++j) {
    float mean = 0.f, var = 1.f;
    double s = 0.0, q = 0.0;
    if (train) {
      const float* pj = partials + (int64_t)j * part_stride;
      for (int r = threadIdx.x; r < rows; r += 256) {
        s += (double)pj[((int64_t)r * 2) * c + ch];
        q += (double)pj[((int64_t)r * 2 + 1) * c + ch];
      }
      block_sum2(s, q, sh);
    }
    if (threadIdx.x == 0) {
      if (train) {
        const double m = s / count;
        double v = q / count - m * m;
        if (v < 0.0) v = 0.0;
        mean = (float)m;
        var = (float)v;
        const float unbiased = count > 1.f ? (float)(v * (double)count / ((double)count - 1.0)) : var;
        rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * mean;
        rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * unbiased;
      } else {
        mean = rmean[ch];
        var = rvar[ch];
      }
      const float inv = 1.0f / sqrtf(var + eps);
      const float sc = gamma[ch] * inv;
      const int o = j * out_stride + ch;
      scale[o] = sc;
      shift[o] = beta[ch] - mean * sc;
      smean[o] = mean;
      sinv[o] = inv;
    }
  }
}

// stand-alone form (consumers that apply the affine map themselves: C1Affine, HeadArgs::scale4)
__global__ void __launch_bounds__(256) bn_finalize_acc_kernel(BnAccP a, int c) {
  if (blockIdx.x == 0 && a.zero_next) zero_words64(a.zero_next, a.zero_words);
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= c) return;
  float sc, sh;
  for (int j = 0; j < a.groups; ++j) bn_from_acc(a, c, ch, j, true, sc, sh);
}

// grid-stride is a multiple of the chunks-per-pixel (a power of two <= 256), so a thread's channel
// chunk never changes: scale/shift live in registers, index math is shifts.
// FUSED: scale / shift come from the exact accumulators (every block derives the vectors of all channels into LDS,
// block 0 also publishes them and updates the running statistics): no finalize launch between GEMM and this pass.
// GEN: the dropout keep-mask is drawn here (and stored for the backward) instead of by a separate fill launch.
// G2: two BatchNorm populations in one pass, pixels >= pg use the second scale/shift set (at +gstride floats).
template <typename T, bool G2, bool FUSED>
__global__ void __launch_bounds__(256) bn_apply_kernel(const char* x, char* y, int64_t pixels, int c, int ldy, int coffy,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int act, uint8_t* drop, float drop_scale, int64_t pg, int gstride,
                                                       BnAccP fa, uint64_t drop_seed, uint32_t drop_thresh,
                                                       const u4_t* side_src, u4_t* side_dst, int64_t side_chunks) {
  constexpr int EPC = 16 / (int)sizeof(T);
  // side copy (op_bn_apply_acc): this thread's 16-byte chunks are requested first and stored after the pass's own first loads
  u4_t sc_[2] = {u4_t{0u, 0u, 0u, 0u}, u4_t{0u, 0u, 0u, 0u}};
  const int64_t sid = (int64_t)blockIdx.x * 256 + threadIdx.x, sstride = (int64_t)gridDim.x * 256;
  if (side_chunks > 0) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (sid + u * sstride < side_chunks) sc_[u] = side_src[sid + u * sstride];
  }
  const int cpp = c / EPC;
  const int lg = 31 - __builtin_clz(cpp);
  const int64_t total = pixels * cpp;
  const int64_t gid0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cc = (int)(gid0 & (cpp - 1));
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t esz = (int64_t)sizeof(T);
  // the pass is HBM-bound: a thread's first four 16-byte loads are issued BEFORE the scale / shift vectors are derived (FUSED:
  // a chain of dependent accumulator / parameter loads of 2 - 3 us during which every co-resident block would otherwise
  // request nothing), and from then on the next four are in flight while the current four are finished
  int64_t gid = gid0;
  bool full = gid + 3 * stride < total;
  u4_t r[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) r[u] = full ? *(const u4_t*)(x + (((gid + u * stride) >> lg) * c + cc * EPC) * esz) : u4_t{0u, 0u, 0u, 0u};
  if (side_chunks > 0) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (sid + u * sstride < side_chunks) side_dst[sid + u * sstride] = sc_[u];
    for (int64_t i = sid + 2 * sstride; i < side_chunks; i += sstride) side_dst[i] = side_src[i];
  }
  float sc[EPC], sh[EPC], sc1[G2 ? EPC : 1], sh1[G2 ? EPC : 1];
  if constexpr (FUSED) {
    extern __shared__ __attribute__((aligned(16))) float aff[];   // [groups][2][c]
    for (int ch = threadIdx.x; ch < c; ch += 256)
      for (int j = 0; j < fa.groups; ++j) {
        float a, b;
        bn_from_acc(fa, c, ch, j, blockIdx.x == 0, a, b);
        aff[(j * 2) * c + ch] = a;
        aff[(j * 2 + 1) * c + ch] = b;
      }
    if (blockIdx.x == 0 && fa.zero_next) zero_words64(fa.zero_next, fa.zero_words);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = aff[cc * EPC + e]; sh[e] = aff[c + cc * EPC + e]; }
    if constexpr (G2) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { sc1[e] = aff[2 * c + cc * EPC + e]; sh1[e] = aff[3 * c + cc * EPC + e]; }
    }
  } else {
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = scale ? scale[cc * EPC + e] : 1.f; sh[e] = scale ? shift[cc * EPC + e] : 0.f; }
    if constexpr (G2) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { sc1[e] = scale[gstride + cc * EPC + e]; sh1[e] = shift[gstride + cc * EPC + e]; }
    }
  }
  // one 16-byte chunk: affine map + activation (+ dropout) + store. The activation and the dropout mode are the same for every
  // element of a launch: the streaming loops below run inside gi_with_act with both as compile-time constants (written as run-time
  // tests inside `finish`, hipcc kept them as scalar branches per ELEMENT: 299 branches in this kernel's 3000 instructions)
  gi_with_act(act, [&](auto ACTc) {
  constexpr int ACT = decltype(ACTc)::value;
  auto body = [&](auto DROPc) {
  constexpr int DROP = decltype(DROPc)::value;   // 0 none, 1 read the keep-mask, 2 draw it here
  auto finish = [&](int64_t pix, u4_t raw) {
    float v[EPC];
    if constexpr (std::is_same<T, half_t>::value) {
      const h8_t h = __builtin_bit_cast(h8_t, raw);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] = (float)h[e];
    } else {
      const f4_t f = __builtin_bit_cast(f4_t, raw);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] = f[e];
    }
    const bool second = G2 && pix >= pg;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float t = G2 ? fmaf(v[e], second ? sc1[e] : sc[e], second ? sh1[e] : sh[e]) : fmaf(v[e], sc[e], sh[e]);
      v[e] = gi_act_c<ACT>(t);
    }
    if constexpr (DROP != 0) {
      const int64_t e0 = pix * c + cc * EPC;
      if constexpr (DROP == 2) {     // draw the keep-mask here; the backward reads it back
        uint8_t k[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) k[e] = dropout_keep(drop_seed, e0 + e, drop_thresh);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = k[e] ? v[e] * drop_scale : 0.f;
        if constexpr (EPC == 8) {
          *(uint2*)(drop + e0) = uint2{(unsigned)k[0] | ((unsigned)k[1] << 8) | ((unsigned)k[2] << 16) | ((unsigned)k[3] << 24),
                                       (unsigned)k[4] | ((unsigned)k[5] << 8) | ((unsigned)k[6] << 16) | ((unsigned)k[7] << 24)};
        } else {
          *(unsigned*)(drop + e0) = (unsigned)k[0] | ((unsigned)k[1] << 8) | ((unsigned)k[2] << 16) | ((unsigned)k[3] << 24);
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = drop[e0 + e] ? v[e] * drop_scale : 0.f;
      }
    }
    store_vec<T, EPC>(y, pix * ldy + coffy + cc * EPC, v);
  };
  while (full) {
    const int64_t gn = gid + 4 * stride;
    const bool nfull = gn + 3 * stride < total;
    u4_t rn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) rn[u] = nfull ? *(const u4_t*)(x + (((gn + u * stride) >> lg) * c + cc * EPC) * esz) : u4_t{0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < 4; ++u) finish((gid + u * stride) >> lg, r[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = rn[u];
    gid = gn;
    full = nfull;
  }
  for (; gid < total; gid += stride) finish(gid >> lg, *(const u4_t*)(x + ((gid >> lg) * c + cc * EPC) * esz));
  };   // body
  if (!drop) body(std::integral_constant<int, 0>{});
  else if (!drop_thresh) body(std::integral_constant<int, 1>{});
  else body(std::integral_constant<int, 2>{});
  });  // gi_with_act
}

// column sum / sumsq of a dense (pixels,c) tensor -> partials [blocks][2][c]
template <typename T>
__global__ void __launch_bounds__(256) col_stats_kernel(const char* x, int64_t pixels, int c, float* partials,
                                                        int rows_per_block) {
  constexpr int EPC = 16 / (int)sizeof(T);
  __shared__ float red[2 * 256 * EPC];
  const int Q = c / EPC, RL = 256 / Q;
  const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(pixels, r0 + rows_per_block);
  float s[EPC], sq[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) s[e] = sq[e] = 0.f;
  for (int64_t r = r0 + rl; r < r1; r += RL) {
    float v[EPC];
    load_vec<T, EPC>(x, r * c + q * EPC, v);
#pragma unroll
    for (int e = 0; e < EPC; ++e) { s[e] += v[e]; sq[e] = fmaf(v[e], v[e], sq[e]); }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) { red[threadIdx.x * EPC + e] = s[e]; red[256 * EPC + threadIdx.x * EPC + e] = sq[e]; }
  __syncthreads();
  if (rl == 0) {
    for (int i = 1; i < RL; ++i)
#pragma unroll
      for (int e = 0; e < EPC; ++e) { s[e] += red[(i * Q + q) * EPC + e]; sq[e] += red[256 * EPC + (i * Q + q) * EPC + e]; }
    float* ps = partials + ((int64_t)blockIdx.x * 2) * c + q * EPC;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { ps[e] = s[e]; ps[c + e] = sq[e]; }
  }
}

// ---- backward through [dropout] -> activation -> [BatchNorm] ---------------------------------
struct BwdP {
  const char* g1; int ldg1, coffg1;
  const char* g2; int ldg2, coffg2;
  const char* y; int ldy, coffy;
  const char* x;
  char* dx;
  int64_t pixels; int c; int act; float drop_scale;
  const float* gamma; const float* mean; const float* inv;
  float* partials; const float* sums;
  int rows_per_block;
  int64_t pg;        // pixels per BatchNorm population (= pixels for one group); blocks never straddle populations
  int stat_stride;   // floats between the populations' mean / inv vectors; sums are [group][2][c]
  const float* scale; const float* shift;   // non-null: the activation's sign comes from fma(x, scale, shift), the
                                            // forward's own pre-activation value, instead of a read of y
  // exact-accumulator form (stat_acc.h): the reduce pass adds {sum dz, sum dz*xhat} into acc[group][c][4] and the apply
  // pass derives its coefficients from them itself (no sums launch). acc null in the apply pass = running-statistics
  // BatchNorm (both sums zero). Block 0 of the apply pass accumulates dgamma / dbeta and clears zero_next.
  unsigned long long* acc; int acc_reps;
  float* dgamma; float* dbeta; float inv_loss_scale; float invM;
  unsigned long long* zero_next; int zero_words;
};

// The upstream gradients of one 16-byte chunk: g1 (unmasked) and g2 (passes the parent's ReLU, masked by [y > 0]).
// All loads are issued before any arithmetic (with_y: also the saved activation y).
template <typename T, int EPC>
struct DzIn { float g1[EPC], g2[EPC], y[EPC]; };
template <typename T, int EPC>
__device__ __forceinline__ void load_dz(const BwdP& p, int64_t pix, int ch0, bool with_y, DzIn<T, EPC>& in) {
  if (p.g1) load_vec<T, EPC>(p.g1, pix * p.ldg1 + p.coffg1 + ch0, in.g1);
  if (p.g2) load_vec<T, EPC>(p.g2, pix * p.ldg2 + p.coffg2 + ch0, in.g2);
  if (with_y) load_vec<T, EPC>(p.y, pix * p.ldy + p.coffy + ch0, in.y);
}
// The same loads as raw 16-byte chunks, so that a thread can keep the loads of SEVERAL pixels in flight before it converts
// and computes anything (these passes are HBM-bound; with one pixel per iteration a lane has 32 - 48 bytes outstanding and
// the chip 4 - 6 MB, i.e. 2 TB/s at the ~2 us a loaded memory system takes to answer).
struct RawIn { u4_t x, g1, g2, y; };
template <typename T>
__device__ __forceinline__ void load_raw(const BwdP& p, int64_t pix, int ch0, bool with_x, bool with_y, RawIn& r) {
  constexpr int64_t ES = (int64_t)sizeof(T);
  if (with_x) r.x = *(const u4_t*)(p.x + (pix * p.c + ch0) * ES);
  if (p.g1) r.g1 = *(const u4_t*)(p.g1 + (pix * p.ldg1 + p.coffg1 + ch0) * ES);
  if (p.g2) r.g2 = *(const u4_t*)(p.g2 + (pix * p.ldg2 + p.coffg2 + ch0) * ES);
  if (with_y) r.y = *(const u4_t*)(p.y + (pix * p.ldy + p.coffy + ch0) * ES);
}
template <typename T, int EPC>
__device__ __forceinline__ void cvt_raw(u4_t raw, float (&v)[EPC]) {
  if constexpr (std::is_same<T, half_t>::value) {
    const h8_t h = __builtin_bit_cast(h8_t, raw);
#pragma unroll
    for (int e = 0; e < EPC; ++e) v[e] = (float)h[e];
  } else {
    const f4_t f = __builtin_bit_cast(f4_t, raw);
#pragma unroll
    for (int e = 0; e < EPC; ++e) v[e] = f[e];
  }
}
// the loads of one chunk with the input combination known at compile time (IN: 1 = g1, 2 = g2, 3 = both; LOAD_Y: the saved
// activation): fields that are not loaded cost no registers
template <typename T, int IN, bool LOAD_Y>
struct RawB { u4_t x, g1, g2, y; };
template <typename T, int EPC>
__device__ __forceinline__ void unpack_raw(const BwdP& p, const RawIn& r, bool with_x, bool with_y, float (&xv)[EPC], DzIn<T, EPC>& in) {
  if (with_x) cvt_raw<T, EPC>(r.x, xv);
  if (p.g1) cvt_raw<T, EPC>(r.g1, in.g1);
  if (p.g2) cvt_raw<T, EPC>(r.g2, in.g2);
  if (with_y) cvt_raw<T, EPC>(r.y, in.y);
}
// sgn: values with the sign of the activation's input (the saved y, or fma(x, scale, shift) of the forward)
// slope of the activation's derivative on the negative side (1 / 0 / 0.2 for none / ReLU / LeakyReLU): one select per element
// instead of a test of the activation id per element (which hipcc keeps as scalar branches inside the unrolled loops)
__device__ __forceinline__ float neg_slope_of(int act) { return act == GI_ACT_LRELU ? 0.2f : (act == GI_ACT_RELU ? 0.f : 1.f); }
template <typename T, int EPC>
__device__ __forceinline__ void finish_dz(const BwdP& p, const DzIn<T, EPC>& in, const float (&sgn)[EPC], float (&dz)[EPC]) {
  const float ns = neg_slope_of(p.act);
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const bool pos = sgn[e] > 0.f;
    float g = p.g1 ? in.g1[e] : 0.f;
    if (p.g2) g += pos ? in.g2[e] : 0.f;
    const float sl = pos ? 1.f : ns;
    dz[e] = g * sl * p.drop_scale;
  }
}
// the same with the set of inputs (IN: bit 0 = g1, bit 1 = g2) known at compile time and the two launch constants in registers: no
// test per element
template <typename T, int EPC, int IN>
__device__ __forceinline__ void finish_dz_c(float ns, float drop_scale, const DzIn<T, EPC>& in, const float (&sgn)[EPC], float (&dz)[EPC]) {
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const bool pos = sgn[e] > 0.f;
    float g = (IN & 1) ? in.g1[e] : 0.f;
    if constexpr ((IN & 2) != 0) g += pos ? in.g2[e] : 0.f;
    const float sl = pos ? 1.f : ns;
    dz[e] = g * sl * drop_scale;
  }
}
template <typename T, int EPC>
__device__ __forceinline__ void compute_dz(const BwdP& p, int64_t pix, int ch0, float (&dz)[EPC]) {
  DzIn<T, EPC> in;
  const bool need_y = p.g2 != nullptr || p.act != GI_ACT_NONE;
  load_dz<T, EPC>(p, pix, ch0, need_y, in);
  if (!need_y) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) in.y[e] = 1.f;
  }
  finish_dz<T, EPC>(p, in, in.y, dz);
}

// pass 1: partial sums of dz and dz*xhat. IN / LOAD_Y: which inputs exist (RawB); the host picks the instantiation from the
// pointers (a null g1 / g2 contributes 0, exactly as the run-time test of finish_dz does)
template <typename T, int IN, bool LOAD_Y>
__global__ void __launch_bounds__(256) act_bn_bwd_reduce_kernel(BwdP p) {
  constexpr int EPC = 16 / (int)sizeof(T);
  __shared__ float red[2 * 256 * EPC];
  const int Q = p.c / EPC, RL = 256 / Q;
  const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block, r1 = min(p.pixels, r0 + p.rows_per_block);
  float s[EPC], sx[EPC], mu[EPC], iv[EPC];
  const int so = r0 >= p.pg ? p.stat_stride : 0;
#pragma unroll
  for (int e = 0; e < EPC; ++e) { s[e] = sx[e] = 0.f; mu[e] = p.mean[so + q * EPC + e]; iv[e] = p.inv[so + q * EPC + e]; }
  float sc[EPC], sh[EPC];
  if (p.scale) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = p.scale[so + q * EPC + e]; sh[e] = p.shift[so + q * EPC + e]; }
  }
  const bool need_y = p.g2 != nullptr || p.act != GI_ACT_NONE;
  const float ns = neg_slope_of(p.act), drop_scale = p.drop_scale;
  constexpr int U = 4;   // rows whose loads are in flight together (the sums still run over the rows in ascending order)
  constexpr int64_t ES = (int64_t)sizeof(T);
  const bool have_scale = p.scale != nullptr;
  for (int64_t r = r0 + rl; r < r1; r += (int64_t)U * RL) {
    RawB<T, IN, LOAD_Y> raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t rr = r + (int64_t)u * RL;
      if (rr < r1) {
        raw[u].x = *(const u4_t*)(p.x + (rr * p.c + q * EPC) * ES);
        if constexpr ((IN & 1) != 0) raw[u].g1 = *(const u4_t*)(p.g1 + (rr * p.ldg1 + p.coffg1 + q * EPC) * ES);
        if constexpr ((IN & 2) != 0) raw[u].g2 = *(const u4_t*)(p.g2 + (rr * p.ldg2 + p.coffg2 + q * EPC) * ES);
        if constexpr (LOAD_Y) raw[u].y = *(const u4_t*)(p.y + (rr * p.ldy + p.coffy + q * EPC) * ES);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r + (int64_t)u * RL < r1) {
        float dz[EPC], xv[EPC];
        DzIn<T, EPC> in;
        cvt_raw<T, EPC>(raw[u].x, xv);
        if constexpr ((IN & 1) != 0) cvt_raw<T, EPC>(raw[u].g1, in.g1);
        if constexpr ((IN & 2) != 0) cvt_raw<T, EPC>(raw[u].g2, in.g2);
        if constexpr (LOAD_Y) cvt_raw<T, EPC>(raw[u].y, in.y);
        else {
#pragma unroll
          for (int e = 0; e < EPC; ++e) in.y[e] = have_scale ? fmaf(xv[e], sc[e], sh[e]) : 1.f;
        }
        (void)need_y;
        finish_dz_c<T, EPC, IN>(ns, drop_scale, in, in.y, dz);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { s[e] += dz[e]; sx[e] = fmaf(dz[e], (xv[e] - mu[e]) * iv[e], sx[e]); }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) { red[threadIdx.x * EPC + e] = s[e]; red[256 * EPC + threadIdx.x * EPC + e] = sx[e]; }
  __syncthreads();
  if (p.acc) {   // one lane per channel: the lanes of an atomic instruction are consecutive words (stat_acc.h)
    const int grp = r0 >= p.pg ? 1 : 0, rep = blockIdx.x & (p.acc_reps - 1);
    for (int ch = threadIdx.x; ch < p.c; ch += 256) {
      float a = 0.f, b = 0.f;
      for (int i = 0; i < RL; ++i) { a += red[i * Q * EPC + ch]; b += red[256 * EPC + i * Q * EPC + ch]; }
      gi_stat_add(p.acc, p.c, rep, grp, 0, ch, a);
      gi_stat_add(p.acc, p.c, rep, grp, 1, ch, b);
    }
  } else if (rl == 0) {
    for (int i = 1; i < RL; ++i)
#pragma unroll
      for (int e = 0; e < EPC; ++e) { s[e] += red[(i * Q + q) * EPC + e]; sx[e] += red[256 * EPC + (i * Q + q) * EPC + e]; }
    float* ps = p.partials + ((int64_t)blockIdx.x * 2) * p.c + q * EPC;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { ps[e] = s[e]; ps[p.c + e] = sx[e]; }
  }
}

// partials [rows][2][c] -> sums [2][c]; also accumulates dgamma/dbeta
// groups > 1: `rows` partial rows per population, one after the other; the parameter gradients accumulate population by
// population (the order of separate calls). Output per population [8][c]: s1, s2, then the coefficients of the apply pass
//   dx = k1*dz + k2*x + k3  (k1 = gamma*inv, k2 = -k1*inv*s2/M, k3 = -k1*s1/M - k2*mean)  and the forward's scale / shift.
// rows = 0 (running-statistics BatchNorm): s1 = s2 = 0, dx = gamma*inv*dz.
__global__ void __launch_bounds__(256) bwd_sums_kernel(const float* partials, int rows, int c, float* sums, float* dgamma,
                                                       float* dbeta, float inv_loss_scale, int groups, const float* gamma,
                                                       const float* mean, const float* inv, const float* scale,
                                                       const float* shift, int stat_stride, float invM) {
  __shared__ double sh[8];
  const int ch = blockIdx.x;
  for (int j = 0; j < groups; ++j) {
    const float* pj = partials + (int64_t)j * rows * 2 * c;
    double s = 0.0, q = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) {
      s += (double)pj[((int64_t)r * 2) * c + ch];
      q += (double)pj[((int64_t)r * 2 + 1) * c + ch];
    }
    block_sum2(s, q, sh);
    if (threadIdx.x == 0) {
      float* o = sums + (int64_t)j * 8 * c;
      const float s1 = (float)s, s2 = (float)q;
      o[ch] = s1;
      o[c + ch] = s2;
      if (dbeta) dbeta[ch] += s1 * inv_loss_scale;
      if (dgamma) dgamma[ch] += s2 * inv_loss_scale;
      const int so = j * stat_stride + ch;
      const float iv = inv[so];
      const float k1 = gamma[ch] * iv;
      const float k2 = -k1 * iv * s2 * invM;
      o[2 * c + ch] = k1;
      o[3 * c + ch] = k2;
      o[4 * c + ch] = -k1 * s1 * invM - k2 * mean[so];
      o[5 * c + ch] = scale ? scale[so] : 0.f;
      o[6 * c + ch] = scale ? shift[so] : 0.f;
    }
  }
}

// pass 2 (or the only pass when there is no BN): writes dx. With BatchNorm
//   dx = gamma*inv*(dz - s1/M - xhat*s2/M) = k1*dz + k2*x + k3,   k1 = gamma*inv, k2 = -k1*inv*s2/M, k3 = -k1*s1/M - k2*mean:
// the grid stride is a multiple of the chunks per pixel, so a thread's channels never change and the coefficients
// (both populations' with G2) stay in registers.
// HAS_BN = 2: the coefficients come from the exact accumulators (BwdP::acc), derived by every block into LDS
template <typename T, int HAS_BN, bool G2>
__global__ void __launch_bounds__(256) act_bn_bwd_apply_kernel(BwdP p) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int cpp = p.c / EPC;
  const int lg = 31 - __builtin_clz(cpp);
  const int64_t total = p.pixels * cpp;
  const int64_t gid0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cc = (int)(gid0 & (cpp - 1));
  constexpr int NG = G2 ? 2 : 1;
  constexpr int U = 4;   // chunks whose loads are in flight together
  const int64_t stride = (int64_t)gridDim.x * 256;
  const bool need_y = p.g2 != nullptr || p.act != GI_ACT_NONE;
  const bool load_y = need_y && !(HAS_BN && p.scale);
  // the first U chunks are requested BEFORE the coefficients are derived (HAS_BN == 2: a chain of dependent accumulator and
  // parameter loads during which all co-resident blocks would otherwise request nothing)
  RawIn raw[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (gid0 + u * stride < total) load_raw<T>(p, (gid0 + u * stride) >> lg, cc * EPC, HAS_BN != 0, load_y, raw[u]);
  float k1[NG][EPC], k2[NG][EPC], k3[NG][EPC], sc[NG][EPC], sh[NG][EPC];
  if constexpr (HAS_BN == 2) {
    extern __shared__ __attribute__((aligned(16))) float coef[];   // [groups][5][c]: k1, k2, k3, scale, shift
    const int c = p.c;
    for (int ch = threadIdx.x; ch < c; ch += 256)
      for (int g = 0; g < NG; ++g) {
        const int so = g * p.stat_stride + ch;
        // (parameter loads first: they and the accumulator words are one memory round trip, not two)
        const float iv = p.inv[so], gam = p.gamma[ch], mu = p.mean[so];
        const float fsc = p.scale ? p.scale[so] : 0.f, fsh = p.scale ? p.shift[so] : 0.f;
        float s1 = 0.f, s2 = 0.f;
        if (p.acc) {
          double t1, t2;
          gi_stat_read2(p.acc, c, p.acc_reps, g, ch, t1, t2);
          s1 = (float)t1;
          s2 = (float)t2;
        }
        const float a1 = gam * iv;
        const float a2 = -a1 * iv * s2 * p.invM;
        float* o = coef + (int64_t)g * 5 * c;
        o[ch] = a1;
        o[c + ch] = a2;
        o[2 * c + ch] = -a1 * s1 * p.invM - a2 * mu;
        o[3 * c + ch] = fsc;
        o[4 * c + ch] = fsh;
        if (blockIdx.x == 0) {     // parameter gradients accumulate population by population, as separate calls would
          if (p.dbeta) p.dbeta[ch] += s1 * p.inv_loss_scale;
          if (p.dgamma) p.dgamma[ch] += s2 * p.inv_loss_scale;
        }
      }
    if (blockIdx.x == 0 && p.zero_next) zero_words64(p.zero_next, p.zero_words);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float* o = coef + (int64_t)g * 5 * c + cc * EPC;
#pragma unroll
      for (int e = 0; e < EPC; ++e) { k1[g][e] = o[e]; k2[g][e] = o[c + e]; k3[g][e] = o[2 * c + e]; sc[g][e] = o[3 * c + e]; sh[g][e] = o[4 * c + e]; }
    }
  } else if (HAS_BN) {   // coefficient rows written by bwd_sums_kernel: [group][8][c]
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float* o = p.sums + (int64_t)g * 8 * p.c + cc * EPC;
      load_f32<EPC>(o + 2 * p.c, k1[g]);
      load_f32<EPC>(o + 3 * p.c, k2[g]);
      load_f32<EPC>(o + 4 * p.c, k3[g]);
      if (p.scale) {
        load_f32<EPC>(o + 5 * p.c, sc[g]);
        load_f32<EPC>(o + 6 * p.c, sh[g]);
      }
    }
  }
  for (int64_t gid = gid0; gid < total; gid += U * stride) {
    if (gid != gid0) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (gid + u * stride < total) load_raw<T>(p, (gid + u * stride) >> lg, cc * EPC, HAS_BN != 0, load_y, raw[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (gid + u * stride >= total) break;
      const int64_t pix = (gid + u * stride) >> lg;
      float dz[EPC], xv[EPC];
      DzIn<T, EPC> in;
      unpack_raw<T, EPC>(p, raw[u], HAS_BN != 0, load_y, xv, in);
      if (!HAS_BN) {
        if (!need_y) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) in.y[e] = 1.f;
        }
        finish_dz<T, EPC>(p, in, in.y, dz);
      } else {
        const int g = (G2 && pix >= p.pg) ? 1 : 0;
        if (p.scale || !need_y) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) in.y[e] = p.scale ? fmaf(xv[e], G2 ? (g ? sc[NG - 1][e] : sc[0][e]) : sc[0][e], G2 ? (g ? sh[NG - 1][e] : sh[0][e]) : sh[0][e]) : 1.f;
        }
        finish_dz<T, EPC>(p, in, in.y, dz);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float a1 = G2 ? (g ? k1[NG - 1][e] : k1[0][e]) : k1[0][e];
          const float a2 = G2 ? (g ? k2[NG - 1][e] : k2[0][e]) : k2[0][e];
          const float a3 = G2 ? (g ? k3[NG - 1][e] : k3[0][e]) : k3[0][e];
          dz[e] = fmaf(a1, dz[e], fmaf(a2, xv[e], a3));
        }
      }
      store_vec<T, EPC>(p.dx, pix * p.c + cc * EPC, dz);
    }
  }
}

// The apply pass of the accumulator path (HAS_BN == 2) for the input combinations the networks produce, built for bytes in
// flight: measured with hardware counters (round 3, critic conv2 at n = 64: 200 MB in 57 us), a wave of the generic kernel
// above lives ~18 us - a chain of dependent accumulator / parameter loads, then eight load -> wait -> compute -> store
// rounds - and its 100 - 250 registers leave 2 - 4 waves per SIMD, so the chip has ~7 MB outstanding where a plain copy of
// the same three streams runs at 6.7 TB/s (tools/micro/stream_bw.hip). Here the coefficient vectors stay in LDS (ten
// ds_read_b128 per chunk instead of 40 - 80 registers), only the pointers that exist are loaded (IN: 1 = g1, 2 = g2 masked by
// the activation's sign, 3 = both; LOAD_Y: the sign comes from the saved activation, else from fma(x, scale, shift)), two
// chunks per thread are requested before the coefficients are derived and the next two before the current two are finished.
// Same arithmetic, same order: results are bit-identical to the generic kernel.
template <typename T, int IN, bool LOAD_Y, bool G2>
__global__ void __launch_bounds__(256) act_bn_bwd_apply_acc_kernel(BwdP p) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int64_t ES = (int64_t)sizeof(T);
  constexpr int NG = G2 ? 2 : 1;
  constexpr int U = 2;
  extern __shared__ __attribute__((aligned(16))) float coef[];   // [groups][5][c]: k1, k2, k3, scale, shift
  const int c = p.c;
  const int cpp = c / EPC;
  const int lg = 31 - __builtin_clz(cpp);
  const int64_t total = p.pixels * cpp;
  const int64_t gid0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cc = (int)(gid0 & (cpp - 1));
  const int64_t stride = (int64_t)gridDim.x * 256;
  auto load = [&](int64_t gid, RawB<T, IN, LOAD_Y>& r) {
    const int64_t pix = gid >> lg;
    r.x = *(const u4_t*)(p.x + (pix * c + cc * EPC) * ES);
    if constexpr ((IN & 1) != 0) r.g1 = *(const u4_t*)(p.g1 + (pix * p.ldg1 + p.coffg1 + cc * EPC) * ES);
    if constexpr ((IN & 2) != 0) r.g2 = *(const u4_t*)(p.g2 + (pix * p.ldg2 + p.coffg2 + cc * EPC) * ES);
    if constexpr (LOAD_Y) r.y = *(const u4_t*)(p.y + (pix * p.ldy + p.coffy + cc * EPC) * ES);
  };
  RawB<T, IN, LOAD_Y> cur[U], nxt[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (gid0 + u * stride < total) load(gid0 + u * stride, cur[u]);
  for (int ch = threadIdx.x; ch < c; ch += 256)
    for (int g = 0; g < NG; ++g) {
      const int so = g * p.stat_stride + ch;
      const float iv = p.inv[so], gam = p.gamma[ch], mu = p.mean[so];
      const float fsc = p.scale ? p.scale[so] : 0.f, fsh = p.scale ? p.shift[so] : 0.f;
      float s1 = 0.f, s2 = 0.f;
      if (p.acc) {
        double t1, t2;
        gi_stat_read2(p.acc, c, p.acc_reps, g, ch, t1, t2);
        s1 = (float)t1;
        s2 = (float)t2;
      }
      const float a1 = gam * iv;
      const float a2 = -a1 * iv * s2 * p.invM;
      float* o = coef + (int64_t)g * 5 * c;
      o[ch] = a1;
      o[c + ch] = a2;
      o[2 * c + ch] = -a1 * s1 * p.invM - a2 * mu;
      o[3 * c + ch] = fsc;
      o[4 * c + ch] = fsh;
      if (blockIdx.x == 0) {     // parameter gradients accumulate population by population, as separate calls would
        if (p.dbeta) p.dbeta[ch] += s1 * p.inv_loss_scale;
        if (p.dgamma) p.dgamma[ch] += s2 * p.inv_loss_scale;
      }
    }
  if (blockIdx.x == 0 && p.zero_next) zero_words64(p.zero_next, p.zero_words);
  __syncthreads();
  const bool have_scale = p.scale != nullptr;
  const float ns = neg_slope_of(p.act);
  const float drop_scale = p.drop_scale;
  for (int64_t gid = gid0; gid < total; gid += U * stride) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (gid + (U + u) * stride < total) load(gid + (U + u) * stride, nxt[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (gid + u * stride < total) {
        const int64_t pix = (gid + u * stride) >> lg;
        const float* o = coef + ((G2 && pix >= p.pg) ? 5 * c : 0) + cc * EPC;
        float xv[EPC], sg[EPC], g1v[EPC], g2v[EPC], k[EPC], out[EPC];
        cvt_raw<T, EPC>(cur[u].x, xv);
        if constexpr (LOAD_Y) cvt_raw<T, EPC>(cur[u].y, sg);
        else {
          float scv[EPC], shv[EPC];
          load_f32<EPC>(o + 3 * c, scv);
          load_f32<EPC>(o + 4 * c, shv);
#pragma unroll
          for (int e = 0; e < EPC; ++e) sg[e] = have_scale ? fmaf(xv[e], scv[e], shv[e]) : 1.f;
        }
        if constexpr ((IN & 1) != 0) cvt_raw<T, EPC>(cur[u].g1, g1v);
        if constexpr ((IN & 2) != 0) cvt_raw<T, EPC>(cur[u].g2, g2v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const bool pos = sg[e] > 0.f;
          float g = (IN & 1) ? g1v[e] : 0.f;
          if constexpr ((IN & 2) != 0) g += pos ? g2v[e] : 0.f;
          const float sl = pos ? 1.f : ns;
          out[e] = g * sl * drop_scale;
        }
        // dx = k1 * dz + (k2 * x + k3), the same fmaf nest as the generic kernel
        float k1v[EPC], k3v[EPC];
        load_f32<EPC>(o, k1v);
        load_f32<EPC>(o + c, k);
        load_f32<EPC>(o + 2 * c, k3v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) out[e] = fmaf(k1v[e], out[e], fmaf(k[e], xv[e], k3v[e]));
        store_vec<T, EPC>(p.dx, pix * c + cc * EPC, out);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = nxt[u];
  }
}

// Small tensors (the U-Net's innermost levels: <= 512 pixels by default, the kernel serves up to 2048): reduce and apply in ONE launch. A workgroup owns a 16-byte
// channel chunk and all pixels, every thread holds its <= 8 chunks of dz and x in registers: sums of dz and dz * xhat (fp32
// per thread, double across the block, fixed order: reproducible), coefficients, dx. The two-launch form costs 13 + 14 us
// per layer there, almost all of it launch and accumulator round trips (12 us in this form). Beyond ~512 pixels the
// channel-chunk-per-workgroup layout loses: a workgroup uses 16 bytes of every 128-byte line it touches (28 us at 2048 pixels).
template <typename T, int NIT>   // NIT = pixels / 256 rounded up: every thread keeps its NIT chunks (dz, x) in registers between the sweeps
__global__ void __launch_bounds__(256) act_bn_bwd_small_kernel(BwdP p) {
  constexpr int EPC = 16 / (int)sizeof(T);
  __shared__ double sh[8];
  __shared__ float coef[3 * 8];
  const int cc = blockIdx.x;
  float mu[EPC], iv[EPC], sc[EPC], shf[EPC], s[EPC], sx[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    mu[e] = p.mean[cc * EPC + e]; iv[e] = p.inv[cc * EPC + e];
    sc[e] = p.scale ? p.scale[cc * EPC + e] : 0.f; shf[e] = p.scale ? p.shift[cc * EPC + e] : 0.f;
    s[e] = sx[e] = 0.f;
  }
  const bool need_y = p.g2 != nullptr || p.act != GI_ACT_NONE;
  float xr[NIT][EPC], dzr[NIT][EPC];
  DzIn<T, EPC> in[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {    // all loads first
    const int64_t pix = threadIdx.x + it * 256;
    if (pix < p.pixels) {
      load_vec<T, EPC>(p.x, pix * p.c + cc * EPC, xr[it]);
      load_dz<T, EPC>(p, pix, cc * EPC, need_y && !p.scale, in[it]);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int64_t pix = threadIdx.x + it * 256;
    if (pix < p.pixels) {
      if (p.scale || !need_y) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) in[it].y[e] = p.scale ? fmaf(xr[it][e], sc[e], shf[e]) : 1.f;
      }
      finish_dz<T, EPC>(p, in[it], in[it].y, dzr[it]);
#pragma unroll
      for (int e = 0; e < EPC; ++e) { s[e] += dzr[it][e]; sx[e] = fmaf(dzr[it][e], (xr[it][e] - mu[e]) * iv[e], sx[e]); }
    }
  }
  __shared__ float tot[2 * 8];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    double a = s[e], b = sx[e];
    block_sum2(a, b, sh);
    if (threadIdx.x == 0) { tot[e] = (float)a; tot[8 + e] = (float)b; }
  }
  __syncthreads();
  if (threadIdx.x < EPC) {     // one lane per channel: the parameter-gradient updates overlap instead of queueing behind one lane
    const int e = threadIdx.x, ch = cc * EPC + e;
    const float s1 = tot[e], s2 = tot[8 + e];
    const float ive = p.inv[ch], mue = p.mean[ch];
    const float a1 = p.gamma[ch] * ive;
    const float a2 = -a1 * ive * s2 * p.invM;
    coef[e] = a1; coef[8 + e] = a2; coef[16 + e] = -a1 * s1 * p.invM - a2 * mue;
    if (p.dbeta) p.dbeta[ch] += s1 * p.inv_loss_scale;
    if (p.dgamma) p.dgamma[ch] += s2 * p.inv_loss_scale;
  }
  if (blockIdx.x == 0 && p.zero_next) zero_words64(p.zero_next, p.zero_words);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int64_t pix = threadIdx.x + it * 256;
    if (pix < p.pixels) {
      float o[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) o[e] = fmaf(coef[e], dzr[it][e], fmaf(coef[8 + e], xr[it][e], coef[16 + e]));
      store_vec<T, EPC>(p.dx, pix * p.c + cc * EPC, o);
    }
  }
}

// ---- InstanceNorm2d(affine=False, track_running_stats=False), get_norm_layer('instance') (networks.py:38-40) -------------
// Statistics per (image, channel) over the H*W pixels, eps 1e-5, the same in train and eval mode; no parameters. One
// workgroup owns (image, 16-byte channel chunk): a first sweep over the image's pixels sums x and x^2 (fp32 per thread,
// double across the block, fixed order), a second one writes act((x - mean) * inv) (+ dropout) - the plane is re-read
// from L2. stats [n][c][2] keeps mean / inv for the backward.
template <typename T>
__global__ void __launch_bounds__(256) in_forward_kernel(const char* x, char* y, int hw, int c, int ldy, int coffy, int act,
                                                         const uint8_t* drop, float drop_scale, float eps, float* stats) {
  constexpr int EPC = 16 / (int)sizeof(T);
  __shared__ double sh[8];
  __shared__ float aff[2 * 8];
  const int cpp = c / EPC;
  const int img = blockIdx.x / cpp, cc = blockIdx.x % cpp;
  const int64_t p0 = (int64_t)img * hw;
  float s[EPC], q[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) s[e] = q[e] = 0.f;
  for (int i = threadIdx.x; i < hw; i += 256) {
    float v[EPC];
    load_vec<T, EPC>(x, (p0 + i) * c + cc * EPC, v);
#pragma unroll
    for (int e = 0; e < EPC; ++e) { s[e] += v[e]; q[e] = fmaf(v[e], v[e], q[e]); }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    double a = s[e], b = q[e];
    block_sum2(a, b, sh);
    if (threadIdx.x == 0) {
      const double m = a / hw;
      double var = b / hw - m * m;
      if (var < 0.0) var = 0.0;
      const float inv = 1.0f / sqrtf((float)var + eps);
      aff[e] = (float)m; aff[8 + e] = inv;
      stats[((int64_t)img * c + cc * EPC + e) * 2] = (float)m;
      stats[((int64_t)img * c + cc * EPC + e) * 2 + 1] = inv;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < hw; i += 256) {
    float v[EPC];
    load_vec<T, EPC>(x, (p0 + i) * c + cc * EPC, v);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float t = (v[e] - aff[e]) * aff[8 + e];
      if (act == GI_ACT_RELU) t = t > 0.f ? t : 0.f;
      else if (act == GI_ACT_LRELU) t = t > 0.f ? t : 0.2f * t;
      v[e] = t;
    }
    if (drop) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] = drop[(p0 + i) * c + cc * EPC + e] ? v[e] * drop_scale : 0.f;
    }
    store_vec<T, EPC>(y, (p0 + i) * ldy + coffy + cc * EPC, v);
  }
}

// backward through [dropout] -> activation -> InstanceNorm: dz as in act_bn_bwd (the masks come from the saved output y),
//   dx = inv * (dz - mean_p dz - xhat * mean_p (dz * xhat)),  the means over the pixels of ONE (image, channel)
template <typename T>
__global__ void __launch_bounds__(256) act_in_bwd_kernel(BwdP p, int hw, const float* stats) {
  constexpr int EPC = 16 / (int)sizeof(T);
  __shared__ double sh[8];
  __shared__ float red[2 * 8];
  const int cpp = p.c / EPC;
  const int img = blockIdx.x / cpp, cc = blockIdx.x % cpp;
  const int64_t p0 = (int64_t)img * hw;
  float mu[EPC], iv[EPC], s[EPC], sx[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    mu[e] = stats[((int64_t)img * p.c + cc * EPC + e) * 2];
    iv[e] = stats[((int64_t)img * p.c + cc * EPC + e) * 2 + 1];
    s[e] = sx[e] = 0.f;
  }
  for (int i = threadIdx.x; i < hw; i += 256) {
    float dz[EPC], xv[EPC];
    compute_dz<T, EPC>(p, p0 + i, cc * EPC, dz);
    load_vec<T, EPC>(p.x, (p0 + i) * p.c + cc * EPC, xv);
#pragma unroll
    for (int e = 0; e < EPC; ++e) { s[e] += dz[e]; sx[e] = fmaf(dz[e], (xv[e] - mu[e]) * iv[e], sx[e]); }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    double a = s[e], b = sx[e];
    block_sum2(a, b, sh);
    if (threadIdx.x == 0) { red[e] = (float)(a / hw); red[8 + e] = (float)(b / hw); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < hw; i += 256) {
    float dz[EPC], xv[EPC];
    compute_dz<T, EPC>(p, p0 + i, cc * EPC, dz);
    load_vec<T, EPC>(p.x, (p0 + i) * p.c + cc * EPC, xv);
#pragma unroll
    for (int e = 0; e < EPC; ++e) dz[e] = iv[e] * (dz[e] - red[e] - (xv[e] - mu[e]) * iv[e] * red[8 + e]);
    store_vec<T, EPC>(p.dx, (p0 + i) * p.c + cc * EPC, dz);
  }
}

// bias gradient of a convolution: dbias[ch] += scale * sum over the partial rows of column sums (col_stats_kernel)
__global__ void __launch_bounds__(256) bias_grad_kernel(const float* partials, int rows, int c, float scale, float* dbias) {
  __shared__ double sh[8];
  const int ch = blockIdx.x;
  double s = 0.0, dummy = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) s += (double)partials[((int64_t)r * 2) * c + ch];
  block_sum2(s, dummy, sh);
  if (threadIdx.x == 0) dbias[ch] += (float)(s * scale);
}

// ---- gradient-penalty helpers (WGAN-GP extension) ------------------------------------
// t_out = t_in * slope(a): LeakyReLU Jacobian applied to a tangent (a = primal activation output)
template <typename T>
__global__ void __launch_bounds__(256) mul_slope_kernel(const char* tin, const char* a, char* tout, int64_t chunks) {
  constexpr int EPC = 16 / (int)sizeof(T);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (int64_t)gridDim.x * 256) {
    float t[EPC], av[EPC];
    load_vec<T, EPC>(tin, i * EPC, t);
    load_vec<T, EPC>(a, i * EPC, av);
#pragma unroll
    for (int e = 0; e < EPC; ++e) t[e] *= av[e] > 0.f ? 1.f : 0.2f;
    store_vec<T, EPC>(tout, i * EPC, t);
  }
}

// BatchNorm (train mode) tangent map tz = (gamma/sigma)(tx - mean(tx) - xhat*mean(xhat*tx)). Its dependence
// on the PRIMAL x (through sigma and xhat) sends a gradient into x when the tangent graph is
// differentiated: with a = slope(y) * d(ta) (gradient w.r.t. tz), per channel
//   A   = sum(a*u),  u = tx - m1 - xhat*m2,  m1 = mean(tx), m2 = mean(xhat*tx)
//   inj = -(gamma*inv^2)(xhat/M) A - gamma*inv*( m2*P(a) + mean(a*xhat)*P(tx) ),  P(w) = inv*(w - mean(w) - xhat*mean(w*xhat))
//   dgamma += A*inv
// pass 1: partial sums [blocks][5][c] of a, a*xhat, tx, tx*xhat, a*tx
struct InjP {
  const char* dta; const char* y; const char* tx; const char* x; char* dxp;   // dxp: += inj (in place on the primal gradient)
  int64_t pixels; int c;
  const float* gamma; const float* mean; const float* inv;
  float* partials; const float* sums; float* dgamma;
  int rows_per_block;
};
template <typename T>
__global__ void __launch_bounds__(256) bn_inject_reduce_kernel(InjP p) {
  constexpr int EPC = 16 / (int)sizeof(T);
  __shared__ float red[5 * 256 * EPC];
  const int Q = p.c / EPC, RL = 256 / Q;
  const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block, r1 = min(p.pixels, r0 + p.rows_per_block);
  float s[5][EPC], mu[EPC], iv[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k][e] = 0.f;
    mu[e] = p.mean[q * EPC + e];
    iv[e] = p.inv[q * EPC + e];
  }
  for (int64_t r = r0 + rl; r < r1; r += RL) {
    float d[EPC], yv[EPC], t[EPC], xv[EPC];
    load_vec<T, EPC>(p.dta, r * p.c + q * EPC, d);
    load_vec<T, EPC>(p.y, r * p.c + q * EPC, yv);
    load_vec<T, EPC>(p.tx, r * p.c + q * EPC, t);
    load_vec<T, EPC>(p.x, r * p.c + q * EPC, xv);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float a = d[e] * (yv[e] > 0.f ? 1.f : 0.2f);
      const float xh = (xv[e] - mu[e]) * iv[e];
      s[0][e] += a; s[1][e] = fmaf(a, xh, s[1][e]); s[2][e] += t[e]; s[3][e] = fmaf(t[e], xh, s[3][e]); s[4][e] = fmaf(a, t[e], s[4][e]);
    }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[(k * 256 + threadIdx.x) * EPC + e] = s[k][e];
  __syncthreads();
  if (rl == 0) {
    for (int i = 1; i < RL; ++i)
#pragma unroll
      for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int e = 0; e < EPC; ++e) s[k][e] += red[(k * 256 + i * Q + q) * EPC + e];
    float* ps = p.partials + ((int64_t)blockIdx.x * 5) * p.c + q * EPC;
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
      for (int e = 0; e < EPC; ++e) ps[k * p.c + e] = s[k][e];
  }
}
__global__ void __launch_bounds__(256) bn_inject_sums_kernel(const float* partials, int rows, int c, float* sums) {
  __shared__ double sh[5][256];
  const int ch = blockIdx.x;
  double s[5] = {0, 0, 0, 0, 0};
  for (int r = threadIdx.x; r < rows; r += 256)
    for (int k = 0; k < 5; ++k) s[k] += (double)partials[((int64_t)r * 5 + k) * c + ch];
  for (int k = 0; k < 5; ++k) sh[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) for (int k = 0; k < 5; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) for (int k = 0; k < 5; ++k) sums[k * c + ch] = (float)sh[k][0];
}
template <typename T>
__global__ void __launch_bounds__(256) bn_inject_apply_kernel(InjP p) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int cpp = p.c / EPC;
  const int64_t total = p.pixels * cpp;
  const float M = (float)p.pixels, invM = 1.f / M;
  for (int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (int64_t)gridDim.x * 256) {
    const int cc = (int)(gid % cpp);
    const int64_t pix = gid / cpp;
    float d[EPC], yv[EPC], t[EPC], xv[EPC], o[EPC];
    load_vec<T, EPC>(p.dta, pix * p.c + cc * EPC, d);
    load_vec<T, EPC>(p.y, pix * p.c + cc * EPC, yv);
    load_vec<T, EPC>(p.tx, pix * p.c + cc * EPC, t);
    load_vec<T, EPC>(p.x, pix * p.c + cc * EPC, xv);
    load_vec<T, EPC>(p.dxp, pix * p.c + cc * EPC, o);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int ch = cc * EPC + e;
      const float inv = p.inv[ch], g = p.gamma[ch];
      const float a = d[e] * (yv[e] > 0.f ? 1.f : 0.2f);
      const float xh = (xv[e] - p.mean[ch]) * inv;
      const float Sa = p.sums[ch], Sax = p.sums[p.c + ch], St = p.sums[2 * p.c + ch], Stx = p.sums[3 * p.c + ch], Sat = p.sums[4 * p.c + ch];
      const float m2 = Stx * invM, max_ = Sax * invM;
      const float A = Sat - Sa * St * invM - Sax * Stx * invM;
      const float Pa = inv * (a - Sa * invM - xh * max_);
      const float Pt = inv * (t[e] - St * invM - xh * m2);
      o[e] += -(g * inv * inv) * (xh * invM) * A - g * inv * (m2 * Pa + max_ * Pt);
    }
    store_vec<T, EPC>(p.dxp, pix * p.c + cc * EPC, o);
  }
  if (p.dgamma && blockIdx.x == 0) {
    for (int ch = threadIdx.x; ch < p.c; ch += 256) {
      const float Sa = p.sums[ch], Sax = p.sums[p.c + ch], St = p.sums[2 * p.c + ch], Stx = p.sums[3 * p.c + ch], Sat = p.sums[4 * p.c + ch];
      p.dgamma[ch] += (Sat - Sa * St * invM - Sax * Stx * invM) * p.inv[ch];
    }
  }
}

// per-sample squared L2 norm of g (n, hw) -> sumsq[n]
__global__ void __launch_bounds__(256) sample_sumsq_kernel(const float* g, int64_t hw, float* sumsq) {
  __shared__ double sh[4];
  const float* p = g + (int64_t)blockIdx.x * hw;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < hw; i += 256) s += (double)p[i] * p[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sumsq[blockIdx.x] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}
// v[n,:] = lam/N * 2 (||g_n|| - 1) / ||g_n|| * g[n,:] ; penalty = lam * mean_n (||g_n|| - 1)^2 (written by block 0)
__global__ void __launch_bounds__(256) gp_direction_kernel(const float* g, const float* sumsq, int n, int64_t hw, float lam,
                                                           float* v, float* penalty) {
  const int nn = blockIdx.y;
  const float norm = sqrtf(sumsq[nn]);
  const float coef = norm > 0.f ? lam / (float)n * 2.f * (norm - 1.f) / norm : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256)
    v[(int64_t)nn * hw + i] = coef * g[(int64_t)nn * hw + i];
  if (penalty && blockIdx.x == 0 && nn == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int k = 0; k < n; ++k) { const double d = sqrt((double)sumsq[k]) - 1.0; s += d * d; }
    penalty[0] = (float)(lam * s / n);
  }
}
// out[n,:] = fake[n,:] + eps[n] * (real[n,:] - fake[n,:])
__global__ void __launch_bounds__(256) interpolate_kernel(const float* real, const float* fake, const float* eps, int64_t hw,
                                                          float* out) {
  const int nn = blockIdx.y;
  const float e = eps[nn];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
    const int64_t j = (int64_t)nn * hw + i;
    out[j] = e * real[j] + (1.f - e) * fake[j];
  }
}

// ---- mask pipeline (bit-exact: selects / multiplies by 0 or 1) --------------------------------
__global__ void __launch_bounds__(256) mask_apply_kernel(const float* ground, const float* mask, float* mask_c,
                                                         float* masked, int64_t count, int do_ceil) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    float m = (do_ceil & 1) ? ceilf(mask[i]) : mask[i];
    if (do_ceil & 2) m = 1.f - m;
    if (mask_c) mask_c[i] = m;
    masked[i] = ground[i] * (1.f - m);
  }
}
__global__ void __launch_bounds__(256) mask_composite_kernel(const float* masked, const float* gen, const float* mask_c,
                                                             float* out, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
    out[i] = masked[i] + gen[i] * mask_c[i];   // two roundings like the reference (mul, then add)
}
__global__ void __launch_bounds__(256) mul_kernel(const float* a, const float* b, float* out, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) out[i] = a[i] * b[i];
}
__global__ void __launch_bounds__(256) add_kernel(const float* a, const float* b, float* out, int64_t count, float alpha) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) out[i] = a[i] + alpha * b[i];
}
__global__ void __launch_bounds__(256) tanh_bwd_kernel(const float* dy, const float* y, float* dx, int64_t count, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
    dx[i] = dy[i] * (1.f - y[i] * y[i]) * scale;
}

// ---- reconstruction losses --------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* sh) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// kind: 0 sum|a-b| ; 1 sum (a-b)^2 ; 2/3 masked versions (y*m - yhat*m), second sum = count(m != 0)
__global__ void __launch_bounds__(256) loss_partial_kernel(const float* a, const float* b, const float* m, int64_t count,
                                                           int kind, float* scratch) {
  __shared__ double sh[4];
  double s = 0.0, cnt = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    float d;
    if (m) {
      const float mm = m[i];
      d = b[i] * mm - a[i] * mm;
      cnt += (mm != 0.f) ? 1.0 : 0.0;
    } else {
      d = a[i] - b[i];
    }
    s += (kind & 1) ? (double)d * d : (double)fabsf(d);
  }
  s = block_sum(s, sh);
  cnt = block_sum(cnt, sh);
  if (threadIdx.x == 0) { scratch[blockIdx.x * 2] = (float)s; scratch[blockIdx.x * 2 + 1] = (float)cnt; }
}
// mode: 0 mean ; 1 sqrt(mean + eps)
__global__ void __launch_bounds__(256) loss_final_kernel(const float* scratch, int nblocks, double denom, int use_cnt,
                                                         int mode, float eps, float* loss_out) {
  __shared__ double sh[4];
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) { s += scratch[i * 2]; c += scratch[i * 2 + 1]; }
  s = block_sum(s, sh);
  c = block_sum(c, sh);
  if (threadIdx.x == 0) {
    const double d = use_cnt ? c : denom;
    double l = s / d;
    if (mode == 1) l = sqrt(l + (double)eps);
    loss_out[0] = (float)l;
    loss_out[1] = (float)d;   // denominator kept for the gradient kernel
  }
}
// gkind: 0 L1 ; 1 MSE ; 2 RMSE  (masked when m != null: gradient w.r.t. yhat = a)
__global__ void __launch_bounds__(256) loss_grad_kernel(const float* a, const float* b, const float* m, int64_t count,
                                                        int gkind, const float* loss, float gscale, float* grad) {
  const float l = loss[0], den = loss[1];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    float d, mm = 1.f;
    if (m) { mm = m[i]; d = a[i] * mm - b[i] * mm; }
    else d = a[i] - b[i];
    float g;
    if (gkind == 0) g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / den;
    else if (gkind == 1) g = 2.f * d / den;
    else g = d / (den * l);
    grad[i] = g * mm * gscale;
  }
}

// adversarial losses on (n,) predictions. Workgroup b of a two-workgroup launch (gi_loss_adv_pair: the [real | fake] halves of a
// stacked discriminator batch, their targets, loss slots and gradient signs) works on the b-th half.
__global__ void __launch_bounds__(256) adv_loss_kernel(const float* pred, int n, int kind, float target, float* loss_out,
                                                       float* grad, float gscale, float target1, float* loss_out1, float gscale1) {
  __shared__ double sh[4];
  if (blockIdx.x == 1) { pred += n; if (grad) grad += n; target = target1; loss_out = loss_out1; gscale = gscale1; }
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float p = pred[i];
    float l, g;
    if (kind == 0) {  // nn.BCELoss: log clamped at -100; backward (p-t)/max(p(1-p),1e-12)
      const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
      l = -(target * lp + (1.f - target) * l1p);
      g = (p - target) / fmaxf(p * (1.f - p), 1e-12f);
    } else if (kind == 1) {
      l = (p - target) * (p - target);
      g = 2.f * (p - target);
    } else {
      l = p;
      g = 1.f;
    }
    s += l;
    if (grad) grad[i] = g / (float)n * gscale;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) loss_out[0] = (float)(s / n);
}

// ---- optimizers --------------------------------------------------------------------------------
// The verdict of the finite check (gi_check_finite*): guard[1] after the finish launch (vword = 0), or - vword = 2 / 3, no finish
// launch - the scan word itself: every workgroup reads it, and for the first kernel of an update (finish) one thread also does what
// the finish launch did (guard[1] = verdict, guard[0] += verdict) and clears the OTHER scan word for the next update's scan. The word
// in use cannot be cleared here (later workgroups still read it): updates alternate between the two.
__device__ __forceinline__ int gi_guard_verdict(int* guard, int vword, int finish) {
  if (!guard) return 0;
  if (vword == 0) return guard[1];
  const int verdict = guard[vword];
  if (finish && blockIdx.x == 0 && threadIdx.x == 0) {
    guard[1] = verdict;
    if (verdict) guard[0] += 1;
    guard[vword ^ 1] = 0;
  }
  return verdict;
}
// guard (may be null): guard[1] != 0 means "the gradients of this update hold inf/NaN" (gi_check_finite): skip it whole
// omb1 / omb2 / step_size: 1 - beta1, 1 - beta2 and lr / bias_correction1 are formed in double on the host and rounded once,
// as torch.optim does with its Python floats (1.f - 0.999f differs from float(1 - 0.999) by 1e-4 relative)
// seen >= 0 (with a guard): the host's step count still includes the updates skipped since its last poll (guard[0] - seen of them):
// the bias corrections are formed here from step - (guard[0] - seen), in double by one thread, so a skipped update never advances them
__global__ void __launch_bounds__(256) adam_kernel(float* p, const float* g, float* m, float* v, int64_t count, float step_size,
                                                   float b1, float b2, float eps, float omb1, float omb2, float sqrt_bc2, float gs,
                                                   int* guard, int step, int seen, double lrd, double b1d, double b2d, int vword, int finish) {
  if (gi_guard_verdict(guard, vword, finish)) return;
  if (guard && seen >= 0) {
    __shared__ float s_ss, s_sq;
    if (threadIdx.x == 0) {
      int st = step - (guard[0] - seen);
      if (st < 1) st = 1;
      s_ss = (float)(lrd / (1.0 - pow(b1d, (double)st)));
      s_sq = (float)sqrt(1.0 - pow(b2d, (double)st));
    }
    __syncthreads();
    step_size = s_ss;
    sqrt_bc2 = s_sq;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const float gg = g[i] * gs;
    const float mm = b1 * m[i] + omb1 * gg;
    const float vv = b2 * v[i] + omb2 * gg * gg;
    m[i] = mm;
    v[i] = vv;
    const float denom = sqrtf(vv) / sqrt_bc2 + eps;
    p[i] = p[i] - step_size * (mm / denom);
  }
}
__global__ void __launch_bounds__(256) rmsprop_kernel(float* p, const float* g, float* sq, int64_t count, float lr,
                                                      float alpha, float oma, float eps, float clampv, float gs, int* guard, int vword, int finish) {
  if (gi_guard_verdict(guard, vword, finish)) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const float gg = g[i] * gs;
    const float s = alpha * sq[i] + oma * gg * gg;
    sq[i] = s;
    float w = p[i] - lr * (gg / (sqrtf(s) + eps));
    if (clampv > 0.f) w = fminf(fmaxf(w, -clampv), clampv);
    p[i] = w;
  }
}
// flag[2] |= 1 when any element of g is inf or NaN.
// (Measured and dropped in round 4: the workgroup that ends last also doing the finish - every workgroup adds to a ticket on the
//  same word after its OR - made the scan 5 -> 20 us on the critic's 1350 workgroups: 1350 returning atomics on one address at ~12 ns.)
__global__ void __launch_bounds__(256) check_finite_kernel(const float* __restrict__ g, int64_t count, int* flag, int word) {
  int bad = 0;
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? count / 4 : 0;   // 16-byte chunks, two in flight per thread
  const u4_t* g4 = (const u4_t*)g;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const u4_t a = g4[i], b = g4[i + stride];
#pragma unroll
    for (int e = 0; e < 4; ++e) bad |= (((a[e] & 0x7F800000u) == 0x7F800000u) || ((b[e] & 0x7F800000u) == 0x7F800000u)) ? 1 : 0;
  }
  if (i < n4) {
    const u4_t a = g4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) bad |= ((a[e] & 0x7F800000u) == 0x7F800000u) ? 1 : 0;
  }
  for (int64_t j = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; j < count; j += stride) {   // what 16-byte chunks do not cover
    const unsigned u = __float_as_uint(g[j]);
    bad |= ((u & 0x7F800000u) == 0x7F800000u) ? 1 : 0;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag + word, 1);
}
// flag[1] = this update's verdict, flag[0] += it (running count of skipped updates), flag[2] = 0
__global__ void check_finite_finish_kernel(int* flag) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int b = flag[2];
    flag[1] = b;
    flag[0] += b;
    flag[2] = 0;
  }
}
__global__ void __launch_bounds__(256) clamp_kernel(float* p, int64_t count, float lo, float hi) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
    p[i] = fminf(fmaxf(p[i], lo), hi);
}
// grid (slices, nseg): out[s] += sum|g| over slice / len
__global__ void __launch_bounds__(256) absmean_kernel(const float* g, const int64_t* off, const int64_t* len, float* out) {
  __shared__ double sh[4];
  const int sgm = blockIdx.y;
  const int64_t o = off[sgm], l = len[sgm];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < l; i += (int64_t)gridDim.x * 256) s += fabsf(g[o + i]);
  s = block_sum(s, sh);
  if (threadIdx.x == 0 && s != 0.0) atomicAdd(out + sgm, (float)(s / (double)l));
}

// ---- weight packing / conversion ---------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) convert_kernel(const float* src, T* dst, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) dst[i] = (T)src[i];
}
template <typename T>
__global__ void __launch_bounds__(256) convert_back_kernel(const T* src, float* dst, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) dst[i] = (float)src[i];
}
// w [a][16][b] -> wph [4 phases][b][4 taps][a], tap t=(ty,tx) of phase (py,px) is (ky,kx)=(1-py+2ty, 1-px+2tx).
// 32x32 (a,b) tile transposed through LDS so that both the read (b fastest) and the write
// (a fastest) are coalesced.
template <typename T>
__global__ void __launch_bounds__(256) pack_phase_kernel(const float* w, T* wph, int ca, int cb) {
  __shared__ float tile[32][33];
  const int k16 = blockIdx.z;   // ph*4 + t
  const int ph = k16 >> 2, t = k16 & 3;
  const int ky = 1 - (ph >> 1) + 2 * (t >> 1), kx = 1 - (ph & 1) + 2 * (t & 1);
  const int a0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int a = a0 + r, b = b0 + tx;
    tile[r][tx] = (a < ca && b < cb) ? w[((int64_t)a * 16 + ky * 4 + kx) * cb + b] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int b = b0 + r, a = a0 + tx;
    if (a < ca && b < cb) wph[(((int64_t)ph * cb + b) * 4 + t) * ca + a] = (T)tile[tx][r];
  }
}

// All layers of a network in one launch: a workgroup takes one 32x32 (a,b) tile of one tap of one layer, writes the
// plain converted copy (same [a][16][b] layout) and the transposed sub-pixel-phase copy from a single read.
template <typename T>
__global__ void __launch_bounds__(256) pack_batch_kernel(PackJobs P) {
  __shared__ float tile[32][33];
  int jb = 0;
  while (jb + 1 < P.n && (int)blockIdx.x >= P.j[jb + 1].tile0) ++jb;
  const float* w = P.j[jb].w;
  T* packed = (T*)P.j[jb].packed;
  T* wph = (T*)P.j[jb].phase;
  const int ca = P.j[jb].ca, cb = P.j[jb].cb, tiles_a = (ca + 31) / 32, tiles_b = (cb + 31) / 32;
  int t = (int)blockIdx.x - P.j[jb].tile0;
  const int a0 = (t % tiles_a) * 32;
  t /= tiles_a;
  const int b0 = (t % tiles_b) * 32, k16 = t / tiles_b;   // k16 = ph*4 + tap
  const int ph = k16 >> 2, tp = k16 & 3;
  const int ky = 1 - (ph >> 1) + 2 * (tp >> 1), kx = 1 - (ph & 1) + 2 * (tp & 1);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int a = a0 + r, b = b0 + tx;
    float v = 0.f;
    if (a < ca && b < cb) {
      const int64_t idx = ((int64_t)a * 16 + ky * 4 + kx) * cb + b;
      v = w[idx];
      if (P.j[jb].scale) v *= P.j[jb].scale[P.j[jb].scale_on_b ? b : a];   // inference: the following BatchNorm's scale folded in
      if (packed) packed[idx] = (T)v;
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  if (!wph) return;
  for (int r = ty; r < 32; r += 8) {
    const int b = b0 + r, a = a0 + tx;
    if (a < ca && b < cb) wph[(((int64_t)ph * cb + b) * 4 + tp) * ca + a] = (T)tile[tx][r];
  }
}

// The same on 64 x 64 tiles with four values per lane (layers whose channel counts are multiples of 64, i.e. every layer of the
// generator and the critic but the single-channel ones): 128-byte rows on both fp16 copies instead of 64-byte ones, a quarter of
// the workgroups.
template <typename T>
__global__ void __launch_bounds__(256) pack_batch64_kernel(PackJobs P) {
  __shared__ float tile[64][65];
  int jb = 0;
  while (jb + 1 < P.n && (int)blockIdx.x >= P.j[jb + 1].tile0) ++jb;
  const float* w = P.j[jb].w;
  T* packed = (T*)P.j[jb].packed;
  T* wph = (T*)P.j[jb].phase;
  const int ca = P.j[jb].ca, cb = P.j[jb].cb, tiles_a = ca / 64, tiles_b = cb / 64;
  int t = (int)blockIdx.x - P.j[jb].tile0;
  const int a0 = (t % tiles_a) * 64;
  t /= tiles_a;
  const int b0 = (t % tiles_b) * 64, k16 = t / tiles_b;   // k16 = ph*4 + tap
  const int ph = k16 >> 2, tp = k16 & 3;
  const int ky = 1 - (ph >> 1) + 2 * (tp >> 1), kx = 1 - (ph & 1) + 2 * (tp & 1);
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16, four consecutive values per lane
  const float* sc = P.j[jb].scale;
  const int on_b = P.j[jb].scale_on_b;
  for (int r = ty; r < 64; r += 16) {
    const int a = a0 + r, b = b0 + tx * 4;
    const int64_t idx = ((int64_t)a * 16 + ky * 4 + kx) * cb + b;
    f4_t v = *(const f4_t*)(w + idx);
    if (sc) {
      if (on_b) v *= *(const f4_t*)(sc + b);
      else v *= sc[a];
    }
    if (packed) {
      T o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (T)v[e];
      if constexpr (sizeof(T) == 2) *(uint2*)(packed + idx) = *(const uint2*)o;
      else *(f4_t*)(packed + idx) = *(const f4_t*)o;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[r][tx * 4 + e] = v[e];
  }
  __syncthreads();
  if (!wph) return;
  for (int r = ty; r < 64; r += 16) {
    const int b = b0 + r, a = a0 + tx * 4;
    T o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (T)tile[tx * 4 + e][r];
    T* dst = wph + (((int64_t)ph * cb + b) * 4 + tp) * ca + a;
    if constexpr (sizeof(T) == 2) *(uint2*)dst = *(const uint2*)o;
    else *(f4_t*)dst = *(const f4_t*)o;
  }
}

__global__ void __launch_bounds__(256) dropout_fill_kernel(uint8_t* mask, int64_t count, uint64_t seed, uint32_t thresh) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
    mask[i] = dropout_keep(seed, i, thresh);
}
__global__ void __launch_bounds__(256) mask_layout_kernel(const uint8_t* src, uint8_t* dst, int n, int c, int hw, int to_nhwc) {
  const int64_t total = (int64_t)n * c * hw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // i indexes the NHWC tensor
    const int ch = (int)(i % c);
    const int64_t t = i / c;
    const int p = (int)(t % hw);
    const int nn = (int)(t / hw);
    const int64_t j = ((int64_t)nn * c + ch) * hw + p;   // NCHW index
    if (to_nhwc) dst[i] = src[j];
    else dst[j] = src[i];
  }
}

inline int nblocks(int64_t count, int per_thread = 4) {
  int64_t b = (count + 256ll * per_thread - 1) / (256ll * per_thread);
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

// =================================================================================================
int op_bn_finalize(hipStream_t st, const float* partials, int rows, int c, int64_t count, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, float* scale, float* shift,
                   float* save_mean, float* save_invstd, int train, float momentum, float eps, int groups,
                   int64_t part_stride, int out_stride) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(c), dim3(256), 0, st, partials, rows, c, (float)count, gamma, beta,
                     running_mean, running_var, scale, shift, save_mean, save_invstd, train, momentum, eps, groups, part_stride,
                     out_stride);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

static uint32_t dropout_thresh(float p) { return gi_dropout_thresh(p); }

int op_bn_apply(hipStream_t st, int dtype, const void* x, void* y, int64_t pixels, int c, int ldy, int coffy,
                const float* scale, const float* shift, int act, const uint8_t* drop_mask, float drop_scale, int64_t pg,
                int gstride) {
  const int epc = dtype == GI_F16 ? 8 : 4;
  GI_REQUIRE(c % epc == 0 && gi_is_pow2(c / epc) && c / epc <= 256, "bn_apply: c=%d", c);
  const int grid = nblocks(pixels * (c / epc), 4);
  const bool g2 = pg > 0 && pg < pixels;
  GI_REQUIRE(!g2 || scale, "bn_apply: two groups need scale/shift");
  BnAccP fa = {};
#define GI_BN_APPLY(T, G) hipLaunchKernelGGL((bn_apply_kernel<T, G, false>), dim3(grid), dim3(256), 0, st, (const char*)x, (char*)y, pixels, c, ldy, coffy, scale, shift, act, (uint8_t*)drop_mask, drop_scale, pg, gstride, fa, (uint64_t)0, (uint32_t)0, (const u4_t*)nullptr, (u4_t*)nullptr, (int64_t)0)
  if (dtype == GI_F16) { if (g2) GI_BN_APPLY(half_t, true); else GI_BN_APPLY(half_t, false); }
  else { if (g2) GI_BN_APPLY(float, true); else GI_BN_APPLY(float, false); }
#undef GI_BN_APPLY
  GI_LAUNCH_CHECK();
  return GI_OK;
}

static void fill_acc_params(BnAccP& fa, const BnAccArgs& b) { gi_fill_acc_params(fa, b); }

int op_bn_finalize_acc(hipStream_t st, int c, const BnAccArgs& b) {
  GI_REQUIRE(b.groups == 1 || b.groups == 2, "bn_finalize_acc: groups=%d", b.groups);
  BnAccP fa;
  fill_acc_params(fa, b);
  hipLaunchKernelGGL(bn_finalize_acc_kernel, dim3((c + 255) / 256), dim3(256), 0, st, fa, c);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int op_bn_apply_acc(hipStream_t st, int dtype, const void* x, void* y, int64_t pixels, int c, int ldy, int coffy, int act,
                    uint8_t* drop_mask, float drop_scale, uint64_t drop_seed, float drop_p, const BnAccArgs& b,
                    const void* side_src, void* side_dst, int64_t side_bytes) {
  GI_REQUIRE(side_bytes % 16 == 0 && (side_bytes == 0 || (side_src && side_dst)), "bn_apply_acc: side copy of %lld bytes", (long long)side_bytes);
  const int epc = dtype == GI_F16 ? 8 : 4;
  GI_REQUIRE(c % epc == 0 && gi_is_pow2(c / epc) && c / epc <= 256, "bn_apply_acc: c=%d", c);
  GI_REQUIRE(b.groups == 1 || b.groups == 2, "bn_apply_acc: groups=%d", b.groups);
  GI_REQUIRE(pixels % b.groups == 0 && b.count == pixels / b.groups, "bn_apply_acc: %lld pixels, %d populations of %lld", (long long)pixels,
             b.groups, (long long)b.count);
  // fewer, fatter blocks than the plain pass: every block derives the affine maps of all channels first
  int grid = nblocks(pixels * (c / epc), 8);
  // at most the 1024 workgroups that are resident at once (126 VGPRs: four per CU): a second round would derive the affine maps again
  // (the critic's conv2 pass at 64 images: 2048 -> 1024 workgroups, 35.5 -> 33.4 us)
  if (grid > 1024) grid = 1024;
  const bool g2 = b.groups == 2;
  BnAccP fa;
  fill_acc_params(fa, b);
  const size_t lds = (size_t)b.groups * 2 * c * sizeof(float);
  const uint32_t thresh = drop_p > 0.f ? dropout_thresh(drop_p) : 0u;
#define GI_BN_APPLY(T, G) hipLaunchKernelGGL((bn_apply_kernel<T, G, true>), dim3(grid), dim3(256), lds, st, (const char*)x, (char*)y, pixels, c, ldy, coffy, (const float*)nullptr, (const float*)nullptr, act, drop_mask, drop_scale, (int64_t)b.count, 0, fa, drop_seed, thresh, (const u4_t*)side_src, (u4_t*)side_dst, side_bytes / 16)
  if (dtype == GI_F16) { if (g2) GI_BN_APPLY(half_t, true); else GI_BN_APPLY(half_t, false); }
  else { if (g2) GI_BN_APPLY(float, true); else GI_BN_APPLY(float, false); }
#undef GI_BN_APPLY
  GI_LAUNCH_CHECK();
  return GI_OK;
}

static int rows_per_block_for(int64_t pixels, int* blocks_out) {
  int64_t rpb = (pixels + 1023) / 1024;
  if (rpb < 32) rpb = 32;
  *blocks_out = (int)((pixels + rpb - 1) / rpb);
  return (int)rpb;
}

int op_col_stats(hipStream_t st, int dtype, const void* x, int64_t pixels, int c, float* partials, int* rows_out) {
  const int epc = dtype == GI_F16 ? 8 : 4;
  const int Q = c / epc;
  GI_REQUIRE(c % epc == 0 && gi_is_pow2(Q) && Q <= 256, "col_stats: c=%d unsupported", c);
  int blocks;
  const int rpb = rows_per_block_for(pixels, &blocks);
  if (dtype == GI_F16)
    hipLaunchKernelGGL(col_stats_kernel<half_t>, dim3(blocks), dim3(256), 0, st, (const char*)x, pixels, c, partials, rpb);
  else
    hipLaunchKernelGGL(col_stats_kernel<float>, dim3(blocks), dim3(256), 0, st, (const char*)x, pixels, c, partials, rpb);
  GI_LAUNCH_CHECK();
  *rows_out = blocks;
  return GI_OK;
}

int op_bwd_rows_per_block(int64_t pixels) {
  int blocks;
  return rows_per_block_for(pixels, &blocks);
}

static int launch_bwd_reduce(hipStream_t st, int dtype, const BwdP& p, int blocks) {
  const bool need_y = p.g2 != nullptr || p.act != GI_ACT_NONE;
  const bool load_y = need_y && !p.scale;
  const int in = (p.g1 ? 1 : 0) | (p.g2 ? 2 : 0);
  GI_REQUIRE(in != 0, "act_bn_bwd: no upstream gradient");
#define GI_RED(T, IN_, LY_) hipLaunchKernelGGL((act_bn_bwd_reduce_kernel<T, IN_, LY_>), dim3(blocks), dim3(256), 0, st, p)
#define GI_RED_T(IN_, LY_) do { if (dtype == GI_F16) GI_RED(half_t, IN_, LY_); else GI_RED(float, IN_, LY_); } while (0)
  if (in == 1) { if (load_y) GI_RED_T(1, true); else GI_RED_T(1, false); }
  else if (in == 2) { if (load_y) GI_RED_T(2, true); else GI_RED_T(2, false); }
  else { if (load_y) GI_RED_T(3, true); else GI_RED_T(3, false); }
#undef GI_RED_T
#undef GI_RED
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int op_act_bn_bwd(hipStream_t st, int dtype, const ActBnBwdArgs& a) {
  const int epc = dtype == GI_F16 ? 8 : 4;
  const int Q = a.c / epc;
  GI_REQUIRE(a.c % epc == 0 && gi_is_pow2(Q) && Q <= 256, "act_bn_bwd: c=%d unsupported", a.c);
  BwdP p;
  p.g1 = (const char*)a.g1; p.ldg1 = a.ldg1; p.coffg1 = a.coffg1;
  p.g2 = (const char*)a.g2; p.ldg2 = a.ldg2; p.coffg2 = a.coffg2;
  p.y = (const char*)a.y; p.ldy = a.ldy; p.coffy = a.coffy;
  p.x = (const char*)a.x; p.dx = (char*)a.dx;
  p.pixels = a.pixels; p.c = a.c; p.act = a.act; p.drop_scale = a.drop_scale;
  p.gamma = a.gamma; p.mean = a.save_mean; p.inv = a.save_invstd;
  p.partials = a.partials; p.sums = a.sums;
  p.scale = a.has_bn ? a.fwd_scale : nullptr; p.shift = a.has_bn ? a.fwd_shift : nullptr;
  const int groups = (a.groups == 2 && a.has_bn && !a.eval_bn) ? 2 : 1;
  GI_REQUIRE(a.groups <= 1 || groups == 2, "act_bn_bwd: groups=%d needs a train-mode BatchNorm", a.groups);
  GI_REQUIRE(a.pixels % groups == 0, "act_bn_bwd: %lld pixels in %d groups", (long long)a.pixels, groups);
  p.pg = a.pixels / groups; p.stat_stride = a.stat_stride;
  int blocks = 0;   // per population
  p.rows_per_block = rows_per_block_for(p.pg, &blocks);
  const int grid2 = nblocks(a.pixels * Q, 4);
  p.acc = nullptr; p.acc_reps = a.acc_reps > 0 ? a.acc_reps : 1; p.dgamma = a.dgamma; p.dbeta = a.dbeta; p.inv_loss_scale = a.inv_loss_scale; p.invM = 1.f / (float)p.pg;
  p.zero_next = nullptr; p.zero_words = 0;
  int small_max = gi_opt(GI_OPT_BN_BWD_SMALL);   // GI_BN_BWD_SMALL: largest pixel count served by the one-launch kernel (0: off)
  if (small_max > 2048) small_max = 2048;
  if (a.has_bn && !a.eval_bn && groups == 1 && a.pixels <= small_max && !a.reduce_done) {
    p.zero_next = a.acc ? a.zero_next : nullptr; p.zero_words = a.acc ? a.zero_words : 0;   // (keeps the accumulator ping-pong of net.hip consistent)
    const int nit = (int)((a.pixels + 255) / 256);
#define GI_SMALL(T) do { if (nit <= 1) hipLaunchKernelGGL((act_bn_bwd_small_kernel<T, 1>), dim3(Q), dim3(256), 0, st, p); \
      else if (nit <= 2) hipLaunchKernelGGL((act_bn_bwd_small_kernel<T, 2>), dim3(Q), dim3(256), 0, st, p); \
      else if (nit <= 4) hipLaunchKernelGGL((act_bn_bwd_small_kernel<T, 4>), dim3(Q), dim3(256), 0, st, p); \
      else hipLaunchKernelGGL((act_bn_bwd_small_kernel<T, 8>), dim3(Q), dim3(256), 0, st, p); } while (0)
    if (dtype == GI_F16) GI_SMALL(half_t); else GI_SMALL(float);
#undef GI_SMALL
    GI_LAUNCH_CHECK();
    return GI_OK;
  }
  if (a.has_bn && a.acc) {
    // exact accumulators: reduce pass (train mode only) + apply pass, no sums launch in between
    // (at most 1024 workgroups: with 214 - 250 VGPRs two are resident per CU, and every round of workgroups reads the accumulators and
    //  derives its coefficients again - the critic's conv2-level pass 38.5 -> 34.5 us; fewer, fatter workgroups on the smaller passes
    //  measured slower)
    int grid3 = nblocks(a.pixels * Q, gi_tune("GI_BWD_APPLY_CPT", 8));
    if (grid3 > 1024) grid3 = 1024;
    const size_t lds = (size_t)groups * 5 * a.c * sizeof(float);
    if (!a.eval_bn) {
      GI_REQUIRE(groups == 1 || p.pg % p.rows_per_block == 0, "act_bn_bwd: %lld pixels per group not a multiple of %d rows",
                 (long long)p.pg, p.rows_per_block);
      p.acc = a.acc;
      if (!a.reduce_done) {
        GI_TRY(launch_bwd_reduce(st, dtype, p, blocks * groups));
      }
      p.zero_next = a.zero_next; p.zero_words = a.zero_words;
    } else {
      p.dgamma = nullptr; p.dbeta = nullptr;
    }
    const bool need_y = a.g2 != nullptr || a.act != GI_ACT_NONE;
    const bool load_y = need_y && !p.scale;
    const int in = (a.g1 ? 1 : 0) | (a.g2 ? 2 : 0);
    // the input combinations the networks produce (critic: g1; encoder: g1 + g2; decoder: g2, with the saved activation where
    // dropout follows the norm) take the kernel built for bytes in flight; anything else the generic one
#define GI_APPLY_ACC(T, IN_, LY_, G2_) hipLaunchKernelGGL((act_bn_bwd_apply_acc_kernel<T, IN_, LY_, G2_>), dim3(grid3), dim3(256), lds, st, p)
#define GI_APPLY_ACC_T(IN_, LY_, G2_) do { if (dtype == GI_F16) GI_APPLY_ACC(half_t, IN_, LY_, G2_); else GI_APPLY_ACC(float, IN_, LY_, G2_); } while (0)
    if (in == 1 && !load_y && groups == 2) GI_APPLY_ACC_T(1, false, true);
    else if (in == 1 && !load_y && groups == 1) GI_APPLY_ACC_T(1, false, false);
    else if (in == 3 && !load_y && groups == 1) GI_APPLY_ACC_T(3, false, false);
    else if (in == 2 && !load_y && groups == 1) GI_APPLY_ACC_T(2, false, false);
    else if (in == 2 && load_y && groups == 1) GI_APPLY_ACC_T(2, true, false);
    else if (groups == 2) {
      if (dtype == GI_F16) hipLaunchKernelGGL((act_bn_bwd_apply_kernel<half_t, 2, true>), dim3(grid3), dim3(256), lds, st, p);
      else hipLaunchKernelGGL((act_bn_bwd_apply_kernel<float, 2, true>), dim3(grid3), dim3(256), lds, st, p);
    } else {
      if (dtype == GI_F16) hipLaunchKernelGGL((act_bn_bwd_apply_kernel<half_t, 2, false>), dim3(grid3), dim3(256), lds, st, p);
      else hipLaunchKernelGGL((act_bn_bwd_apply_kernel<float, 2, false>), dim3(grid3), dim3(256), lds, st, p);
    }
#undef GI_APPLY_ACC_T
#undef GI_APPLY_ACC
  } else if (a.has_bn && a.eval_bn) {
    // running-statistics BatchNorm is a per-channel affine map: the batch-mean terms vanish (sums = 0)
    hipLaunchKernelGGL(bwd_sums_kernel, dim3(a.c), dim3(256), 0, st, a.partials, 0, a.c, a.sums, (float*)nullptr, (float*)nullptr, 0.f, 1,
                       a.gamma, a.save_mean, a.save_invstd, p.scale, p.shift, 0, 0.f);
    GI_LAUNCH_CHECK();
    if (dtype == GI_F16) hipLaunchKernelGGL((act_bn_bwd_apply_kernel<half_t, 1, false>), dim3(grid2), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((act_bn_bwd_apply_kernel<float, 1, false>), dim3(grid2), dim3(256), 0, st, p);
  } else if (a.has_bn) {
    GI_REQUIRE(groups == 1 || p.pg % p.rows_per_block == 0, "act_bn_bwd: %lld pixels per group not a multiple of %d rows",
               (long long)p.pg, p.rows_per_block);
    GI_TRY(launch_bwd_reduce(st, dtype, p, blocks * groups));
    hipLaunchKernelGGL(bwd_sums_kernel, dim3(a.c), dim3(256), 0, st, a.partials, blocks, a.c, a.sums, a.dgamma, a.dbeta, a.inv_loss_scale,
                       groups, a.gamma, a.save_mean, a.save_invstd, p.scale, p.shift, a.stat_stride, 1.f / (float)p.pg);
    GI_LAUNCH_CHECK();
    if (groups == 2) {
      if (dtype == GI_F16) hipLaunchKernelGGL((act_bn_bwd_apply_kernel<half_t, 1, true>), dim3(grid2), dim3(256), 0, st, p);
      else hipLaunchKernelGGL((act_bn_bwd_apply_kernel<float, 1, true>), dim3(grid2), dim3(256), 0, st, p);
    } else {
      if (dtype == GI_F16) hipLaunchKernelGGL((act_bn_bwd_apply_kernel<half_t, 1, false>), dim3(grid2), dim3(256), 0, st, p);
      else hipLaunchKernelGGL((act_bn_bwd_apply_kernel<float, 1, false>), dim3(grid2), dim3(256), 0, st, p);
    }
  } else {
    if (dtype == GI_F16) hipLaunchKernelGGL((act_bn_bwd_apply_kernel<half_t, 0, false>), dim3(grid2), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((act_bn_bwd_apply_kernel<float, 0, false>), dim3(grid2), dim3(256), 0, st, p);
  }
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int op_in_forward(hipStream_t st, int dtype, const void* x, void* y, int n, int hw, int c, int ldy, int coffy, int act,
                  const uint8_t* drop_mask, float drop_scale, float eps, float* stats) {
  const int epc = dtype == GI_F16 ? 8 : 4;
  GI_REQUIRE(c % epc == 0 && n >= 1 && hw >= 1, "in_forward: n=%d hw=%d c=%d", n, hw, c);
  const dim3 grid((unsigned)n * (c / epc));
  if (dtype == GI_F16) hipLaunchKernelGGL(in_forward_kernel<half_t>, grid, dim3(256), 0, st, (const char*)x, (char*)y, hw, c, ldy, coffy, act, drop_mask, drop_scale, eps, stats);
  else hipLaunchKernelGGL(in_forward_kernel<float>, grid, dim3(256), 0, st, (const char*)x, (char*)y, hw, c, ldy, coffy, act, drop_mask, drop_scale, eps, stats);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int op_act_in_bwd(hipStream_t st, int dtype, const ActBnBwdArgs& a, int n, int hw, const float* stats) {
  const int epc = dtype == GI_F16 ? 8 : 4;
  GI_REQUIRE(a.c % epc == 0 && (int64_t)n * hw == a.pixels && a.x && stats, "act_in_bwd: n=%d hw=%d pixels=%lld c=%d", n, hw, (long long)a.pixels, a.c);
  BwdP p;
  memset(&p, 0, sizeof(p));
  p.g1 = (const char*)a.g1; p.ldg1 = a.ldg1; p.coffg1 = a.coffg1;
  p.g2 = (const char*)a.g2; p.ldg2 = a.ldg2; p.coffg2 = a.coffg2;
  p.y = (const char*)a.y; p.ldy = a.ldy; p.coffy = a.coffy;
  p.x = (const char*)a.x; p.dx = (char*)a.dx;
  p.pixels = a.pixels; p.c = a.c; p.act = a.act; p.drop_scale = a.drop_scale;
  const dim3 grid((unsigned)n * (a.c / epc));
  if (dtype == GI_F16) hipLaunchKernelGGL(act_in_bwd_kernel<half_t>, grid, dim3(256), 0, st, p, hw, stats);
  else hipLaunchKernelGGL(act_in_bwd_kernel<float>, grid, dim3(256), 0, st, p, hw, stats);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int op_bias_grad(hipStream_t st, int dtype, const void* dz, int64_t pixels, int c, float scale, float* dbias, float* partials) {
  int rows = 0;
  GI_TRY(op_col_stats(st, dtype, dz, pixels, c, partials, &rows));
  hipLaunchKernelGGL(bias_grad_kernel, dim3(c), dim3(256), 0, st, partials, rows, c, scale, dbias);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int op_pack_weights(hipStream_t st, int dtype, const float* w, int ca, int cb, void* w_packed, void* w_phase) {
  const int64_t count = (int64_t)ca * 16 * cb;
  if (w_packed) GI_TRY(op_convert(st, dtype, w, w_packed, count));
  if (w_phase) {
    dim3 grid((ca + 31) / 32, (cb + 31) / 32, 16);
    if (dtype == GI_F16) hipLaunchKernelGGL(pack_phase_kernel<half_t>, grid, dim3(256), 0, st, w, (half_t*)w_phase, ca, cb);
    else hipLaunchKernelGGL(pack_phase_kernel<float>, grid, dim3(256), 0, st, w, (float*)w_phase, ca, cb);
    GI_LAUNCH_CHECK();
  }
  return GI_OK;
}

int op_pack_weights_batch(hipStream_t st, int dtype, PackJobs& P) {
  if (P.n <= 0) return GI_OK;
  GI_REQUIRE(P.n <= 16, "pack_weights_batch: %d layers", P.n);
  bool all64 = true;
  for (int i = 0; i < P.n; ++i) all64 = all64 && P.j[i].ca % 64 == 0 && P.j[i].cb % 64 == 0;
  const int ts = all64 ? 64 : 32;
  int tiles = 0;
  for (int i = 0; i < P.n; ++i) {
    P.j[i].tile0 = tiles;
    tiles += ((P.j[i].ca + ts - 1) / ts) * ((P.j[i].cb + ts - 1) / ts) * 16;
  }
  if (all64) {
    if (dtype == GI_F16) hipLaunchKernelGGL(pack_batch64_kernel<half_t>, dim3(tiles), dim3(256), 0, st, P);
    else hipLaunchKernelGGL(pack_batch64_kernel<float>, dim3(tiles), dim3(256), 0, st, P);
  } else if (dtype == GI_F16) hipLaunchKernelGGL(pack_batch_kernel<half_t>, dim3(tiles), dim3(256), 0, st, P);
  else hipLaunchKernelGGL(pack_batch_kernel<float>, dim3(tiles), dim3(256), 0, st, P);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int op_convert(hipStream_t st, int dtype, const float* src, void* dst, int64_t count) {
  if (dtype == GI_F16) hipLaunchKernelGGL(convert_kernel<half_t>, dim3(nblocks(count)), dim3(256), 0, st, src, (half_t*)dst, count);
  else hipLaunchKernelGGL(convert_kernel<float>, dim3(nblocks(count)), dim3(256), 0, st, src, (float*)dst, count);
  GI_LAUNCH_CHECK();
  return GI_OK;
}
int op_convert_back(hipStream_t st, int dtype, const void* src, float* dst, int64_t count) {
  if (dtype == GI_F16) hipLaunchKernelGGL(convert_back_kernel<half_t>, dim3(nblocks(count)), dim3(256), 0, st, (const half_t*)src, dst, count);
  else hipLaunchKernelGGL(convert_back_kernel<float>, dim3(nblocks(count)), dim3(256), 0, st, (const float*)src, dst, count);
  GI_LAUNCH_CHECK();
  return GI_OK;
}
int op_tanh_bwd(hipStream_t st, const float* dy, const float* y, float* dx, int64_t count, float scale) {
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(nblocks(count)), dim3(256), 0, st, dy, y, dx, count, scale);
  GI_LAUNCH_CHECK();
  return GI_OK;
}
int op_fill_dropout(hipStream_t st, uint8_t* mask, int64_t count, uint64_t seed, float p) {
  const uint32_t thresh = dropout_thresh(p);
  hipLaunchKernelGGL(dropout_fill_kernel, dim3(nblocks(count)), dim3(256), 0, st, mask, count, seed, thresh);
  GI_LAUNCH_CHECK();
  return GI_OK;
}
int op_mask_nchw_to_nhwc(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int c, int hw, int to_nhwc) {
  hipLaunchKernelGGL(mask_layout_kernel, dim3(nblocks((int64_t)n * c * hw)), dim3(256), 0, st, src, dst, n, c, hw, to_nhwc);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int op_mul_slope(hipStream_t st, int dtype, const void* tin, const void* a, void* tout, int64_t count) {
  const int epc = dtype == GI_F16 ? 8 : 4;
  GI_REQUIRE(count % epc == 0, "mul_slope: count must be a multiple of %d", epc);
  if (dtype == GI_F16) hipLaunchKernelGGL(mul_slope_kernel<half_t>, dim3(nblocks(count / epc, 2)), dim3(256), 0, st, (const char*)tin, (const char*)a, (char*)tout, count / epc);
  else hipLaunchKernelGGL(mul_slope_kernel<float>, dim3(nblocks(count / epc, 2)), dim3(256), 0, st, (const char*)tin, (const char*)a, (char*)tout, count / epc);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int op_bn_tangent_inject(hipStream_t st, int dtype, const void* dta, const void* y, const void* tx, const void* x, void* dxp,
                         int64_t pixels, int c, const float* gamma, const float* mean, const float* inv, float* dgamma,
                         float* partials, float* sums) {
  const int epc = dtype == GI_F16 ? 8 : 4;
  const int Q = c / epc;
  GI_REQUIRE(c % epc == 0 && gi_is_pow2(Q) && Q <= 256, "bn_tangent_inject: c=%d unsupported", c);
  InjP p;
  p.dta = (const char*)dta; p.y = (const char*)y; p.tx = (const char*)tx; p.x = (const char*)x; p.dxp = (char*)dxp;
  p.pixels = pixels; p.c = c; p.gamma = gamma; p.mean = mean; p.inv = inv; p.partials = partials; p.sums = sums; p.dgamma = dgamma;
  int blocks = 0;
  p.rows_per_block = rows_per_block_for(pixels, &blocks);
  if (dtype == GI_F16) hipLaunchKernelGGL(bn_inject_reduce_kernel<half_t>, dim3(blocks), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(bn_inject_reduce_kernel<float>, dim3(blocks), dim3(256), 0, st, p);
  GI_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_inject_sums_kernel, dim3(c), dim3(256), 0, st, partials, blocks, c, sums);
  GI_LAUNCH_CHECK();
  const int grid2 = nblocks(pixels * Q, 2);
  if (dtype == GI_F16) hipLaunchKernelGGL(bn_inject_apply_kernel<half_t>, dim3(grid2), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(bn_inject_apply_kernel<float>, dim3(grid2), dim3(256), 0, st, p);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int op_gp_direction(hipStream_t st, const float* g, int n, int64_t hw, float lam, float* sumsq, float* v, float* penalty) {
  hipLaunchKernelGGL(sample_sumsq_kernel, dim3(n), dim3(256), 0, st, g, hw, sumsq);
  GI_LAUNCH_CHECK();
  hipLaunchKernelGGL(gp_direction_kernel, dim3(nblocks(hw) > 64 ? 64 : nblocks(hw), n), dim3(256), 0, st, g, sumsq, n, hw, lam, v, penalty);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

// ---- C-ABI entry points that are pure elementwise ops -------------------------------------------
extern "C" {

int gi_mask_apply(gi_ctx* ctx, const float* ground, const float* mask, float* mask_c, float* masked, int64_t count, int do_ceil) {
  hipLaunchKernelGGL(mask_apply_kernel, dim3(nblocks(count)), dim3(256), 0, ctx->stream, ground, mask, mask_c, masked, count, do_ceil);
  GI_LAUNCH_CHECK();
  return GI_OK;
}
int gi_mask_composite(gi_ctx* ctx, const float* masked, const float* gen, const float* mask_c, float* inpainted, int64_t count) {
  hipLaunchKernelGGL(mask_composite_kernel, dim3(nblocks(count)), dim3(256), 0, ctx->stream, masked, gen, mask_c, inpainted, count);
  GI_LAUNCH_CHECK();
  return GI_OK;
}
int gi_mul(gi_ctx* ctx, const float* a, const float* b, float* out, int64_t count) {
  hipLaunchKernelGGL(mul_kernel, dim3(nblocks(count)), dim3(256), 0, ctx->stream, a, b, out, count);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int gi_interpolate(gi_ctx* ctx, const float* real, const float* fake, const float* eps, int n, int64_t hw, float* out) {
  GI_REQUIRE(ctx && real && fake && eps && out && n > 0, "interpolate: bad argument");
  hipLaunchKernelGGL(interpolate_kernel, dim3(nblocks(hw) > 64 ? 64 : nblocks(hw), n), dim3(256), 0, ctx->stream, real, fake, eps, hw, out);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int gi_add(gi_ctx* ctx, const float* a, const float* b, float* out, int64_t count, float alpha) {
  hipLaunchKernelGGL(add_kernel, dim3(nblocks(count)), dim3(256), 0, ctx->stream, a, b, out, count, alpha);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

static int loss_generic(gi_ctx* ctx, const float* a, const float* b, const float* m, int64_t count, int pkind, int mode,
                        int gkind, float eps, float* loss_out, float* grad, float gscale, float* scratch) {
  GI_REQUIRE(scratch != nullptr, "loss: scratch is required (>= 4096 floats)");
  const int nb = nblocks(count) > 1024 ? 1024 : nblocks(count);
  hipLaunchKernelGGL(loss_partial_kernel, dim3(nb), dim3(256), 0, ctx->stream, a, b, m, count, pkind, scratch);
  GI_LAUNCH_CHECK();
  // loss_out needs two floats (value, denominator): stage in scratch tail then copy the value
  float* tmp = scratch + 2048;
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, ctx->stream, scratch, nb, (double)count, m ? 1 : 0, mode, eps, tmp);
  GI_LAUNCH_CHECK();
  GI_HIP(hipMemcpyAsync(loss_out, tmp, sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  if (grad) {
    hipLaunchKernelGGL(loss_grad_kernel, dim3(nblocks(count)), dim3(256), 0, ctx->stream, a, b, m, count, gkind, tmp, gscale, grad);
    GI_LAUNCH_CHECK();
  }
  return GI_OK;
}

int gi_loss_l1(gi_ctx* ctx, const float* a, const float* b, int64_t count, float* loss_out, float* grad_a, float gscale, float* scratch) {
  return loss_generic(ctx, a, b, nullptr, count, 0, 0, 0, 0.f, loss_out, grad_a, gscale, scratch);
}
int gi_loss_mse(gi_ctx* ctx, const float* a, const float* b, int64_t count, float* loss_out, float* grad_a, float gscale, float* scratch) {
  return loss_generic(ctx, a, b, nullptr, count, 1, 0, 1, 0.f, loss_out, grad_a, gscale, scratch);
}
int gi_loss_rmse(gi_ctx* ctx, const float* a, const float* b, int64_t count, float eps, float* loss_out, float* grad_a, float gscale, float* scratch) {
  return loss_generic(ctx, a, b, nullptr, count, 1, 1, 2, eps, loss_out, grad_a, gscale, scratch);
}
int gi_loss_local(gi_ctx* ctx, const float* yhat, const float* y, const float* mask, int64_t count, int kind, float* loss_out,
                  float* grad_yhat, float gscale, float* scratch) {
  GI_REQUIRE(kind >= 0 && kind <= 2, "loss_local: kind=%d", kind);
  return loss_generic(ctx, yhat, y, mask, count, kind == 0 ? 0 : 1, kind == 2 ? 1 : 0, kind, 1e-16f, loss_out, grad_yhat, gscale, scratch);
}
int gi_loss_adv(gi_ctx* ctx, const float* pred, int n, int kind, float target, float* loss_out, float* grad_pred, float gscale) {
  GI_REQUIRE(kind >= 0 && kind <= 2 && n > 0, "loss_adv: kind=%d n=%d", kind, n);
  hipLaunchKernelGGL(adv_loss_kernel, dim3(1), dim3(256), 0, ctx->stream, pred, n, kind, target, loss_out, grad_pred, gscale, 0.f, (float*)nullptr, 0.f);
  GI_LAUNCH_CHECK();
  return GI_OK;
}
int gi_loss_adv_pair(gi_ctx* ctx, const float* pred2, int n_each, int kind, float target_a, float target_b, float* loss_a, float* loss_b,
                     float* grad_pred2, float gscale_a, float gscale_b) {
  GI_REQUIRE(ctx && pred2 && loss_a && loss_b && kind >= 0 && kind <= 2 && n_each > 0, "loss_adv_pair: kind=%d n=%d", kind, n_each);
  hipLaunchKernelGGL(adv_loss_kernel, dim3(2), dim3(256), 0, ctx->stream, pred2, n_each, kind, target_a, loss_a, grad_pred2, gscale_a, target_b, loss_b,
                     gscale_b);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int gi_adam_step(gi_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
                 float eps, int step, float grad_scale) {
  return gi_adam_step_guarded(ctx, p, g, m, v, count, lr, beta1, beta2, eps, step, grad_scale, nullptr);
}
int gi_adam_step_guarded(gi_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
                         float eps, int step, float grad_scale, const int* guard) {
  return gi_adam_step_guarded2(ctx, p, g, m, v, count, lr, beta1, beta2, eps, step, -1, grad_scale, guard);
}
int gi_adam_step_guarded2(gi_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
                          float eps, int step, int skipped_seen, float grad_scale, const int* guard) {
  return gi_adam_step_scan(ctx, p, g, m, v, count, lr, beta1, beta2, eps, step, skipped_seen, grad_scale, (int*)guard, 0, 0);
}
int gi_adam_step_scan(gi_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
                      float eps, int step, int skipped_seen, float grad_scale, int* guard, int word, int finish) {
  GI_REQUIRE(word == 0 || ((word == 2 || word == 3) && guard), "adam: scan word %d", word);
  GI_REQUIRE(step >= 1, "adam: step=%d must be >= 1", step);
  // torch.optim forms 1 - beta and lr / bias_correction1 from Python floats (doubles) and rounds once. The hyper-parameters
  // arrive here as C floats (0.999f = 0.99900001...): recover the decimal the caller wrote (7 significant digits) first
  auto py = [](float x) { char b[32]; snprintf(b, sizeof b, "%.7g", (double)x); return atof(b); };
  const double b1 = py(beta1), b2 = py(beta2), lrd = py(lr);
  const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(nblocks(count)), dim3(256), 0, ctx->stream, p, g, m, v, count, (float)(lrd / bc1), beta1, beta2, eps,
                     (float)(1.0 - b1), (float)(1.0 - b2), (float)sqrt(bc2), grad_scale, (int*)guard, step, guard ? skipped_seen : -1, lrd, b1, b2, word, finish);
  GI_LAUNCH_CHECK();
  return GI_OK;
}
int gi_rmsprop_step(gi_ctx* ctx, float* p, const float* g, float* sq, int64_t count, float lr, float alpha, float eps, float clamp,
                    float grad_scale) {
  return gi_rmsprop_step_guarded(ctx, p, g, sq, count, lr, alpha, eps, clamp, grad_scale, nullptr);
}
int gi_rmsprop_step_guarded(gi_ctx* ctx, float* p, const float* g, float* sq, int64_t count, float lr, float alpha, float eps,
                            float clamp, float grad_scale, const int* guard) {
  return gi_rmsprop_step_scan(ctx, p, g, sq, count, lr, alpha, eps, clamp, grad_scale, (int*)guard, 0, 0);
}
int gi_rmsprop_step_scan(gi_ctx* ctx, float* p, const float* g, float* sq, int64_t count, float lr, float alpha, float eps,
                         float clamp, float grad_scale, int* guard, int word, int finish) {
  GI_REQUIRE(word == 0 || ((word == 2 || word == 3) && guard), "rmsprop: scan word %d", word);
  char b[32];
  snprintf(b, sizeof b, "%.7g", (double)alpha);          // the Python float the caller meant (0.99), see gi_adam_step_guarded
  const float oma = (float)(1.0 - atof(b));
  hipLaunchKernelGGL(rmsprop_kernel, dim3(nblocks(count)), dim3(256), 0, ctx->stream, p, g, sq, count, lr, alpha, oma, eps, clamp, grad_scale,
                     guard, word, finish);
  GI_LAUNCH_CHECK();
  return GI_OK;
}
int gi_check_finite_scan(gi_ctx* ctx, const float* g, int64_t count, int* flag3) { return gi_check_finite_scan_word(ctx, g, count, flag3, 2); }
int gi_check_finite_scan_word(gi_ctx* ctx, const float* g, int64_t count, int* flag4, int word) {
  GI_REQUIRE(ctx && g && flag4 && count > 0 && (word == 2 || word == 3), "check_finite_scan: bad argument");
  hipLaunchKernelGGL(check_finite_kernel, dim3(nblocks((count + 3) / 4, 2)), dim3(256), 0, ctx->stream, g, count, flag4, word);
  GI_LAUNCH_CHECK();
  return GI_OK;
}
int gi_check_finite_finish(gi_ctx* ctx, int* flag3) {
  GI_REQUIRE(ctx && flag3, "check_finite_finish: bad argument");
  hipLaunchKernelGGL(check_finite_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, flag3);
  GI_LAUNCH_CHECK();
  return GI_OK;
}
int gi_check_finite(gi_ctx* ctx, const float* g, int64_t count, int* flag3) {
  GI_TRY(gi_check_finite_scan(ctx, g, count, flag3));
  return gi_check_finite_finish(ctx, flag3);
}
int gi_clamp(gi_ctx* ctx, float* p, int64_t count, float lo, float hi) {
  hipLaunchKernelGGL(clamp_kernel, dim3(nblocks(count)), dim3(256), 0, ctx->stream, p, count, lo, hi);
  GI_LAUNCH_CHECK();
  return GI_OK;
}
int gi_grad_absmean(gi_ctx* ctx, const float* g, const int64_t* seg_off, const int64_t* seg_len, int nseg, float* out) {
  GI_REQUIRE(nseg > 0, "grad_absmean: nseg=%d", nseg);
  GI_HIP(hipMemsetAsync(out, 0, sizeof(float) * nseg, ctx->stream));
  hipLaunchKernelGGL(absmean_kernel, dim3(32, nseg), dim3(256), 0, ctx->stream, g, seg_off, seg_len, out);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

}  // extern "C"
