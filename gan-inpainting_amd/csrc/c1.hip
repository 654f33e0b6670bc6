// HBM-bound kernels for the layers whose large-resolution side has ONE channel:
//   generator d1  Conv2d(1->ngf)            (lib/models/networks.py:285, outermost block :296)
//   generator u1  ConvTranspose2d(2ngf->1)  (networks.py:293-297) + bias + tanh
//   discriminator Conv2d(1->64)             (networks.py:337)
// and their gradients. Weights are fp32 [c][16] (= [a][ky][kx][b] with b == 1). The image side is
// the public fp32 (n,1,H,W) tensor, the feature side NHWC of type T.
// These layers carry <1% of the FLOPs and ~45% of the activation bytes, so they are written as
// vectorised streaming kernels (16-byte feature accesses, weights in registers/LDS), not GEMMs.
#include "common.h"
#include "stat_acc.h"
#include "bn_acc.h"

namespace {

__device__ __forceinline__ float act_f(float v, int act) {
  if (act == GI_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == GI_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
  return v;
}

// ---- gather: out[p][c] = act(sum_tap img[n,2y-1+ky,2x-1+kx] * w[c][tap]) ---------------------
// One thread per (pixel, 8-channel group): 16 image taps (L1 broadcast among the pixel's threads),
// weights staged TRANSPOSED in LDS (swT[tap][c]) so the 8 channels of a tap are two 16-byte reads and
// the lanes of a wave (consecutive groups) hit consecutive LDS slots; one 16-byte store per thread.
template <typename T>
__global__ void __launch_bounds__(256) c1_gather_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                        char* out, int n, int Hs, int Ws, int c, int ldout,
                                                        int coffout, int act, float in_scale, const float* __restrict__ bias) {
  constexpr int G = 8;  // channels per thread
  extern __shared__ __attribute__((aligned(16))) float swT[];  // [16][c]
  for (int i = threadIdx.x; i < c * 16; i += 256) swT[(i & 15) * c + (i >> 4)] = w[i];
  __syncthreads();
  const int groups = c / G;
  const int lg = 31 - __builtin_clz(groups);   // groups is a power of two
  const int64_t total = (int64_t)n * Hs * Ws * groups;
  const int H = 2 * Hs, W = 2 * Ws;
  for (int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (int64_t)gridDim.x * 256) {
    const int grp = (int)(gid & (groups - 1));
    const int64_t pix = gid >> lg;
    const int x = (int)(pix % Ws);
    const int64_t t = pix / Ws;
    const int y = (int)(t % Hs);
    const int nn = (int)(t / Hs);
    const float* ip = img + (int64_t)nn * H * W;
    float o[G];
#pragma unroll
    for (int j = 0; j < G; ++j) o[j] = bias ? bias[grp * G + j] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int iy = 2 * y - 1 + ky;
      const bool yok = iy >= 0 && iy < H;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int ix = 2 * x - 1 + kx;
        const float v = (yok && ix >= 0 && ix < W) ? ip[(int64_t)iy * W + ix] * in_scale : 0.f;
        const float* wr = swT + (ky * 4 + kx) * c + grp * G;
        const f4_t w0 = *(const f4_t*)wr, w1 = *(const f4_t*)(wr + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = fmaf(v, w0[j], o[j]); o[4 + j] = fmaf(v, w1[j], o[4 + j]); }
      }
    }
#pragma unroll
    for (int j = 0; j < G; ++j) o[j] = act_f(o[j], act);
    T* dst = (T*)(out + (pix * ldout + coffout + grp * G) * (int64_t)sizeof(T));
    if constexpr (std::is_same<T, half_t>::value) {
      h8_t h;
#pragma unroll
      for (int j = 0; j < G; ++j) h[j] = (half_t)o[j];
      *(h8_t*)dst = h;
    } else {
      *(f4_t*)dst = f4_t{o[0], o[1], o[2], o[3]};
      *(f4_t*)(dst + 4) = f4_t{o[4], o[5], o[6], o[7]};
    }
  }
}

// Strip form of the gather (used when a row of Ws pixels splits evenly into strips of 256/groups
// pixels): a block walks strips; the 4 x (2S+2) image window of a strip is staged once in LDS
// (double-buffered, one barrier per strip) instead of 16 bounds-checked global loads per thread.
template <typename T>
__global__ void __launch_bounds__(256) c1_gather_strip_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                              char* out, int n, int Hs, int Ws, int c, int ldout,
                                                              int coffout, int act, float in_scale, const float* __restrict__ bias) {
  constexpr int G = 8;
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int groups = c / G;
  const int S = 256 / groups;            // pixels per strip
  const int TW = 2 * S + 2;              // staged columns
  float* swT = smem_f;                   // [16][c]
  float* tile = smem_f + 16 * c;         // [2][4][TW]
  for (int i = threadIdx.x; i < c * 16; i += 256) swT[(i & 15) * c + (i >> 4)] = w[i];
  const int H = 2 * Hs, W = 2 * Ws;
  const int spr = Ws / S;                                  // strips per row
  const int64_t nstrips = (int64_t)n * Hs * spr;
  const int grp = threadIdx.x % groups, pi = threadIdx.x / groups;
  auto stage = [&](int64_t strip, int buf) {
    const int sx = (int)(strip % spr);
    const int64_t t = strip / spr;
    const int y = (int)(t % Hs);
    const int nn = (int)(t / Hs);
    const float* ip = img + (int64_t)nn * H * W;
    for (int idx = threadIdx.x; idx < 4 * TW; idx += 256) {
      const int r = idx / TW, cc = idx % TW;
      const int iy = 2 * y - 1 + r, ix = 2 * sx * S - 1 + cc;
      tile[(buf * 4 + r) * TW + cc] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? ip[(int64_t)iy * W + ix] * in_scale : 0.f;
    }
  };
  int buf = 0;
  int64_t strip = blockIdx.x;
  if (strip < nstrips) stage(strip, 0);
  __syncthreads();
  for (; strip < nstrips; strip += gridDim.x) {
    const int64_t next = strip + gridDim.x;
    if (next < nstrips) stage(next, buf ^ 1);
    const float* tl = tile + buf * 4 * TW + 2 * pi;
    float o[G];
#pragma unroll
    for (int j = 0; j < G; ++j) o[j] = bias ? bias[grp * G + j] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const float v = tl[ky * TW + kx];
        const float* wr = swT + (ky * 4 + kx) * c + grp * G;
        const f4_t w0 = *(const f4_t*)wr, w1 = *(const f4_t*)(wr + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = fmaf(v, w0[j], o[j]); o[4 + j] = fmaf(v, w1[j], o[4 + j]); }
      }
#pragma unroll
    for (int j = 0; j < G; ++j) o[j] = act_f(o[j], act);
    const int64_t pix = strip * S + pi;      // strips enumerate pixels in (n, y, x) order
    T* dst = (T*)(out + (pix * ldout + coffout + grp * G) * (int64_t)sizeof(T));
    if constexpr (std::is_same<T, half_t>::value) {
      h8_t h;
#pragma unroll
      for (int j = 0; j < G; ++j) h[j] = (half_t)o[j];
      *(h8_t*)dst = h;
    } else {
      *(f4_t*)dst = f4_t{o[0], o[1], o[2], o[3]};
      *(f4_t*)(dst + 4) = f4_t{o[4], o[5], o[6], o[7]};
    }
    __syncthreads();
    buf ^= 1;
  }
}

// ---- fp16 gather on the matrix cores ---------------------------------------------------------------
// out[p][c] = act(sum_tap G[p][tap] * w[c][tap]) as D[c][pixel] = A[c][tap] * B[tap][pixel], K = 16 taps = one
// v_mfma_f32_16x16x16_f16: the weights (A) live in registers, each wave gathers the 16 taps of 16 consecutive pixels of
// one image row - lane (pixel lr, k-chunk kq) holds tap row ky = kq, i.e. the four image columns 2x-1 .. 2x+2 of image row
// 2y-1+kq: one 8-byte load for the aligned middle pair and two single loads, from clamped addresses with the padding
// zeroed by selects (no divergent branches; the group index is wave-uniform, so the (image, row, column) split is
// scalar 32-bit arithmetic). Every lane ends with 4 consecutive channels of its pixel per 16-channel tile.
// The first form of this kernel (K padded to 32, half the lanes loading 8 taps behind per-tap bounds branches, 64-bit
// divisions per group) spent 22 of its 39 us on the critic's conv1 in address arithmetic, measured with loads and stores
// switched off; the layer is 134 MB of stores.
// (Measured and dropped in round 4: the same launch also writing the copy of the image that the generator keeps for d1's weight
//  gradient - lanes kq = 1, 2 hold exactly the 2 x 2 input pixels under their output pixel - made this store-bound kernel 5.6 us
//  slower, what the separate 8.4 MB device copy costs; the copy now rides in a latency-bound pass instead: bn_apply's side copy.)
template <int MTC, int ACT>   // MTC = c / 16
__global__ void __launch_bounds__(256) c1_gather_mfma_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                             char* out, int n, int Hs, int Ws, int ldout, int coffout,
                                                             float in_scale, unsigned long long* __restrict__ bits) {
  __shared__ __attribute__((aligned(16))) char slab[4][2048];   // a wave's 16 pixels x 128 bytes, re-read pixel-major for the stores
  const int lane = threadIdx.x & 63;
  const int lr = lane & 15, kq = lane >> 4;
  // A[row lr of tile mt][k = 4*kq + j] = w[ch][tap 4*kq + j] with
  //   ch(mt, lr) = (mt >> 2) * 64 + ((mt >> 1) & 1) * 32 + (lr >> 2) * 8 + (mt & 1) * 4 + (lr & 3)
  // D row i of a tile lands in lane group kq = i >> 2, register r = i & 3, so after tiles 0,1 a lane owns the 8
  // consecutive channels kq * 8 .. + 7 of its pixel's first 32 and after tiles 2,3 those of the second 32 (the 16-byte
  // chunks kq and 4 + kq of the pixel's 128-byte row, staged through LDS below).
  h4_t af[MTC];
#pragma unroll
  for (int mt = 0; mt < MTC; ++mt) {
    const int ch = (mt >> 2) * 64 + ((mt >> 1) & 1) * 32 + (lr >> 2) * 8 + (mt & 1) * 4 + (lr & 3);
    const f4_t v = *(const f4_t*)(w + ch * 16 + kq * 4);
    af[mt] = h4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
  }
  const int H = 2 * Hs, W = 2 * Ws;
  const int gpr = Ws >> 4;                       // groups per small-grid row (Ws % 16 == 0)
  const int ngroups = n * Hs * gpr;              // < 2^31 (checked by the host)
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
  const int nwaves = (int)gridDim.x * 4;
  auto load_b = [&](int g, float (&bv)[4]) {
    const int rowi = g / gpr;                    // wave-uniform
    const int x = (g - rowi * gpr) * 16 + lr;
    const int nn = rowi / Hs, y = rowi - nn * Hs;
    const int iy = 2 * y - 1 + kq;
    const bool rok = iy >= 0 && iy < H;
    const float* rp = img + ((int64_t)nn * H + (rok ? iy : 0)) * W;
    const int ix0 = 2 * x - 1;                   // -1 only at x = 0; ix0 + 3 = W only at x = Ws - 1
    const float a0 = rp[ix0 < 0 ? 0 : ix0];
    const float2 mid = *(const float2*)(rp + ix0 + 1);
    const float a3 = rp[ix0 + 3 < W ? ix0 + 3 : W - 1];
    bv[0] = (rok && ix0 >= 0) ? a0 : 0.f;
    bv[1] = rok ? mid.x : 0.f;
    bv[2] = rok ? mid.y : 0.f;
    bv[3] = (rok && ix0 + 3 < W) ? a3 : 0.f;
  };
  float bv[4], bn[4];
  int g = wave;
  if (g < ngroups) load_b(g, bv);
  for (; g < ngroups; g += nwaves) {
    if (g + nwaves < ngroups) load_b(g + nwaves, bn);
    const h4_t bf = {(half_t)(bv[0] * in_scale), (half_t)(bv[1] * in_scale), (half_t)(bv[2] * in_scale), (half_t)(bv[3] * in_scale)};
#pragma unroll
    for (int mq = 0; mq < MTC / 4; ++mq) {
      h8_t o[2];
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) {
        f4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(af[mq * 4 + m4], bf, acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[r];
          if (ACT == GI_ACT_RELU) v = fmaxf(v, 0.f);
          if (ACT == GI_ACT_LRELU) v = fmaxf(v, 0.2f * v);      // = v > 0 ? v : 0.2 v
          o[m4 >> 1][(m4 & 1) * 4 + r] = (half_t)v;
        }
      }
      if (MTC == 4 && bits) {
        // sign word of the pixel (bit c = [out[p][c] > 0], 64 channels): what the fused activation backward of the next layer's
        // input-gradient GEMM needs of this tensor (IgemmArgs::mask_bits: 8 bytes per pixel instead of 128). Lane (lr, kq) holds
        // channels 8 kq .. + 7 and 32 + 8 kq .. + 7 of pixel lr; the four kq lanes of a pixel are 16 lanes apart.
        typedef short s8_t __attribute__((ext_vector_type(8)));
        const s8_t s0 = __builtin_bit_cast(s8_t, o[0]), s1 = __builtin_bit_cast(s8_t, o[1]);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { lo |= (s0[j] > 0 ? 1u : 0u) << j; hi |= (s1[j] > 0 ? 1u : 0u) << j; }
        lo <<= 8 * kq; hi <<= 8 * kq;
        // OR over the four 16-lane rows: v_permlane16_swap / v_permlane32_swap (gfx950) pair the rows in registers, no LDS crossbar
        typedef unsigned u2v_t __attribute__((ext_vector_type(2)));
        u2v_t t = __builtin_amdgcn_permlane16_swap(lo, lo, false, false); lo = t[0] | t[1];
        t = __builtin_amdgcn_permlane16_swap(hi, hi, false, false); hi = t[0] | t[1];
        t = __builtin_amdgcn_permlane32_swap(lo, lo, false, false); lo = t[0] | t[1];
        t = __builtin_amdgcn_permlane32_swap(hi, hi, false, false); hi = t[0] | t[1];
        if (kq == 0) bits[(int64_t)g * 16 + lr] = (unsigned long long)lo | ((unsigned long long)hi << 32);
      }
      {
        // through a wave-private LDS slab so that one store instruction covers whole 128-byte pixel rows (8 pixels each; measured
        // against two 64-byte pieces per pixel and instruction: d1 17.6 -> 16.2 us, critic conv1 30.4 -> 30.0): pixel lr's 16-byte
        // chunk q at physical chunk q ^ (lr & 7); lane l reads chunk l & 7 of pixels l >> 3 and (l >> 3) + 8
        char* my = slab[threadIdx.x >> 6];
        __builtin_amdgcn_wave_barrier();
        *(h8_t*)(my + lr * 128 + ((kq ^ (lr & 7)) << 4)) = o[0];
        *(h8_t*)(my + lr * 128 + (((4 + kq) ^ (lr & 7)) << 4)) = o[1];
        __builtin_amdgcn_wave_barrier();
        const int p0 = lane >> 3, q = lane & 7;
        const h8_t v0 = *(const h8_t*)(my + p0 * 128 + ((q ^ (p0 & 7)) << 4));
        const h8_t v1 = *(const h8_t*)(my + (p0 + 8) * 128 + ((q ^ (p0 & 7)) << 4));
        char* d2 = out + ((((int64_t)g * 16 + p0) * ldout + coffout + mq * 64 + q * 8) << 1);
        *(h8_t*)d2 = v0;
        *(h8_t*)(d2 + (((int64_t)8 * ldout) << 1)) = v1;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = bn[j];
  }
}

// ---- fp16 transposed conv to one channel on the matrix cores -------------------------------------
// col[p][tap] = sum_c relu?(X[p][c]) * w[c][tap]: D[tap][pixel] = A[tap][c] * B[c][pixel] with
// v_mfma_f32_16x16x32_f16; B fragments are 16-byte row loads of X straight from global memory (each
// lane: 8 consecutive channels of its pixel), A fragments (the 16 x c weights) live in registers.
// Each lane ends with 4 consecutive taps of one pixel -> one 8-byte store; a wave writes 512 B.
// X2 != null: channels [c/2, c) are not read from X but produced on the fly as relu(fma(X2[p][ch - c/2], sc2, sh2)),
// the BatchNorm + ReLU of the raw decoder output X2 (dense, ld2): the same fp32 expression and fp16 rounding as the
// separate apply pass, which then never has to materialise that half of the concat buffer.
// use_fa: sc2 / sh2 do not exist yet - every workgroup derives them from the layer's exact accumulators (bn_acc.h) into LDS while
// its first rows are in flight, workgroup 0 publishes them with the backward's vectors and moves the running statistics (the
// bn_finalize_acc launch between the GEMM and this kernel, 5 us behind a dependent launch, is gone).
template <int KS>   // KS = c / 32
__global__ void __launch_bounds__(256) c1_col_kernel(const char* X, const float* __restrict__ w, half_t* col,
                                                     int64_t P, int ldx, int coffx, int relu_in, const char* X2, int ld2,
                                                     const float* __restrict__ sc2, const float* __restrict__ sh2, BnAccP fa, int use_fa) {
  constexpr int KH = KS / 2;
  __shared__ float aff[2][KH * 32];
  const int lane = threadIdx.x & 63;
  const int tapr = lane & 15, kq = lane >> 4;
  const int64_t ngroups = (P + 15) / 16;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
  const h8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
  // a wave walks 4+ groups: the next group's 16-byte row pieces are requested before this group's are used
  auto load_g = [&](int64_t g, h8_t (&raw)[KS]) {
    const int64_t pix = g * 16 + tapr;   // this lane's B column = pixel
    const bool live = pix < P;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (X2 && ks >= KH) raw[ks] = live ? *(const h8_t*)(X2 + ((pix * ld2 + (ks - KH) * 32 + kq * 8) << 1)) : zero;
      else raw[ks] = live ? *(const h8_t*)(X + ((pix * ldx + coffx + ks * 32 + kq * 8) << 1)) : zero;
    }
  };
  h8_t raw[KS], nxt[KS];
  int64_t g = wave;
  if (g < ngroups) load_g(g, raw);
  h8_t af[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) af[ks][j] = (half_t)w[(ks * 32 + kq * 8 + j) * 16 + tapr];
  float s2[KH][8], h2[KH][8];
  if (X2) {
    if (use_fa) {
      if (threadIdx.x < KH * 32) {
        float a, b;
        bn_from_acc(fa, KH * 32, threadIdx.x, 0, blockIdx.x == 0, a, b);
        aff[0][threadIdx.x] = a;
        aff[1][threadIdx.x] = b;
      }
      if (blockIdx.x == 0 && fa.zero_next) zero_words64(fa.zero_next, fa.zero_words);
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < KH; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s2[ks][j] = aff[0][ks * 32 + kq * 8 + j]; h2[ks][j] = aff[1][ks * 32 + kq * 8 + j]; }
    } else {
#pragma unroll
      for (int ks = 0; ks < KH; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s2[ks][j] = sc2[ks * 32 + kq * 8 + j]; h2[ks][j] = sh2[ks * 32 + kq * 8 + j]; }
    }
  }
  for (; g < ngroups; g += nwaves) {
    if (g + nwaves < ngroups) load_g(g + nwaves, nxt);
    const int64_t pix = g * 16 + tapr;
    const bool live = pix < P;
    h8_t bf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (X2 && ks >= KH) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = fmaf((float)raw[ks][j], s2[ks >= KH ? ks - KH : 0][j], h2[ks >= KH ? ks - KH : 0][j]);
          bf[ks][j] = live ? (half_t)(t > 0.f ? t : 0.f) : (half_t)0.f;
        }
      } else {
        bf[ks] = raw[ks];
        if (relu_in) bf[ks] = __builtin_elementwise_max(bf[ks], zero);
      }
    }
    f4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks], bf[ks], acc, 0, 0, 0);
    if (live) {
      h4_t o = {(half_t)acc[0], (half_t)acc[1], (half_t)acc[2], (half_t)acc[3]};
      *(h4_t*)(col + pix * 16 + kq * 4) = o;   // taps 4kq..4kq+3 of this pixel
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) raw[ks] = nxt[ks];
  }
}

// overlap-add of the 16 taps: out(2j+py, 2i+px) = post(bias + sum_{ty,tx} col[(j+py-ty, i+px-tx)][(1-py+2ty)*4 + 1-px+2tx])
__global__ void __launch_bounds__(256) c1_col2im_kernel(const half_t* __restrict__ col, const float* __restrict__ bias,
                                                        float* img, float* img2, int n, int Hs, int Ws, int post, float out_scale) {
  const int64_t total = (int64_t)n * Hs * Ws;
  const int H = 2 * Hs, W = 2 * Ws;
  const float b = bias ? bias[0] : 0.f;
  for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (int64_t)gridDim.x * 256) {
    const int i = (int)(pix % Ws);
    const int64_t t = pix / Ws;
    const int j = (int)(t % Hs);
    const int nn = (int)(t / Hs);
    float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int sy = j + dy, sx = i + dx;
        if (sy < 0 || sy >= Hs || sx < 0 || sx >= Ws) continue;
        const half_t* cp = col + (((int64_t)nn * Hs + sy) * Ws + sx) * 16;
        const h8_t c0 = *(const h8_t*)cp, c1 = *(const h8_t*)(cp + 8);
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int ty = py - dy, tx = px - dx;
            if (ty < 0 || ty > 1 || tx < 0 || tx > 1) continue;
            const int tap = (1 - py + 2 * ty) * 4 + (1 - px + 2 * tx);
            o[py * 2 + px] += tap < 8 ? (float)c0[tap & 7] : (float)c1[tap & 7];
          }
      }
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      float v0 = o[py * 2] + b, v1 = o[py * 2 + 1] + b;
      if (post == 1) { v0 = tanhf(v0); v1 = tanhf(v1); }
      const int64_t oo = ((int64_t)nn * H + 2 * j + py) * W + 2 * i;
      *(float2*)(img + oo) = make_float2(v0 * out_scale, v1 * out_scale);
      if (img2) *(float2*)(img2 + oo) = make_float2(v0 * out_scale, v1 * out_scale);   // the caller's copy of the saved output
    }
  }
}

// ---- the same two steps in ONE launch (round 4) ----------------------------------------------------------------------------
// A workgroup owns TH rows of one image of the small grid (all Ws columns, so only rows have a halo): it computes col[p][tap] for
// its TH + 2 rows on the MFMA exactly as c1_col_kernel does (same fragments, same fp16 rounding of col), keeps them in LDS
// ((TH + 2) x Ws x 32 bytes) and overlap-adds its own TH rows with c1_col2im_kernel's loop order - bit-identical output, the col
// tensor (16.8 MB written and re-read at the headline shape) and one launch gone; the price is (TH + 2) / TH of the input reads,
// the extra rows being the neighbour workgroup's (L2). u1 of the generator at 256x256, n = 32: 28.6 + 15.6 us -> see DESIGN.md.
template <int KS>   // KS = c / 32
__global__ void __launch_bounds__(256) c1_scatter_fused_kernel(const char* X, const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* img, float* img2, int Hs, int Ws, int TH, int ldx, int coffx, int relu_in,
                                                               const char* X2, int ld2, const float* __restrict__ sc2,
                                                               const float* __restrict__ sh2, BnAccP fa, int use_fa, int post, float out_scale) {
  constexpr int KH = KS / 2;
  extern __shared__ __attribute__((aligned(16))) char smem_u1[];
  half_t* col = (half_t*)smem_u1;                              // [(TH + 2) * Ws][16]
  float* aff = (float*)(smem_u1 + (int64_t)(TH + 2) * Ws * 32);  // [2][KH * 32]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tapr = lane & 15, kq = lane >> 4;
  const int bands = (Hs + TH - 1) / TH;
  const int nn = blockIdx.x / bands, y0 = (blockIdx.x % bands) * TH;
  const int gpr = Ws >> 4, G = (TH + 2) * gpr;                 // pixel groups of 16 along a row
  const h8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
  auto load_g = [&](int g, h8_t (&raw)[KS]) {
    const int ry = g / gpr, x = (g - ry * gpr) * 16 + tapr, y = y0 - 1 + ry;
    const bool live = g < G && y >= 0 && y < Hs;
    const int64_t pix = ((int64_t)nn * Hs + (live ? y : 0)) * Ws + x;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (X2 && ks >= KH) raw[ks] = live ? *(const h8_t*)(X2 + ((pix * ld2 + (ks - KH) * 32 + kq * 8) << 1)) : zero;
      else raw[ks] = live ? *(const h8_t*)(X + ((pix * ldx + coffx + ks * 32 + kq * 8) << 1)) : zero;
    }
  };
  h8_t raw[KS], nx1[KS], nx2[KS];
  int g = wave;
  load_g(g, raw);
  load_g(g + 4, nx1);
  h8_t af[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) af[ks][j] = (half_t)w[(ks * 32 + kq * 8 + j) * 16 + tapr];
  float s2[KH][8], h2[KH][8];
  if (X2) {
    if (use_fa) {
      if (threadIdx.x < KH * 32) {
        float a, b;
        bn_from_acc(fa, KH * 32, threadIdx.x, 0, blockIdx.x == 0, a, b);
        aff[threadIdx.x] = a;
        aff[KH * 32 + threadIdx.x] = b;
      }
      if (blockIdx.x == 0 && fa.zero_next) zero_words64(fa.zero_next, fa.zero_words);
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < KH; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s2[ks][j] = aff[ks * 32 + kq * 8 + j]; h2[ks][j] = aff[KH * 32 + ks * 32 + kq * 8 + j]; }
    } else {
#pragma unroll
      for (int ks = 0; ks < KH; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s2[ks][j] = sc2[ks * 32 + kq * 8 + j]; h2[ks][j] = sh2[ks * 32 + kq * 8 + j]; }
    }
  }
  for (; g < G; g += 4) {
    load_g(g + 8, nx2);
    const int ry = g / gpr, x = (g - ry * gpr) * 16 + tapr, y = y0 - 1 + ry;
    const bool live = y >= 0 && y < Hs;
    h8_t bf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (X2 && ks >= KH) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = fmaf((float)raw[ks][j], s2[ks >= KH ? ks - KH : 0][j], h2[ks >= KH ? ks - KH : 0][j]);
          bf[ks][j] = live ? (half_t)(t > 0.f ? t : 0.f) : (half_t)0.f;
        }
      } else {
        bf[ks] = raw[ks];
        if (relu_in) bf[ks] = __builtin_elementwise_max(bf[ks], zero);
      }
    }
    f4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks], bf[ks], acc, 0, 0, 0);
    const h4_t o = {(half_t)acc[0], (half_t)acc[1], (half_t)acc[2], (half_t)acc[3]};     // rows outside the image: zeros
    *(h4_t*)(col + ((int64_t)ry * Ws + x) * 16 + kq * 4) = o;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { raw[ks] = nx1[ks]; nx1[ks] = nx2[ks]; }
  }
  __syncthreads();
  // overlap-add (c1_col2im_kernel's order of additions): out(2j+py, 2i+px) = post(bias + sum col[(j+py-ty, i+px-tx)][tap])
  const int H = 2 * Hs, W = 2 * Ws;
  const float b = bias ? bias[0] : 0.f;
  for (int p = threadIdx.x; p < TH * Ws; p += 256) {
    const int jl = p / Ws, i = p - jl * Ws, j = y0 + jl;
    if (j >= Hs) break;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int sy = j + dy, sx = i + dx;
        if (sy < 0 || sy >= Hs || sx < 0 || sx >= Ws) continue;
        const half_t* cp = col + ((int64_t)(jl + 1 + dy) * Ws + sx) * 16;
        const h8_t c0 = *(const h8_t*)cp, c1 = *(const h8_t*)(cp + 8);
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int ty = py - dy, tx = px - dx;
            if (ty < 0 || ty > 1 || tx < 0 || tx > 1) continue;
            const int tap = (1 - py + 2 * ty) * 4 + (1 - px + 2 * tx);
            o[py * 2 + px] += tap < 8 ? (float)c0[tap & 7] : (float)c1[tap & 7];
          }
      }
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      float v0 = o[py * 2] + b, v1 = o[py * 2 + 1] + b;
      if (post == 1) { v0 = tanhf(v0); v1 = tanhf(v1); }
      const int64_t oo = ((int64_t)nn * H + 2 * j + py) * W + 2 * i;
      *(float2*)(img + oo) = make_float2(v0 * out_scale, v1 * out_scale);
      if (img2) *(float2*)(img2 + oo) = make_float2(v0 * out_scale, v1 * out_scale);
    }
  }
}

// ---- transposed conv to a FEW channels (the frozen face-parsing network's head: 4 classes) -------------------------------
// The same two steps with OC output channels: col[p][o*16 + tap] = sum_c relu?(X[p][c]) * w[c][tap][o] - a 16*OC-row GEMM, OC MFMA
// row tiles per 32-channel step on the same B fragments - and the overlap-add per channel. 2*P*c*16*OC FLOP instead of the
// 2*4*P*(4c)*64 of the general kernels on weights padded to 64 output channels (16x less for OC = 4).
template <int KS, int OC>   // KS = c / 32
__global__ void __launch_bounds__(256) c1_col_mc_kernel(const char* X, const float* __restrict__ w, half_t* col, int64_t P, int ldx,
                                                        int coffx, int relu_in) {
  const int lane = threadIdx.x & 63;
  const int tapr = lane & 15, kq = lane >> 4;
  h8_t af[OC][KS];
#pragma unroll
  for (int o = 0; o < OC; ++o)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) af[o][ks][j] = (half_t)w[((ks * 32 + kq * 8 + j) * 16 + tapr) * OC + o];
  const int64_t ngroups = (P + 15) / 16;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
  const h8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t g = wave; g < ngroups; g += nwaves) {
    const int64_t pix = g * 16 + tapr;
    const bool live = pix < P;
    h8_t bf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf[ks] = live ? *(const h8_t*)(X + ((pix * ldx + coffx + ks * 32 + kq * 8) << 1)) : zero;
      if (relu_in) bf[ks] = __builtin_elementwise_max(bf[ks], zero);
    }
#pragma unroll
    for (int o = 0; o < OC; ++o) {
      f4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[o][ks], bf[ks], acc, 0, 0, 0);
      if (live) {
        h4_t v = {(half_t)acc[0], (half_t)acc[1], (half_t)acc[2], (half_t)acc[3]};
        *(h4_t*)(col + pix * (16 * OC) + o * 16 + kq * 4) = v;   // taps 4kq..4kq+3 of channel o
      }
    }
  }
}

// overlap-add per channel into (n, OC, H, W) fp32, bias + tanh (post = 1)
template <int OC>
__global__ void __launch_bounds__(256) c1_col2im_mc_kernel(const half_t* __restrict__ col, const float* __restrict__ bias, float* img,
                                                           float* img2, int n, int Hs, int Ws, int post) {
  const int64_t total = (int64_t)n * Hs * Ws * OC;
  const int H = 2 * Hs, W = 2 * Ws;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int o = (int)(id % OC);            // (channel fastest: the OC lanes of a pixel read one 32*OC-byte col row)
    const int64_t pix = id / OC;
    const int i = (int)(pix % Ws);
    const int64_t t = pix / Ws;
    const int j = (int)(t % Hs);
    const int nn = (int)(t / Hs);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int sy = j + dy, sx = i + dx;
        if (sy < 0 || sy >= Hs || sx < 0 || sx >= Ws) continue;
        const half_t* cp = col + ((((int64_t)nn * Hs + sy) * Ws + sx) * OC + o) * 16;
        const h8_t c0 = *(const h8_t*)cp, c1 = *(const h8_t*)(cp + 8);
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int ty = py - dy, tx = px - dx;
            if (ty < 0 || ty > 1 || tx < 0 || tx > 1) continue;
            const int tap = (1 - py + 2 * ty) * 4 + (1 - px + 2 * tx);
            acc[py * 2 + px] += tap < 8 ? (float)c0[tap & 7] : (float)c1[tap & 7];
          }
      }
    const float b = bias ? bias[o] : 0.f;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      float v0 = acc[py * 2] + b, v1 = acc[py * 2 + 1] + b;
      if (post == 1) { v0 = tanhf(v0); v1 = tanhf(v1); }
      const int64_t oo = (((int64_t)nn * OC + o) * H + 2 * j + py) * W + 2 * i;
      *(float2*)(img + oo) = make_float2(v0, v1);
      if (img2) *(float2*)(img2 + oo) = make_float2(v0, v1);
    }
  }
}

// input gradient of that head: out[p][c] = sum_{o,tap} G[o][p's tap] * w[c][tap][o], G the (n, OC, H, W) fp32 gradient: the gather of
// c1_gather_mfma_kernel with K = 16*OC (OC = 4: two full 32-wide MFMA steps, k = o*16 + tap)
template <int MTC>   // c / 16; OC = 4
__global__ void __launch_bounds__(256) c1_gather_mc4_kernel(const float* __restrict__ img, const float* __restrict__ w, char* out, int n, int Hs,
                                                            int Ws, int ldout, int coffout) {
  constexpr int OC = 4;
  const int lane = threadIdx.x & 63;
  const int lr = lane & 15, kq = lane >> 4;
  // A[row lr of tile mt][k = 32 s + 8 kq + j] = w[ch][tap = 8 (kq & 1) + j][o = 2 s + (kq >> 1)], ch = 16 consecutive channels per lane (the mapping c1_gather_mfma_kernel used before its 64-byte store pieces)
  h8_t af[MTC][2];
#pragma unroll
  for (int mt = 0; mt < MTC; ++mt) {
    const int ch = (mt >> 2) * 64 + (lr >> 2) * 16 + (mt & 3) * 4 + (lr & 3);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) af[mt][s2][j] = (half_t)w[(ch * 16 + (kq & 1) * 8 + j) * OC + 2 * s2 + (kq >> 1)];
  }
  const int H = 2 * Hs, W = 2 * Ws;
  const int64_t ngroups = (int64_t)n * Hs * Ws / 16;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
  auto load_b = [&](int64_t g, float (&bv)[2][8]) {
    const int64_t p0 = g * 16;
    const int x = (int)(p0 % Ws) + lr;
    const int64_t rowi = p0 / Ws;
    const int y = (int)(rowi % Hs);
    const int nn = (int)(rowi / Hs);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const float* ip = img + ((int64_t)nn * OC + 2 * s2 + (kq >> 1)) * H * W;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int iy = 2 * y - 1 + 2 * (kq & 1) + (j >> 2), ix = 2 * x - 1 + (j & 3);
        bv[s2][j] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? ip[(int64_t)iy * W + ix] : 0.f;
      }
    }
  };
  float bv[2][8], bn[2][8];
  int64_t g = wave;
  if (g < ngroups) load_b(g, bv);
  for (; g < ngroups; g += nwaves) {
    if (g + nwaves < ngroups) load_b(g + nwaves, bn);     // the next group's 16 loads are in flight during this group's MFMAs and stores
    const int64_t p0 = g * 16;
    h8_t bf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) bf[s2][j] = (half_t)bv[s2][j];
    const int64_t pix = p0 + lr;
    char* dst = out + ((pix * ldout + coffout) << 1);
#pragma unroll
    for (int mq = 0; mq < MTC / 4; ++mq) {
      h8_t o[2];
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) {
        f4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mq * 4 + m4][0], bf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mq * 4 + m4][1], bf[1], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[m4 >> 1][(m4 & 1) * 4 + r] = (half_t)acc[r];
      }
      *(h8_t*)(dst + ((mq * 64 + kq * 16) << 1)) = o[0];
      *(h8_t*)(dst + ((mq * 64 + kq * 16 + 8) << 1)) = o[1];
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[s2][j] = bn[s2][j];
  }
}

// ---- scatter (transposed conv to one channel), one lane group per small-resolution pixel ------
// The 2x2 output quad (2j..2j+1, 2i..2i+1) reads the 3x3 neighbourhood of (j,i):
//   out(2j+py, 2i+px) = sum_{ty,tx} X[j+py-ty][i+px-tx] . w[:, (1-py+2ty)*4 + (1-px+2tx)]
template <typename T>
__global__ void __launch_bounds__(256) c1_scatter_kernel(const char* X, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* img, int n, int Hs,
                                                         int Ws, int c, int ldx, int coffx, int relu_in, int post,
                                                         float out_scale) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int LPP = c / EPC;  // lanes per pixel (power of two <= 64)
  const int cl = threadIdx.x % LPP;
  float wr[16][EPC];  // this lane's weights
#pragma unroll
  for (int j = 0; j < EPC; ++j)
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) wr[tt][j] = w[(cl * EPC + j) * 16 + tt];
  const float b = bias ? bias[0] : 0.f;
  const int gpb = 256 / LPP;
  const int64_t total = (int64_t)n * Hs * Ws;
  const int H = 2 * Hs, W = 2 * Ws;
  const int64_t niter = (total + (int64_t)gridDim.x * gpb - 1) / ((int64_t)gridDim.x * gpb);
  for (int64_t it = 0; it < niter; ++it) {
    const int64_t pix = (it * gridDim.x + blockIdx.x) * gpb + threadIdx.x / LPP;
    const bool live = pix < total;
    const int i = live ? (int)(pix % Ws) : 0;
    const int64_t t = live ? pix / Ws : 0;
    const int j = (int)(t % Hs);
    const int nn = (int)(t / Hs);
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int sy = j + dy, sx = i + dx;
          if (sy < 0 || sy >= Hs || sx < 0 || sx >= Ws) continue;
          const char* src = X + ((((int64_t)nn * Hs + sy) * Ws + sx) * ldx + coffx + cl * EPC) * (int64_t)sizeof(T);
          float xv[EPC];
          if constexpr (std::is_same<T, half_t>::value) {
            const h8_t h = *(const h8_t*)src;
#pragma unroll
            for (int e = 0; e < EPC; ++e) xv[e] = (float)h[e];
          } else {
            const f4_t f = *(const f4_t*)src;
#pragma unroll
            for (int e = 0; e < EPC; ++e) xv[e] = f[e];
          }
          if (relu_in) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) xv[e] = xv[e] > 0.f ? xv[e] : 0.f;
          }
#pragma unroll
          for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
              const int ty = py - dy, tx = px - dx;  // source = (j+py-ty, i+px-tx)
              if (ty < 0 || ty > 1 || tx < 0 || tx > 1) continue;
              const int tap = (1 - py + 2 * ty) * 4 + (1 - px + 2 * tx);
              float s = 0.f;
#pragma unroll
              for (int e = 0; e < EPC; ++e) s = fmaf(xv[e], wr[tap][e], s);
              o[py * 2 + px] += s;
            }
        }
    }
    for (int off = LPP >> 1; off > 0; off >>= 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] += __shfl_xor(o[q], off);
    }
    if (live && cl == 0) {
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        float v0 = o[py * 2] + b, v1 = o[py * 2 + 1] + b;
        if (post == 1) { v0 = tanhf(v0); v1 = tanhf(v1); }
        float2 r = make_float2(v0 * out_scale, v1 * out_scale);
        *(float2*)(img + ((int64_t)nn * H + 2 * j + py) * W + 2 * i) = r;
      }
    }
  }
}

// ---- weight gradient: dW[c][tap] += scale * sum_p relu?(X[p][c]) * img[n,2y-1+ky,2x-1+kx] ------
template <typename T>
__global__ void __launch_bounds__(256) c1_wgrad_kernel(const char* X, const float* __restrict__ img, float* dW, int n,
                                                       int Hs, int Ws, int c, int ldx, int coffx, int relu_in,
                                                       float scale, float img_scale, int pix_per_block, float* __restrict__ part) {
  // part != null: the block's sums go to part[block][c*16] and c1_wgrad_reduce_kernel adds the blocks in a fixed order (bit-reproducible,
  // and no queue at the atomic unit: 1024 blocks x 1024 float atomics on the same 64 lines took most of this kernel's 130 us)
  constexpr int EPC = 16 / (int)sizeof(T);
  extern __shared__ float sred[];  // [pl][c][4] per round
  const int LPP = c / EPC;
  const int cl = threadIdx.x % LPP, pl = threadIdx.x / LPP;
  const int gpb = 256 / LPP;
  const int64_t total = (int64_t)n * Hs * Ws;
  const int H = 2 * Hs, W = 2 * Ws;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
  const int64_t p1 = min(total, p0 + pix_per_block);
  float acc[16][EPC];
#pragma unroll
  for (int tt = 0; tt < 16; ++tt)
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[tt][e] = 0.f;
  for (int64_t pix = p0 + pl; pix < p1; pix += gpb) {
    const int x = (int)(pix % Ws);
    const int64_t t = pix / Ws;
    const int y = (int)(t % Hs);
    const int nn = (int)(t / Hs);
    const char* src = X + (pix * ldx + coffx + cl * EPC) * (int64_t)sizeof(T);
    float xv[EPC];
    if constexpr (std::is_same<T, half_t>::value) {
      const h8_t h = *(const h8_t*)src;
#pragma unroll
      for (int e = 0; e < EPC; ++e) xv[e] = (float)h[e];
    } else {
      const f4_t f = *(const f4_t*)src;
#pragma unroll
      for (int e = 0; e < EPC; ++e) xv[e] = f[e];
    }
    if (relu_in) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) xv[e] = xv[e] > 0.f ? xv[e] : 0.f;
    }
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int iy = 2 * y - 1 + ky, ix = 2 * x - 1 + kx;
        const float g = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? img[((int64_t)nn * H + iy) * W + ix] * img_scale : 0.f;
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[ky * 4 + kx][e] = fmaf(xv[e], g, acc[ky * 4 + kx][e]);
      }
  }
  // block reduction over pixel lanes, 4 taps per round: sred[pl][c][4]
#pragma unroll
  for (int rnd = 0; rnd < 4; ++rnd) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPC; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q) sred[(pl * c + cl * EPC + e) * 4 + q] = acc[rnd * 4 + q][e];
    __syncthreads();
    for (int idx = threadIdx.x; idx < c * 4; idx += 256) {
      float s = 0.f;
      for (int k = 0; k < gpb; ++k) s += sred[k * c * 4 + idx];
      const int ch = idx >> 2, q = idx & 3;
      if (part) part[(int64_t)blockIdx.x * (c * 16) + ch * 16 + rnd * 4 + q] = s * scale;
      else atomicAdd(dW + ch * 16 + rnd * 4 + q, s * scale);
    }
  }
}

// ---- fp16 weight gradient of the single-channel layers on the matrix cores ------------------------
// dW[c][tap] += scale * sum_p relu?(X[p][c]) * img[n,2y-1+ky,2x-1+kx]:  D[tap][c] = A[tap][p] * B[p][c],
// K = pixels in tiles of 32 that lie in one image row (Ws % 32 == 0). Every wave works alone on its
// own K tiles: A fragments (16 taps x 8 pixels per lane) are gathered from the fp32 image, the X tile
// (32 pixels x c, pixel-major = K-major) is staged in a wave-private LDS slab and read with the
// transposing ds_read_b64_tr_b16; no block barrier. One float atomic per (c, tap) per wave at the end.
template <int NT>   // NT = c / 16
__global__ void __launch_bounds__(256) c1_wgrad_mfma_kernel(const char* X, const float* __restrict__ img, float* dW,
                                                            int n, int Hs, int Ws, int ldx, int coffx, int relu_in,
                                                            float scale, float img_scale, const char* X2, int ld2,
                                                            const float* __restrict__ sc2, const float* __restrict__ sh2, float* __restrict__ part) {
  constexpr int C = NT * 16;
  constexpr int LROW = C * 2 + 32;                  // padded LDS row: conflict-free transposing reads
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* slab = smem + wave * (32 * LROW);
  const int tap = lane & 15, kq = lane >> 4;
  const int ky = tap >> 2, kx = tap & 3;
  const int H = 2 * Hs, W = 2 * Ws;
  const int64_t ntiles = (int64_t)n * Hs * Ws / 32;
  const int64_t gw = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
  constexpr int CPR = C * 2 / 16;                   // 16-byte chunks per X row
  constexpr int RPP = 64 / CPR;                     // rows per load pass
  constexpr int NP = 32 / RPP;
  const int xr = lane / CPR, xc = lane % CPR;
  f4_t acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = f4_t{0.f, 0.f, 0.f, 0.f};
  const h8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
  const int g16 = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;

  // X2 != null: the channel chunks of the upper half come from the raw decoder output through relu(fma(x, sc2, sh2))
  const bool upper = X2 != nullptr && xc * 8 >= C / 2;
  float s2[8], h2[8];
  if (upper) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { s2[j] = sc2[xc * 8 - C / 2 + j]; h2[j] = sh2[xc * 8 - C / 2 + j]; }
  }
  auto load_x = [&](int64_t t, u4_t (&xv)[NP]) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int64_t row = t * 32 + xr + RPP * i;
      xv[i] = upper ? *(const u4_t*)(X2 + ((row * ld2 + xc * 8 - C / 2) << 1)) : *(const u4_t*)(X + ((row * ldx + coffx) << 1) + xc * 16);
    }
  };
  auto load_a = [&](int64_t t64, float (&av)[8]) {   // 8 pixels x0 + 8*kq + j of the tile's image row, tap (ky,kx)
    // the tile index is wave-uniform and < 2^31 (host check): scalar 32-bit divisions; clamped addresses + selects, no branches
    const int t32 = __builtin_amdgcn_readfirstlane((int)t64);
    const int tpr = Ws >> 5;                      // tiles per small-grid row (Ws % 32 == 0)
    const int rowi = t32 / tpr;
    const int x0 = (t32 - rowi * tpr) * 32;
    const int nn = rowi / Hs, y = rowi - nn * Hs;
    const int iy = 2 * y - 1 + ky;
    const bool yok = iy >= 0 && iy < H;
    const float* ip = img + ((int64_t)nn * H + (yok ? iy : 0)) * W;
    const int ixb = 2 * (x0 + 8 * kq) - 1 + kx;   // -1 only for j = 0 at the row start, W only for j = 7 at the row end
    const float a0 = ip[ixb < 0 ? 0 : ixb];
    const float a7 = ip[ixb + 14 < W ? ixb + 14 : W - 1];
    av[0] = (yok && ixb >= 0) ? a0 : 0.f;
#pragma unroll
    for (int j = 1; j < 7; ++j) { const float a = ip[ixb + 2 * j]; av[j] = yok ? a : 0.f; }
    av[7] = (yok && ixb + 14 < W) ? a7 : 0.f;
  };
  u4_t xv[NP], xn[NP];
  float av[8], an[8];
  int64_t t = gw;
  if (t < ntiles) { load_x(t, xv); load_a(t, av); }
  for (; t < ntiles; t += nw) {
    const int64_t tn = t + nw;
    if (tn < ntiles) { load_x(tn, xn); load_a(tn, an); }   // next tile in flight during this one
    h8_t af;
#pragma unroll
    for (int j = 0; j < 8; ++j) af[j] = (half_t)(av[j] * img_scale);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      u4_t v = xv[i];
      if (upper) {
        const h8_t raw = __builtin_bit_cast(h8_t, v);
        h8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float tt = fmaf((float)raw[j], s2[j], h2[j]);
          o[j] = (half_t)(tt > 0.f ? tt : 0.f);
        }
        v = __builtin_bit_cast(u4_t, o);
      } else if (relu_in) {
        typedef short s8_t __attribute__((ext_vector_type(8)));
        s8_t hh = __builtin_bit_cast(s8_t, v);
        const s8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
        hh = __builtin_elementwise_max(hh, z);
        v = __builtin_bit_cast(u4_t, hh);
      }
      *(u4_t*)(slab + (xr + RPP * i) * LROW + xc * 16) = v;
    }
    // (wave-private slab: the LDS unit executes one wave's accesses in program order)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ch = nt * 16 + 4 * tp;
      const char* b_lo = slab + (8 * g16 + tq) * LROW + ch * 2;
      const char* b_hi = slab + (8 * g16 + 4 + tq) * LROW + ch * 2;
      fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)b_lo);
      fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)b_hi);
      h4_t l4 = __builtin_bit_cast(h4_t, lo), h4 = __builtin_bit_cast(h4_t, hi);
      const h8_t bf = h8_t{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[nt], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) xv[i] = xn[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) av[j] = an[j];
  }
  // block reduction (4 waves) through LDS, then one float atomic per (c, tap) per block.
  // D[row = tap][col = channel]: lane holds channel nt*16 + (lane&15), taps 4*(lane>>4) + r
  __syncthreads();
  float* red = (float*)smem;   // [4][C*16]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * (C * 16) + (nt * 16 + (lane & 15)) * 16 + 4 * (lane >> 4) + r] = acc[nt][r];
  __syncthreads();
  // part != null: this block's sums go to part[block][C*16] (plain stores) and c1_wgrad_reduce_kernel adds the blocks in a fixed
  // order (bit-reproducible); else one float atomic per (c, tap) and block
  for (int i = threadIdx.x; i < C * 16; i += 256) {
    const float v = (red[i] + red[C * 16 + i] + red[2 * C * 16 + i] + red[3 * C * 16 + i]) * scale;
    if (part) part[(int64_t)blockIdx.x * (C * 16) + i] = v;
    else atomicAdd(dW + i, v);
  }
}

// dW[i] += sum over blocks of part[block][i] in a fixed order: a workgroup owns 16 outputs, thread (i, g) adds the blocks of
// group g (a contiguous range, four independent chains) and the 16 group sums are added in ascending order by one thread
// per output. (One thread per output walking all 512 blocks was 41 us on 4 workgroups; this is latency-bound on 32 loads.)
__global__ void __launch_bounds__(256) c1_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, int count, int blocks) {
  __shared__ float red[16][17];
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + o;
  const int per = (blocks + 15) / 16;
  const int b0 = g * per, b1 = min(blocks, b0 + per);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < count) {
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] += part[(int64_t)(b + u) * count + i];
    }
    for (; b < b1; ++b) s[0] += part[(int64_t)b * count + i];
  }
  red[g][o] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (g == 0 && i < count) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][o];
    dW[i] += t;
  }
}

// ---- discriminator head ----------------------------------------------------------------------
// h[n,py,px] = sum_{ky,kx,c} a4[n,py+ky,px+kx,c] * w5[ky*4+kx][c]     (Conv2d(512,1,4,1,0), networks.py:352)
template <typename T>
__global__ void __launch_bounds__(256) head_conv_kernel(const char* a4, const float* __restrict__ w5, float* h, int Hh,
                                                        int Wh, int c) {
  constexpr int EPC = 16 / (int)sizeof(T);
  __shared__ float red[4];
  const int Ph = Hh - 3, Pw = Wh - 3;
  const int o = blockIdx.x;  // (n, py, px)
  const int px = o % Pw, py = (o / Pw) % Ph, nn = o / (Pw * Ph);
  const int cpt = c / EPC;  // chunks per tap
  float s = 0.f;
  for (int idx = threadIdx.x; idx < 16 * cpt; idx += 256) {
    const int tap = idx / cpt, cc = idx % cpt;
    const int iy = py + (tap >> 2), ix = px + (tap & 3);
    const char* src = a4 + ((((int64_t)nn * Hh + iy) * Wh + ix) * c + cc * EPC) * (int64_t)sizeof(T);
    const float* wr = w5 + tap * c + cc * EPC;
    if constexpr (std::is_same<T, half_t>::value) {
      const h8_t v = *(const h8_t*)src;
#pragma unroll
      for (int e = 0; e < EPC; ++e) s = fmaf((float)v[e], wr[e], s);
    } else {
      const f4_t v = *(const f4_t*)src;
#pragma unroll
      for (int e = 0; e < EPC; ++e) s = fmaf(v[e], wr[e], s);
    }
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) h[o] = red[0] + red[1] + red[2] + red[3];
}

// out[n] = [sigmoid](bl + sum_p wl[p]*h[n,p])     (Flatten + Linear(P,1) [+ Sigmoid], networks.py:353-357)
__global__ void __launch_bounds__(64) head_linear_kernel(const float* h, const float* wl, const float* bl, float* out,
                                                         int P, int sigmoid) {
  const int nn = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < P; i += 64) s = fmaf(h[(int64_t)nn * P + i], wl[i], s);
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (threadIdx.x == 0) {
    s += bl[0];
    out[nn] = sigmoid ? 1.f / (1.f + expf(-s)) : s;
  }
}

// dz[n] = dy[n] * (sigmoid ? out(1-out) : 1); dh[n,p] = dz[n]*wl[p]; dwl[p] += sum_n dz[n]*h[n,p]; dbl += sum dz
__global__ void __launch_bounds__(256) head_linear_bwd_kernel(const float* dy, const float* out, const float* h,
                                                              const float* wl, float* dh, float* dwl, float* dbl, int n,
                                                              int P, int sigmoid) {
  for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
    float gw = 0.f;
    for (int nn = 0; nn < n; ++nn) {
      const float o = out[nn];
      const float dz = dy[nn] * (sigmoid ? o * (1.f - o) : 1.f);
      dh[(int64_t)nn * P + p] = dz * wl[p];
      gw = fmaf(dz, h[(int64_t)nn * P + p], gw);
    }
    if (dwl) atomicAdd(dwl + p, gw);
  }
  if (dbl && blockIdx.x == 0 && threadIdx.x == 0) {
    float gb = 0.f;
    for (int nn = 0; nn < n; ++nn) {
      const float o = out[nn];
      gb += dy[nn] * (sigmoid ? o * (1.f - o) : 1.f);
    }
    atomicAdd(dbl, gb);
  }
}

// da4[n,y,x,c] = loss_scale * sum_{tap: (y-ky,x-kx) valid} dh[n,y-ky,x-kx] * w5[tap][c]
template <typename T>
__global__ void __launch_bounds__(256) head_dgrad_kernel(const float* dh, const float* __restrict__ w5, char* da4, int n,
                                                         int Hh, int Wh, int c, float loss_scale) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int Ph = Hh - 3, Pw = Wh - 3;
  const int cpt = c / EPC;
  const int64_t total = (int64_t)n * Hh * Wh * cpt;
  for (int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (int64_t)gridDim.x * 256) {
    const int cc = (int)(gid % cpt);
    const int64_t pix = gid / cpt;
    const int x = (int)(pix % Wh), y = (int)((pix / Wh) % Hh), nn = (int)(pix / ((int64_t)Wh * Hh));
    float s[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
      const int py = y - (tap >> 2), px = x - (tap & 3);
      if (py < 0 || py >= Ph || px < 0 || px >= Pw) continue;
      const float g = dh[((int64_t)nn * Ph + py) * Pw + px];
      const float* wr = w5 + tap * c + cc * EPC;
#pragma unroll
      for (int e = 0; e < EPC; ++e) s[e] = fmaf(g, wr[e], s[e]);
    }
    T* dst = (T*)(da4 + (pix * c + cc * EPC) * (int64_t)sizeof(T));
#pragma unroll
    for (int e = 0; e < EPC; ++e) dst[e] = (T)(s[e] * loss_scale);
  }
}

// dw5[tap][c] += sum_{n,p} dh[n,p] * a4[n,p+tap][c]   (grid.y = n, grid.z = output row py)
// part != null: the (image, row band) blocks store their sums there ([block][16 c]) and head_wgrad_sum_kernel adds them in block
// order - bit-reproducible (the fp32 critic of BASELINE config 2); part == null: float atomics into dw5
template <typename T>
__global__ void __launch_bounds__(256) head_wgrad_kernel(const float* dh, const char* a4, float* dw5, int Hh, int Wh,
                                                         int c, float* part) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int Ph = Hh - 3, Pw = Wh - 3;
  const int cpt = c / EPC;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 16 * cpt) return;
  const int tap = idx / cpt, cc = idx % cpt;
  // blockIdx.z covers a band of output rows: 1/band of the atomics (they contend on 16*c addresses)
  const int nn = blockIdx.y;
  const int band = (Ph + (int)gridDim.z - 1) / (int)gridDim.z;
  const int py0 = blockIdx.z * band, py1 = min(Ph, py0 + band);
  float s[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) s[e] = 0.f;
  for (int py = py0; py < py1; ++py)
  for (int px = 0; px < Pw; ++px) {
    const float g = dh[((int64_t)nn * Ph + py) * Pw + px];
    const char* src = a4 + ((((int64_t)nn * Hh + py + (tap >> 2)) * Wh + px + (tap & 3)) * c + cc * EPC) * (int64_t)sizeof(T);
    if constexpr (std::is_same<T, half_t>::value) {
      const h8_t v = *(const h8_t*)src;
#pragma unroll
      for (int e = 0; e < EPC; ++e) s[e] = fmaf(g, (float)v[e], s[e]);
    } else {
      const f4_t v = *(const f4_t*)src;
#pragma unroll
      for (int e = 0; e < EPC; ++e) s[e] = fmaf(g, v[e], s[e]);
    }
  }
  if (part) {
    float* o = part + ((int64_t)blockIdx.y * gridDim.z + blockIdx.z) * 16 * c + tap * c + cc * EPC;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o[e] = s[e];
    return;
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) atomicAdd(dw5 + tap * c + cc * EPC + e, s[e]);
}
__global__ void __launch_bounds__(256) head_wgrad_sum_kernel(const float* part, int nparts, float* dw5, int count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int k = 0; k < nparts; ++k) s += part[(int64_t)k * count + i];
  dw5[i] += s;
}

// ---- fp16 head at c = 512 (the PatchGAN of networks.py:331-363): one pixel row = 1 KiB = one wave load --------
// Forward, one workgroup per image: t[px][tap] = a4[px][:] . w5[tap][:] on the MFMA (16 px x 16 taps x 32 channels
// per instruction; w5 as fp16 fragments resident in registers, fp32 accumulation), so every activation is read once;
// then h[p] = sum_tap t[p + (ky,kx)][tap], the Linear(P,1) and the optional sigmoid in the same workgroup.
constexpr int HT_PITCH = 17;   // floats per pixel row of t in LDS (16 taps + 1: conflict-free column reads)
__global__ void __launch_bounds__(1024) head_fwd512_kernel(const char* __restrict__ a4, const float* __restrict__ w5,
                                                           const float* __restrict__ wl, const float* __restrict__ bl,
                                                           float* __restrict__ h, float* __restrict__ out,
                                                           float* __restrict__ out2, int Hh, int Wh, int sigmoid,
                                                           const float* __restrict__ sc4, const float* __restrict__ sh4,
                                                           int n_per_group, int gstride, float* __restrict__ tbuf, int slices,
                                                           BnAccP fa, int use_fa) {
  // tbuf != null (large feature maps, few images): `slices` workgroups per image each compute t for their share of
  // the pixel tiles into tbuf[image][pixel][16] and stop; head_h512_kernel finishes (h, Linear, sigmoid).
  // sc4 != null: a4 is the RAW conv4 output; LeakyReLU(fma(x, sc4, sh4)) (the layer's BatchNorm + activation, the
  // population of image nn at +gstride floats) is applied to the fragments as they are loaded, the same fp32
  // expression and fp16 rounding as the separate apply pass
  extern __shared__ float t_lds[];   // [Hh*Wh][HT_PITCH] | 16 partial sums | w5 as fp16 [16 taps][512] | scale, shift [2][512]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nn = tbuf ? blockIdx.x / slices : blockIdx.x, npx = Hh * Wh;
  float* red = t_lds + (tbuf ? 0 : ((npx * HT_PITCH + 3) & ~3));   // 16-byte aligned
  half_t* wh = (half_t*)(red + 16);
  float* aff = (float*)(wh + 16 * 512);
  // this wave's first pixel tile is requested before the prologue (weights to LDS, the scale / shift derivation with its accumulator
  // round trip): at the critic's map sizes a wave owns ONE tile, so that is all of its activation traffic
  const char* img = a4 + (int64_t)nn * npx * 1024;
  const int ntile = (npx + 15) / 16;
  const int tile0 = tbuf ? (int)((int64_t)ntile * (blockIdx.x % slices) / slices) : 0;
  const int tile1 = tbuf ? (int)((int64_t)ntile * (blockIdx.x % slices + 1) / slices) : ntile;
  h8_t av[16];
  constexpr int AV_EARLY = 12;   // chunks requested before the prologue; the rest behind it (all 16 + the prologue's 64 accumulator
                                 // registers went 16 bytes over the 128 VGPRs of a 1024-thread workgroup)
  const char* row0;
  {
    const int t0 = tile0 + wave < tile1 ? tile0 + wave : tile0;     // (a wave without a tile loads one it will not use)
    row0 = img + (int64_t)min(t0 * 16 + (lane & 15), npx - 1) * 1024 + (lane >> 4) * 16;
#pragma unroll
    for (int ks = 0; ks < AV_EARLY; ++ks) av[ks] = *(const h8_t*)(row0 + ks * 64);
  }
  if (sc4 && use_fa) {
    // the scale / shift vectors do not exist yet: derived here from conv4's exact accumulators (bn_acc.h) - every workgroup the
    // vectors of its image's population; workgroup 0 publishes ALL populations' vectors (sc4 / sh4 and the backward's), moves the
    // running statistics population by population and clears the layer's other accumulator region: no finalize launch
    const int grp = nn / n_per_group;
    if (threadIdx.x < 512) {
      float a = 0.f, b = 0.f;
      if (blockIdx.x == 0) {
        for (int j = 0; j < fa.groups; ++j) {
          float aj, bj;
          bn_from_acc(fa, 512, threadIdx.x, j, true, aj, bj);
          if (j == grp) { a = aj; b = bj; }
        }
      } else {
        bn_from_acc(fa, 512, threadIdx.x, grp, false, a, b);
      }
      aff[threadIdx.x] = a;
      aff[512 + threadIdx.x] = b;
    }
    if (blockIdx.x == 0 && fa.zero_next) {
      for (int i = threadIdx.x; i < fa.zero_words; i += 1024) fa.zero_next[i] = 0ull;
    }
  } else if (sc4) {
    const int go = (nn / n_per_group) * gstride;
    for (int i = threadIdx.x; i < 512; i += 1024) { aff[i] = sc4[go + i]; aff[512 + i] = sh4[go + i]; }
  }
  for (int i = threadIdx.x; i < 16 * 512 / 4; i += 1024) {
    const f4_t v = *(const f4_t*)(w5 + i * 4);
    *(h4_t*)(wh + i * 4) = h4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
  }
#pragma unroll
  for (int ks = AV_EARLY; ks < 16; ++ks) av[ks] = *(const h8_t*)(row0 + ks * 64);
  __syncthreads();
  // (the weight fragments are read from LDS next to their MFMA: kept in registers beside av - 64 + 64 of the 128 a 1024-thread
  //  workgroup has per lane - the kernel spilled 132 bytes per lane, and at the critic's map sizes a wave owns one tile anyway)
  const half_t* bfp = wh + (lane & 15) * 512 + (lane >> 4) * 8;
  for (int tile = tile0 + wave; tile < tile1; tile += 16) {
    if (tile != tile0 + wave) {      // later tiles of this wave (large maps)
      const int px = min(tile * 16 + (lane & 15), npx - 1);
      const char* row = img + (int64_t)px * 1024 + (lane >> 4) * 16;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) av[ks] = *(const h8_t*)(row + ks * 64);
    }
    if (sc4) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const float* a0 = aff + ks * 32 + (lane >> 4) * 8;
        const f4_t s0 = *(const f4_t*)a0, s1 = *(const f4_t*)(a0 + 4), h0 = *(const f4_t*)(a0 + 512), h1 = *(const f4_t*)(a0 + 516);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = fmaf((float)av[ks][j], j < 4 ? s0[j & 3] : s1[j & 3], j < 4 ? h0[j & 3] : h1[j & 3]);
          av[ks][j] = (half_t)(t > 0.f ? t : 0.2f * t);
        }
      }
    }
    f4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[ks], *(const h8_t*)(bfp + ks * 32), acc, 0, 0, 0);
    // D[row = pixel][col = tap]: lane holds tap (lane&15), pixels 4*(lane>>4) + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = tile * 16 + 4 * (lane >> 4) + r;
      if (q < npx) {
        if (tbuf) tbuf[((int64_t)nn * npx + q) * 16 + (lane & 15)] = acc[r];
        else t_lds[q * HT_PITCH + (lane & 15)] = acc[r];
      }
    }
  }
  if (tbuf) return;
  __syncthreads();
  const int Ph = Hh - 3, Pw = Wh - 3, P = Ph * Pw;
  float part = 0.f;
  for (int p = threadIdx.x; p < P; p += 1024) {
    const int py = p / Pw, px = p - py * Pw;
    float sacc = 0.f;
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) sacc += t_lds[((py + (tap >> 2)) * Wh + px + (tap & 3)) * HT_PITCH + tap];
    h[(int64_t)nn * P + p] = sacc;
    part = fmaf(sacc, wl[p], part);
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float z = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) z += red[w];
    z += bl[0];
    z = sigmoid ? 1.f / (1.f + expf(-z)) : z;
    out[nn] = z;
    if (out2) out2[nn] = z;
  }
}

// second stage of the sliced forward: h[p] = sum_tap t[p + (ky,kx)][tap], Linear(P,1), sigmoid; one workgroup per image
__global__ void __launch_bounds__(256) head_h512_kernel(const float* __restrict__ tbuf, const float* __restrict__ wl,
                                                        const float* __restrict__ bl, float* __restrict__ h,
                                                        float* __restrict__ out, float* __restrict__ out2, int Hh, int Wh,
                                                        int sigmoid) {
  __shared__ float red[4];
  const int nn = blockIdx.x, npx = Hh * Wh, Ph = Hh - 3, Pw = Wh - 3, P = Ph * Pw;
  const float* t = tbuf + (int64_t)nn * npx * 16;
  float part = 0.f;
  for (int p = threadIdx.x; p < P; p += 256) {
    const int py = p / Pw, px = p - py * Pw;
    float sacc = 0.f;
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) sacc += t[((py + (tap >> 2)) * Wh + px + (tap & 3)) * 16 + tap];
    h[(int64_t)nn * P + p] = sacc;
    part = fmaf(sacc, wl[p], part);
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float z = ((red[0] + red[1]) + (red[2] + red[3])) + bl[0];
    z = sigmoid ? 1.f / (1.f + expf(-z)) : z;
    out[nn] = z;
    if (out2) out2[nn] = z;
  }
}

// Backward, two roles by blockIdx in one launch (dh[n,p] = dz[n]*wl[p] is never materialised):
//   [0, nb)        input gradient: da4[n,y,x,:] = loss_scale * sum_tap dh[n,y-ky,x-kx] * w5[tap][:], a lane owns 8
//                  channels with its 16x8 weights in registers, a wave owns a pixel (one 1-KiB store);
//   [nb, 2nb)      weight-gradient partial of the same pixel range: acc[tap][8ch] += dh[..] * a4[n,y,x,8ch], reduced over
//                  the 4 waves in LDS and stored to part[block][16][512] (summed in fixed order by head_wsum512_kernel);
//   The Linear's gradients dwl[p] += sum_n dz[n]*h[n,p], dbl += sum_n dz[n] ride in head_wsum512_kernel.
// Without parameter gradients (frozen critic) only the first nb blocks are launched.
__global__ void __launch_bounds__(256) head_bwd512_kernel(const float* __restrict__ dy, const float* __restrict__ outv,
                                                          const float* __restrict__ hsave, const float* __restrict__ wl,
                                                          const float* __restrict__ w5, const char* __restrict__ a4,
                                                          char* __restrict__ da4, float* __restrict__ part,
                                                          float* __restrict__ dwl, float* __restrict__ dbl, int n, int Hh,
                                                          int Wh, int bands, int sigmoid, float loss_scale,
                                                          const float* __restrict__ sc4, const float* __restrict__ sh4,
                                                          int n_per_group, int gstride, HeadBwdFuse hf) {
  __shared__ float red[2 * 16 * 512];   // 64 KiB: two waves' accumulators at a time
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Ph = Hh - 3, Pw = Wh - 3, P = Ph * Pw, npx = Hh * Wh;
  const int nb = n * bands;
  int b = blockIdx.x;
  const bool wgrad = b >= nb;
  if (wgrad) b -= nb;
  const int nn = b / bands, band = b - nn * bands;
  const int ppb = (npx + bands - 1) / bands;               // pixels per block
  const int q0 = band * ppb, q1 = min(npx, q0 + ppb);
  const float o = outv[nn];
  const float dz = dy[nn] * (sigmoid ? o * (1.f - o) : 1.f) * (wgrad ? 1.f : loss_scale);
  // gp[(py+3)*GW + px+3] = dz * wl[py*Pw+px] with a 3-wide zero border: dh of this image without bounds checks
  const int GW = Pw + 6;
  float* gp = red;
  for (int i = threadIdx.x; i < (Ph + 6) * GW; i += 256) {
    const int py = i / GW - 3, px = i - (py + 3) * GW - 3;
    gp[i] = (py >= 0 && py < Ph && px >= 0 && px < Pw) ? dz * wl[py * Pw + px] : 0.f;
  }
  __syncthreads();
  if (!wgrad) {
    float w[16][8];
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
      const f4_t lo = *(const f4_t*)(w5 + tap * 512 + lane * 8), hi = *(const f4_t*)(w5 + tap * 512 + lane * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { w[tap][e] = lo[e]; w[tap][4 + e] = hi[e]; }
    }
    // hf.acc != null: the BatchNorm-backward sums of the layer in front (HeadBwdArgs::bwd_*) from the gradient this block produces -
    // the rounded value it stores, as the separate reduction would read it back - and the layer's raw output x, four pixels' rows
    // of x requested ahead per wave
    const bool bw = hf.acc != nullptr;
    float bsc[8], bsh[8], bmu[8], biv[8], bs[8], bsx[8];
    if (bw) {
      const int go = (nn / hf.n_per_group) * hf.stride + lane * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) { bsc[e] = hf.scale[go + e]; bsh[e] = hf.shift[go + e]; bmu[e] = hf.mean[go + e]; biv[e] = hf.inv[go + e]; bs[e] = bsx[e] = 0.f; }
    }
    const char* xrow = hf.x + (int64_t)nn * npx * 1024 + lane * 16;
    for (int qb = q0 + wave; qb < q1; qb += 16) {
      h8_t xs[4];
      if (bw) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[j] = *(const h8_t*)(xrow + (int64_t)min(qb + 4 * j, npx - 1) * 1024);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = qb + 4 * j;
        if (q < q1) {                              // wave-uniform
          const int y = q / Wh, x = q - y * Wh;
          const float* gq = gp + (y + 3) * GW + x + 3;
          float g[16];
#pragma unroll
          for (int tap = 0; tap < 16; ++tap) g[tap] = gq[-(tap >> 2) * GW - (tap & 3)];   // LDS broadcast reads
          float sacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tap = 0; tap < 16; ++tap)
#pragma unroll
            for (int e = 0; e < 8; ++e) sacc[e] = fmaf(g[tap], w[tap][e], sacc[e]);
          h8_t v;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (half_t)sacc[e];
          *(h8_t*)(da4 + ((int64_t)nn * npx + q) * 1024 + lane * 16) = v;
          if (bw) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float xf = (float)xs[j][e];
              const float dz = (float)v[e] * (fmaf(xf, bsc[e], bsh[e]) > 0.f ? 1.f : hf.slope);
              bs[e] += dz;
              bsx[e] = fmaf(dz, (xf - bmu[e]) * biv[e], bsx[e]);
            }
          }
        }
      }
    }
    if (bw) {     // the four waves' sums through LDS (gp is done with), then one exact add per channel and quantity
      __syncthreads();
      float* fold = red;   // [4 waves][512][2]
#pragma unroll
      for (int e = 0; e < 8; ++e) { fold[(wave * 512 + lane * 8 + e) * 2] = bs[e]; fold[(wave * 512 + lane * 8 + e) * 2 + 1] = bsx[e]; }
      __syncthreads();
      const int grp = nn / hf.n_per_group, rep = b & (hf.reps - 1);
      for (int ch = threadIdx.x; ch < 512; ch += 256) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) { s0 += fold[(wv * 512 + ch) * 2]; s1 += fold[(wv * 512 + ch) * 2 + 1]; }
        gi_stat_add(hf.acc, 512, rep, grp, 0, ch, s0);
        gi_stat_add(hf.acc, 512, rep, grp, 1, ch, s1);
      }
    }
    return;
  }
  float acc[16][8];
#pragma unroll
  for (int tap = 0; tap < 16; ++tap)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[tap][e] = 0.f;
  float s4[8], h4[8];   // sc4 != null: a4 is the raw conv4 output (see head_fwd512_kernel); a lane owns 8 channels
  if (sc4) {
    const int go = (nn / n_per_group) * gstride + lane * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { s4[e] = sc4[go + e]; h4[e] = sh4[go + e]; }
  }
  const char* arow = a4 + (int64_t)nn * npx * 1024 + lane * 16;
  for (int qb = q0 + wave; qb < q1; qb += 32) {   // 8 pixel rows in flight per wave
    h8_t v[8];
#pragma unroll
    for (int jx = 0; jx < 8; ++jx) v[jx] = *(const h8_t*)(arow + (int64_t)min(qb + 4 * jx, npx - 1) * 1024);
#pragma unroll
    for (int jx = 0; jx < 8; ++jx) {
      const int q = qb + 4 * jx;
      if (q >= q1) break;                          // wave-uniform
      const int y = q / Wh, x = q - y * Wh;
      const float* gq = gp + (y + 3) * GW + x + 3;
      float g[16];
#pragma unroll
      for (int tap = 0; tap < 16; ++tap) g[tap] = gq[-(tap >> 2) * GW - (tap & 3)];
      float vf[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) vf[e] = (float)v[jx][e];
      if (sc4) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = fmaf(vf[e], s4[e], h4[e]);
          vf[e] = (float)(half_t)(t > 0.f ? t : 0.2f * t);
        }
      }
#pragma unroll
      for (int tap = 0; tap < 16; ++tap)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[tap][e] = fmaf(g[tap], vf[e], acc[tap][e]);
    }
  }
  __syncthreads();   // every wave is done with gp before red is reused
  // waves 2,3 -> LDS, waves 0,1 add; wave 1 -> LDS, wave 0 adds and stores the block's partial
  auto put = [&](int slot) {
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
      float* d = red + slot * 8192 + tap * 512 + lane * 8;
      *(f4_t*)d = f4_t{acc[tap][0], acc[tap][1], acc[tap][2], acc[tap][3]};
      *(f4_t*)(d + 4) = f4_t{acc[tap][4], acc[tap][5], acc[tap][6], acc[tap][7]};
    }
  };
  auto take = [&](int slot) {
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
      const float* d = red + slot * 8192 + tap * 512 + lane * 8;
      const f4_t lo = *(const f4_t*)d, hi = *(const f4_t*)(d + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[tap][e] += lo[e]; acc[tap][4 + e] += hi[e]; }
    }
  };
  if (wave >= 2) put(wave - 2);
  __syncthreads();
  if (wave < 2) take(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  if (wave == 0) {
    take(0);
    float* dst = part + (int64_t)b * 8192;
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
      *(f4_t*)(dst + tap * 512 + lane * 8) = f4_t{acc[tap][0], acc[tap][1], acc[tap][2], acc[tap][3]};
      *(f4_t*)(dst + tap * 512 + lane * 8 + 4) = f4_t{acc[tap][4], acc[tap][5], acc[tap][6], acc[tap][7]};
    }
  }
}

// dw5[i] += sum_b part[b][i] in a fixed order (deterministic): a workgroup owns 32 elements, 8 thread groups
// each sum every 8th partial, LDS combines the groups. Workgroups 256 .. 256+P-1 compute dwl[p] (threads over the
// batch, tree reduction), workgroup 256+P the Linear's bias gradient.
__global__ void __launch_bounds__(256) head_wsum512_kernel(const float* __restrict__ part, int nb, float* __restrict__ dw5,
                                                           const float* __restrict__ dy, const float* __restrict__ outv,
                                                           const float* __restrict__ hsave, float* __restrict__ dwl,
                                                           float* __restrict__ dbl, int n, int P, int sigmoid) {
  __shared__ float red[8][32];
  if (blockIdx.x >= 256) {
    const int p = blockIdx.x - 256;
    if (p == P && !dbl) return;
    float g = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
      const float o = outv[i];
      const float dz = dy[i] * (sigmoid ? o * (1.f - o) : 1.f);
      g += p < P ? dz * hsave[(int64_t)i * P + p] : dz;
    }
    for (int off = 32; off > 0; off >>= 1) g += __shfl_xor(g, off);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = g;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float t = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
      if (p < P) dwl[p] += t; else dbl[0] += t;
    }
    return;
  }
  const int e = threadIdx.x & 31, gq = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e;   // < 8192
  float s0 = 0.f, s1 = 0.f;
  int b = gq;
  for (; b + 8 < nb; b += 16) {
    s0 += part[(int64_t)b * 8192 + i];
    s1 += part[(int64_t)(b + 8) * 8192 + i];
  }
  if (b < nb) s0 += part[(int64_t)b * 8192 + i];
  red[gq][e] = s0 + s1;
  __syncthreads();
  if (gq == 0) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][e];
    dw5[i] += t;
  }
}

// row bands per image for the backward: ~256 workgroups, at least 32 pixels each
static bool head_fast() { return gi_opt(GI_OPT_HEAD_FAST) != 0; }   // GI_HEAD_FAST=0: the generic kernels
static int head_bands(int n, int npx) {
  int bands = 1;
  while (n * bands < 256 && npx / (bands * 2) >= 32) bands *= 2;
  return bands;
}

int grid_for(int64_t work_items, int per_block, int cap) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

bool op_head_affine_ok(int dtype, int c) { return dtype == GI_F16 && c == 512 && head_fast(); }

int64_t op_head_scratch_bytes(int max_n, int Hh, int Wh) {
  (void)Hh; (void)Wh;   // n * head_bands(n, .) < 512 while bands > 1, = n otherwise
  return (int64_t)(max_n > 512 ? max_n : 512) * 8192 * 4;
}

int op_c1_gather(hipStream_t st, int dtype, const float* img, const float* w, void* out, int n, int Hs, int Ws, int c,
                 int ldout, int coffout, int act_out, float in_scale, const float* bias, unsigned long long* bits, int* bits_written) {
  GI_REQUIRE(c % 8 == 0 && c <= 1024, "c1_gather: c=%d", c);
  if (bits_written) *bits_written = 0;
  const int64_t total = (int64_t)n * Hs * Ws * (c / 8);
  const int groups = c / 8;
  if (dtype == GI_F16 && !bias && (c == 64 || c == 128) && Ws % 16 == 0 && ldout % 8 == 0 && coffout % 8 == 0) {
    const int64_t ngroups = (int64_t)n * Hs * Ws / 16;
    GI_REQUIRE(ngroups < (1ll << 31) && (act_out == GI_ACT_NONE || act_out == GI_ACT_RELU || act_out == GI_ACT_LRELU), "c1_gather: %lld pixel groups / activation %d",
               (long long)ngroups, act_out);
    const int grid = grid_for(ngroups, 4, 256 * 8);
    unsigned long long* const bw = c == 64 ? bits : nullptr;   // (the sign words cover 64 channels)
    if (bw && bits_written) *bits_written = 1;
#define GI_C1G(MTC_, ACT_) hipLaunchKernelGGL((c1_gather_mfma_kernel<MTC_, ACT_>), dim3(grid), dim3(256), 0, st, img, w, (char*)out, n, Hs, Ws, ldout, coffout, in_scale, bw)
    if (c == 64) {
      if (act_out == GI_ACT_LRELU) GI_C1G(4, GI_ACT_LRELU); else if (act_out == GI_ACT_RELU) GI_C1G(4, GI_ACT_RELU); else GI_C1G(4, GI_ACT_NONE);
    } else {
      if (act_out == GI_ACT_LRELU) GI_C1G(8, GI_ACT_LRELU); else if (act_out == GI_ACT_RELU) GI_C1G(8, GI_ACT_RELU); else GI_C1G(8, GI_ACT_NONE);
    }
#undef GI_C1G
    GI_LAUNCH_CHECK();
    return GI_OK;
  }
  if (gi_is_pow2(groups) && groups <= 32 && Ws % (256 / groups) == 0) {
    const int S = 256 / groups;
    const int64_t nstrips = (int64_t)n * Hs * (Ws / S);
    const int grid = (int)(nstrips < 256 * 8 ? nstrips : 256 * 8);
    const size_t lds = (size_t)(16 * c + 2 * 4 * (2 * S + 2)) * 4;
    if (dtype == GI_F16)
      hipLaunchKernelGGL(c1_gather_strip_kernel<half_t>, dim3(grid), dim3(256), lds, st, img, w, (char*)out, n, Hs, Ws, c, ldout,
                         coffout, act_out, in_scale, bias);
    else
      hipLaunchKernelGGL(c1_gather_strip_kernel<float>, dim3(grid), dim3(256), lds, st, img, w, (char*)out, n, Hs, Ws, c, ldout,
                         coffout, act_out, in_scale, bias);
    GI_LAUNCH_CHECK();
    return GI_OK;
  }
  const int grid = grid_for(total, 256, 256 * 16);
  if (dtype == GI_F16)
    hipLaunchKernelGGL(c1_gather_kernel<half_t>, dim3(grid), dim3(256), c * 16 * 4, st, img, w, (char*)out, n, Hs, Ws, c,
                       ldout, coffout, act_out, in_scale, bias);
  else
    hipLaunchKernelGGL(c1_gather_kernel<float>, dim3(grid), dim3(256), c * 16 * 4, st, img, w, (char*)out, n, Hs, Ws, c,
                       ldout, coffout, act_out, in_scale, bias);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

// first stage for many rows (the fused form writes one row per GEMM workgroup: 2048 on the critic): workgroup g adds rows
// [g * per, (g + 1) * per) in order for all `count` outputs, 16-byte loads, eight in flight, and writes row g of `out`
__global__ void __launch_bounds__(256) c1_wgrad_rows_kernel(const float* __restrict__ part, float* __restrict__ out, int count, int blocks, int per) {
  const int b0 = blockIdx.x * per, b1 = min(blocks, b0 + per);
  const int stride = count / 4;
  for (int i = threadIdx.x; i < stride; i += 256) {
    f4_t s = {0.f, 0.f, 0.f, 0.f};
    s = gi_ordered_sum_f4((const f4_t*)part + i, stride, b0, b1, s);
    ((f4_t*)out)[(int64_t)blockIdx.x * stride + i] = s;
  }
}

int op_c1_wgrad_reduce(hipStream_t st, const float* part, float* dW, int count, int blocks, float* scratch, int64_t scratch_floats) {
  constexpr int G1 = 64;
  if (blocks >= 4 * G1 && count % 4 == 0 && scratch && scratch_floats >= (int64_t)G1 * count) {
    const int per = (blocks + G1 - 1) / G1;
    hipLaunchKernelGGL(c1_wgrad_rows_kernel, dim3(G1), dim3(256), 0, st, part, scratch, count, blocks, per);
    GI_LAUNCH_CHECK();
    part = scratch;
    blocks = (blocks + per - 1) / per;
  }
  hipLaunchKernelGGL(c1_wgrad_reduce_kernel, dim3((count + 15) / 16), dim3(256), 0, st, part, dW, count, blocks);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

bool op_c1_affine_ok(int dtype, int c, int Ws, int ldx, int coffx) {
  return dtype == GI_F16 && (c == 64 || c == 128) && Ws % 32 == 0 && ldx % 8 == 0 && coffx % 8 == 0;
}

int op_c1_scatter(hipStream_t st, int dtype, const void* X, const float* w, const float* bias, float* img, int n, int Hs,
                  int Ws, int c, int ldx, int coffx, int relu_in, int post, float out_scale, void* col_scratch, float* img2,
                  const C1Affine* aff) {
  GI_REQUIRE(!aff || (col_scratch && op_c1_affine_ok(dtype, c, Ws, ldx, coffx) && aff->ld2 % 8 == 0), "c1_scatter: fused upper half unsupported here");
  if (dtype == GI_F16 && col_scratch && (c == 64 || c == 128) && ldx % 8 == 0 && coffx % 8 == 0) {
    const int64_t P = (int64_t)n * Hs * Ws;
    const int grid = grid_for((P + 15) / 16, 4, 256 * 8);
    const char* x2 = aff ? (const char*)aff->x2 : nullptr;
    const int ld2 = aff ? aff->ld2 : 0;
    const float* sc2 = aff ? aff->scale : nullptr;
    const float* sh2 = aff ? aff->shift : nullptr;
    BnAccP fa = {};
    const int use_fa = (aff && aff->bn) ? 1 : 0;
    if (use_fa) {
      GI_REQUIRE(aff->bn->groups == 1, "c1_scatter: one BatchNorm population expected");
      gi_fill_acc_params(fa, *aff->bn);
    }
    // one launch (c1_scatter_fused_kernel) where a band of rows fits LDS; GI_C1_FUSED=0: the col tensor + the overlap-add launch
    int TH = 1024 / Ws;
    if (TH < 1) TH = 1;
    if (TH > Hs) TH = Hs;
    const size_t lds_f = (size_t)(TH + 2) * Ws * 32 + 2 * (c / 2) * sizeof(float);
    if (gi_opt(GI_OPT_C1_FUSED) && Ws % 16 == 0 && lds_f <= 64 * 1024 && (int64_t)n * ((Hs + TH - 1) / TH) < (1ll << 31)) {
      const int gridf = n * ((Hs + TH - 1) / TH);
      if (c == 128)
        hipLaunchKernelGGL(c1_scatter_fused_kernel<4>, dim3(gridf), dim3(256), lds_f, st, (const char*)X, w, bias, img, img2, Hs, Ws, TH, ldx, coffx, relu_in,
                           x2, ld2, sc2, sh2, fa, use_fa, post, out_scale);
      else
        hipLaunchKernelGGL(c1_scatter_fused_kernel<2>, dim3(gridf), dim3(256), lds_f, st, (const char*)X, w, bias, img, img2, Hs, Ws, TH, ldx, coffx, relu_in,
                           x2, ld2, sc2, sh2, fa, use_fa, post, out_scale);
      GI_LAUNCH_CHECK();
      return GI_OK;
    }
    if (c == 128)
      hipLaunchKernelGGL(c1_col_kernel<4>, dim3(grid), dim3(256), 0, st, (const char*)X, w, (half_t*)col_scratch, P, ldx, coffx, relu_in, x2, ld2, sc2, sh2, fa, use_fa);
    else
      hipLaunchKernelGGL(c1_col_kernel<2>, dim3(grid), dim3(256), 0, st, (const char*)X, w, (half_t*)col_scratch, P, ldx, coffx, relu_in, x2, ld2, sc2, sh2, fa, use_fa);
    GI_LAUNCH_CHECK();
    hipLaunchKernelGGL(c1_col2im_kernel, dim3(grid_for(P, 256, 256 * 8)), dim3(256), 0, st, (const half_t*)col_scratch, bias, img, img2, n, Hs,
                       Ws, post, out_scale);
    GI_LAUNCH_CHECK();
    return GI_OK;
  }
  const int epc = dtype == GI_F16 ? 8 : 4;
  const int lpp = c / epc;
  GI_REQUIRE(c % epc == 0 && gi_is_pow2(lpp) && lpp <= 64, "c1_scatter: c=%d unsupported", c);
  const int64_t total = (int64_t)n * Hs * Ws;
  const int grid = grid_for(total, 256 / lpp, 256 * 8);
  if (dtype == GI_F16)
    hipLaunchKernelGGL(c1_scatter_kernel<half_t>, dim3(grid), dim3(256), 0, st, (const char*)X, w, bias, img, n, Hs, Ws,
                       c, ldx, coffx, relu_in, post, out_scale);
  else
    hipLaunchKernelGGL(c1_scatter_kernel<float>, dim3(grid), dim3(256), 0, st, (const char*)X, w, bias, img, n, Hs, Ws, c,
                       ldx, coffx, relu_in, post, out_scale);
  GI_LAUNCH_CHECK();
  if (img2) GI_HIP(hipMemcpyAsync(img2, img, (size_t)n * 4 * Hs * Ws * 4, hipMemcpyDeviceToDevice, st));
  return GI_OK;
}

int op_c1_wgrad(hipStream_t st, int dtype, const void* X, const float* img, float* dW, int n, int Hs, int Ws, int c,
                int ldx, int coffx, int relu_in, float scale, float img_scale, const C1Affine* aff, float* scratch, int64_t scratch_floats) {
  GI_REQUIRE(!aff || (op_c1_affine_ok(dtype, c, Ws, ldx, coffx) && aff->ld2 % 8 == 0), "c1_wgrad: fused upper half unsupported here");
  const char* x2 = aff ? (const char*)aff->x2 : nullptr;
  const int ld2 = aff ? aff->ld2 : 0;
  const float* sc2 = aff ? aff->scale : nullptr;
  const float* sh2 = aff ? aff->shift : nullptr;
  if (dtype == GI_F16 && (c == 64 || c == 128) && Ws % 32 == 0 && ldx % 8 == 0 && coffx % 8 == 0) {
    const int64_t ntiles = (int64_t)n * Hs * Ws / 32;
    GI_REQUIRE(ntiles < (1ll << 31), "c1_wgrad: %lld pixel tiles", (long long)ntiles);
    int grid = (int)((ntiles + 3) / 4);
    const int cap = gi_tune("GI_C1W_GRID", 512);       // per workgroup: c*16 partial sums (stores or atomics)
    if (grid > cap) grid = cap;
    size_t lds = (size_t)4 * 32 * (c * 2 + 32);
    if (lds < (size_t)4 * c * 16 * 4) lds = (size_t)4 * c * 16 * 4;
    float* part = (scratch && scratch_floats >= (int64_t)grid * c * 16) ? scratch : nullptr;   // deterministic two-stage sum
    if (c == 64)
      hipLaunchKernelGGL(c1_wgrad_mfma_kernel<4>, dim3(grid), dim3(256), lds, st, (const char*)X, img, dW, n, Hs, Ws, ldx, coffx, relu_in,
                         scale, img_scale, x2, ld2, sc2, sh2, part);
    else
      hipLaunchKernelGGL(c1_wgrad_mfma_kernel<8>, dim3(grid), dim3(256), lds, st, (const char*)X, img, dW, n, Hs, Ws, ldx, coffx, relu_in,
                         scale, img_scale, x2, ld2, sc2, sh2, part);
    GI_LAUNCH_CHECK();
    if (part) {
      hipLaunchKernelGGL(c1_wgrad_reduce_kernel, dim3((c * 16 + 15) / 16), dim3(256), 0, st, (const float*)part, dW, c * 16, grid);
      GI_LAUNCH_CHECK();
    }
    gi_note_kernel(part ? "c1_wgrad_mfma" : "c1_wgrad_mfma,atomics");   // (which summation ran: fixed-order partials or float atomics)
    return GI_OK;
  }
  const int epc = dtype == GI_F16 ? 8 : 4;
  const int lpp = c / epc;
  GI_REQUIRE(c % epc == 0 && gi_is_pow2(lpp) && lpp <= 64, "c1_wgrad: c=%d unsupported", c);
  const int64_t total = (int64_t)n * Hs * Ws;
  const int gpb = 256 / lpp;
  int ppb = (int)((total + 1023) / 1024);
  ppb = (ppb + gpb - 1) / gpb * gpb;
  if (ppb < gpb) ppb = gpb;
  const int grid = (int)((total + ppb - 1) / ppb);
  const size_t lds = (size_t)gpb * c * 4 * 4;
  float* part = (scratch && scratch_floats >= (int64_t)grid * c * 16) ? scratch : nullptr;   // deterministic two-stage sum
  if (dtype == GI_F16)
    hipLaunchKernelGGL(c1_wgrad_kernel<half_t>, dim3(grid), dim3(256), lds, st, (const char*)X, img, dW, n, Hs, Ws, c,
                       ldx, coffx, relu_in, scale, img_scale, ppb, part);
  else
    hipLaunchKernelGGL(c1_wgrad_kernel<float>, dim3(grid), dim3(256), lds, st, (const char*)X, img, dW, n, Hs, Ws, c, ldx,
                       coffx, relu_in, scale, img_scale, ppb, part);
  GI_LAUNCH_CHECK();
  if (part) {
    hipLaunchKernelGGL(c1_wgrad_reduce_kernel, dim3((c * 16 + 15) / 16), dim3(256), 0, st, (const float*)part, dW, c * 16, grid);
    GI_LAUNCH_CHECK();
  }
  gi_note_kernel(part ? "c1_wgrad" : "c1_wgrad,atomics");
  return GI_OK;
}

int op_head_forward(hipStream_t st, int dtype, const HeadArgs& a) {
  const int Ph = a.Hh - 3, Pw = a.Wh - 3;
  GI_REQUIRE(Ph >= 1 && Pw >= 1, "head: feature map %dx%d too small", a.Hh, a.Wh);
  const int blocks = a.n * Ph * Pw;
  GI_REQUIRE(!a.scale4 || op_head_affine_ok(dtype, a.c), "head: fused BatchNorm input needs the fp16 c=512 kernels");
  GI_REQUIRE(!a.bn || (a.scale4 && dtype == GI_F16 && a.c == 512 && head_fast()), "head: accumulator-derived BatchNorm needs the fused fp16 c=512 path");
  if (dtype == GI_F16 && a.c == 512 && head_fast()) {
    BnAccP fa = {};
    const int use_fa = a.bn ? 1 : 0;
    if (use_fa) {
      gi_fill_acc_params(fa, *a.bn);
      GI_REQUIRE(fa.out_stride == a.gstride, "head: population stride %d != %d", fa.out_stride, a.gstride);
    }
    const int lds = (((a.Hh * a.Wh * HT_PITCH + 3) & ~3) + 16) * 4 + 16 * 512 * 2 + 2 * 512 * 4;
    GI_REQUIRE(lds <= 160 * 1024, "head: feature map %dx%d too large", a.Hh, a.Wh);
    if (lds > 64 * 1024) {
      static GiDevOnce attr;
      if (attr.first()) { GI_HIP(hipFuncSetAttribute((const void*)head_fwd512_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
    }
    // large maps with few images (512x512 inputs: 32x32 map, 64 tiles per image): slice each image over several
    // workgroups so that ~256 of them stream the activations, and finish in a second small kernel
    const int npx = a.Hh * a.Wh;
    int slices = 1;
    while (a.n * slices < 256 && npx / (slices * 2) >= 256) slices *= 2;
    if (slices > 1 && a.tbuf && a.tbuf_bytes >= (int64_t)a.n * npx * 16 * 4) {
      const int lds2 = (16) * 4 + 16 * 512 * 2 + 2 * 512 * 4;
      hipLaunchKernelGGL(head_fwd512_kernel, dim3(a.n * slices), dim3(1024), lds2, st, (const char*)a.a4, a.w5, a.wl, a.bl, a.h, a.out, a.out2, a.Hh,
                         a.Wh, a.sigmoid, a.scale4, a.shift4, a.n_per_group > 0 ? a.n_per_group : a.n, a.gstride, a.tbuf, slices, fa, use_fa);
      GI_LAUNCH_CHECK();
      hipLaunchKernelGGL(head_h512_kernel, dim3(a.n), dim3(256), 0, st, a.tbuf, a.wl, a.bl, a.h, a.out, a.out2, a.Hh, a.Wh, a.sigmoid);
      GI_LAUNCH_CHECK();
      return GI_OK;
    }
    hipLaunchKernelGGL(head_fwd512_kernel, dim3(a.n), dim3(1024), lds, st, (const char*)a.a4, a.w5, a.wl, a.bl, a.h, a.out, a.out2, a.Hh, a.Wh, a.sigmoid,
                       a.scale4, a.shift4, a.n_per_group > 0 ? a.n_per_group : a.n, a.gstride, (float*)nullptr, 1, fa, use_fa);
    GI_LAUNCH_CHECK();
    return GI_OK;
  }
  if (dtype == GI_F16)
    hipLaunchKernelGGL(head_conv_kernel<half_t>, dim3(blocks), dim3(256), 0, st, (const char*)a.a4, a.w5, a.h, a.Hh, a.Wh, a.c);
  else
    hipLaunchKernelGGL(head_conv_kernel<float>, dim3(blocks), dim3(256), 0, st, (const char*)a.a4, a.w5, a.h, a.Hh, a.Wh, a.c);
  GI_LAUNCH_CHECK();
  hipLaunchKernelGGL(head_linear_kernel, dim3(a.n), dim3(64), 0, st, a.h, a.wl, a.bl, a.out, Ph * Pw, a.sigmoid);
  GI_LAUNCH_CHECK();
  if (a.out2) GI_HIP(hipMemcpyAsync(a.out2, a.out, (size_t)a.n * 4, hipMemcpyDeviceToDevice, st));
  return GI_OK;
}

int op_head_backward(hipStream_t st, int dtype, const HeadBwdArgs& a) {
  const int Ph = a.Hh - 3, Pw = a.Wh - 3, P = Ph * Pw;
  GI_REQUIRE(!a.scale4 || (op_head_affine_ok(dtype, a.c) && (!a.dw5 || (a.scratch && a.scratch_bytes >= (int64_t)a.n * head_bands(a.n, a.Hh * a.Wh) * 8192 * 4))),
             "head: fused BatchNorm input needs the fp16 c=512 kernels");
  if (dtype == GI_F16 && a.c == 512 && head_fast()) {
    const int bands = head_bands(a.n, a.Hh * a.Wh), nb = a.n * bands;
    const bool wg = a.dw5 != nullptr;
    HeadBwdFuse hf;
    hf.x = (const char*)a.bwd_x; hf.scale = a.bwd_scale; hf.shift = a.bwd_shift; hf.mean = a.bwd_mean; hf.inv = a.bwd_inv;
    hf.stride = a.bwd_stride; hf.n_per_group = a.bwd_n_per_group > 0 ? a.bwd_n_per_group : a.n; hf.reps = a.bwd_reps > 0 ? a.bwd_reps : 1;
    hf.slope = a.bwd_slope; hf.acc = (a.bwd_acc && a.bwd_x) ? a.bwd_acc : nullptr;
    if (!wg || (a.scratch && a.scratch_bytes >= (int64_t)nb * 8192 * 4)) {
      GI_REQUIRE(!wg || a.dwl, "head: dw5 without dwl");
      hipLaunchKernelGGL(head_bwd512_kernel, dim3(wg ? 2 * nb : nb), dim3(256), 0, st, a.dy, a.out, a.h, a.wl, a.w5, (const char*)a.a4,
                         (char*)a.da4, a.scratch, a.dwl, a.dbl, a.n, a.Hh, a.Wh, bands, a.sigmoid, a.loss_scale,
                         a.scale4, a.shift4, a.n_per_group > 0 ? a.n_per_group : a.n, a.gstride, hf);
      GI_LAUNCH_CHECK();
      if (hf.acc && a.bwd_applied) *a.bwd_applied = 1;
      if (wg) {
        hipLaunchKernelGGL(head_wsum512_kernel, dim3(256 + P + 1), dim3(256), 0, st, a.scratch, nb, a.dw5, a.dy, a.out, a.h, a.dwl, a.dbl, a.n, P, a.sigmoid);
        GI_LAUNCH_CHECK();
      }
      return GI_OK;
    }
  }
  hipLaunchKernelGGL(head_linear_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, st, a.dy, a.out, a.h, a.wl, a.dh, a.dwl,
                     a.dbl, a.n, P, a.sigmoid);
  GI_LAUNCH_CHECK();
  const int epc = dtype == GI_F16 ? 8 : 4;
  const int64_t total = (int64_t)a.n * a.Hh * a.Wh * (a.c / epc);
  const int grid = grid_for(total, 256, 4096);
  if (dtype == GI_F16)
    hipLaunchKernelGGL(head_dgrad_kernel<half_t>, dim3(grid), dim3(256), 0, st, a.dh, a.w5, (char*)a.da4, a.n, a.Hh, a.Wh, a.c, a.loss_scale);
  else
    hipLaunchKernelGGL(head_dgrad_kernel<float>, dim3(grid), dim3(256), 0, st, a.dh, a.w5, (char*)a.da4, a.n, a.Hh, a.Wh, a.c, a.loss_scale);
  GI_LAUNCH_CHECK();
  if (a.dw5) {
    dim3 g((16 * (a.c / epc) + 255) / 256, a.n, Ph >= 8 ? 2 : 1);
    const int nparts = a.n * (int)g.z;
    float* part = (a.scratch && a.scratch_bytes >= (int64_t)nparts * 16 * a.c * 4) ? a.scratch : nullptr;
    if (dtype == GI_F16)
      hipLaunchKernelGGL(head_wgrad_kernel<half_t>, g, dim3(256), 0, st, a.dh, (const char*)a.a4, a.dw5, a.Hh, a.Wh, a.c, part);
    else
      hipLaunchKernelGGL(head_wgrad_kernel<float>, g, dim3(256), 0, st, a.dh, (const char*)a.a4, a.dw5, a.Hh, a.Wh, a.c, part);
    GI_LAUNCH_CHECK();
    if (part) {
      hipLaunchKernelGGL(head_wgrad_sum_kernel, dim3((16 * a.c + 255) / 256), dim3(256), 0, st, part, nparts, a.dw5, 16 * a.c);
      GI_LAUNCH_CHECK();
    }
  }
  return GI_OK;
}

// The 4-channel head of a narrow generator (fp16, c = 128 input channels): forward (col + col2im, bias + tanh) and input gradient.
bool op_c1_head4_ok(int dtype, int c, int out_c, int Ws, int ldx, int coffx) {
  return dtype == GI_F16 && c == 128 && out_c == 4 && Ws % 16 == 0 && ldx % 8 == 0 && coffx % 8 == 0;
}
int64_t op_c1_head4_col_bytes(int n, int Hs, int Ws) { return (int64_t)n * Hs * Ws * 16 * 4 * 2; }
int op_c1_head4_forward(hipStream_t st, const void* X, const float* w, const float* bias, float* out, float* out2, int n, int Hs, int Ws, int ldx,
                        int coffx, int relu_in, void* col_scratch) {
  const int64_t P = (int64_t)n * Hs * Ws;
  hipLaunchKernelGGL((c1_col_mc_kernel<4, 4>), dim3(grid_for((P + 15) / 16, 4, 256 * 8)), dim3(256), 0, st, (const char*)X, w, (half_t*)col_scratch, P,
                     ldx, coffx, relu_in);
  GI_LAUNCH_CHECK();
  hipLaunchKernelGGL(c1_col2im_mc_kernel<4>, dim3(grid_for(P * 4, 256, 256 * 8)), dim3(256), 0, st, (const half_t*)col_scratch, bias, out, out2, n, Hs,
                     Ws, 1);
  GI_LAUNCH_CHECK();
  return GI_OK;
}
int op_c1_head4_dgrad(hipStream_t st, const float* g, const float* w, void* out, int n, int Hs, int Ws, int ldout, int coffout) {
  GI_REQUIRE(Ws % 16 == 0 && ldout % 8 == 0 && coffout % 8 == 0, "c1_head4_dgrad: layout");
  const int64_t ngroups = (int64_t)n * Hs * Ws / 16;
  hipLaunchKernelGGL(c1_gather_mc4_kernel<8>, dim3(grid_for(ngroups, 4, 256 * 8)), dim3(256), 0, st, g, w, (char*)out, n, Hs, Ws, ldout, coffout);
  GI_LAUNCH_CHECK();
  return GI_OK;
}
