// VGG-19 feature extractor for the perceptual / style losses of the reference
// (lib/models/loss.py:50-115 perceptual_loss / style_loss / perceptual_and_style_loss, gram_matrix
// :117-136; taps after features[1,6,11,20,29] = relu1_1 .. relu5_1).
//
// Forward only: the reference evaluates these losses under no_grad on detached inputs, so nothing
// reaches the generator. `output` and `target` run as ONE stacked batch of 2n images (NHWC fp16):
//   conv1_1  the input is the grey image repeated over 3 channels (loss.py:54-55), so the 3->64 conv is
//            a 1->64 conv with the weights summed over the input channel: one HBM-bound VALU kernel;
//   12 more  3x3/s1/p1 convolutions + bias + ReLU: igemm3 mode 2 (fp16 MFMA implicit GEMM, LDS-DMA ring);
//   2x2 max pooling: 16-byte NHWC kernel;
//   taps     perceptual term mean((F_o - F_t)^2) and the Gram matrices F^T F / (H W C) (MFMA GEMM with K = pixels,
//            both operands pixel-major: transposing LDS reads as in wgrad.hip) in ONE pass per tap over the
//            image pairs (gram2_kernel: partial tiles, no atomics), then mean((G_o - G_t)^2) from the
//            partial tiles in a fixed order; one finish launch for all ten terms.
#include <new>

#include "common.h"

int op_igemm3(hipStream_t st, int mode, IgemmArgs& a);

namespace {

constexpr int NCONV = 13;
const int kCin[NCONV] = {3, 64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512};
const int kCout[NCONV] = {64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512};
const int kFeatIdx[NCONV] = {0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28};   // torchvision vgg19.features indices
// after conv i: tap index (or -1), then pool?
const int kTapAfter[NCONV] = {0, -1, 1, -1, 2, -1, -1, -1, 3, -1, -1, -1, 4};
const int kPoolAfter[NCONV] = {0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0};

// [cout][cin][3][3] fp32 -> fp16 [cout][tap][cin]
__global__ void __launch_bounds__(256) pack3x3_kernel(const float* __restrict__ w, half_t* __restrict__ out, int cout, int cin) {
  const int64_t total = (int64_t)cout * 9 * cin;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cin);
    const int t = (int)((i / cin) % 9);
    const int o = (int)(i / ((int64_t)cin * 9));
    out[i] = (half_t)w[((int64_t)o * cin + c) * 9 + t];
  }
}
// conv1_1 on a channel-replicated grey image: w1[o][tap] = sum_c w[o][c][tap]
__global__ void __launch_bounds__(256) sum_cin_kernel(const float* __restrict__ w, float* __restrict__ w1, int cout, int cin) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cout * 9) return;
  const int o = i / 9, t = i % 9;
  float s = 0.f;
  for (int c = 0; c < cin; ++c) s += w[((int64_t)o * cin + c) * 9 + t];
  w1[i] = s;
}

// out[img,y,x,0:64] = relu(b + sum_tap w1[.][tap] * x[img, y-1+ky, x-1+kx]); images [0,n) from xa, [n,2n) from xb.
// On the matrix cores: D[channel][pixel] = A[channel][tap] * B[tap][pixel] with K = 9 taps padded to 16
// (v_mfma_f32_16x16x16_f16, fp32 accumulate; image and summed weights rounded to fp16 - the layer's output is stored in fp16
// anyway). A wave owns 16 consecutive pixels of an image row; lane (pixel, k-chunk kq) loads taps 4 kq .. 4 kq + 3 from clamped
// addresses (selects, no divergent branches; the group index is wave-uniform: scalar 32-bit index arithmetic). Channel order per
// tile as in c1_gather_mfma_kernel: two 16-byte stores per lane, 64 contiguous bytes per pixel and instruction. The VALU form of
// rounds 1-2 ran at 2 TB/s of stores (72 FMAs and 9 guarded loads per 16 output bytes); this layer is 537 MB of stores at 512x512.
__global__ void __launch_bounds__(256) vgg_conv1_mfma_kernel(const float* __restrict__ xa, const float* __restrict__ xb, int n, int H, int W,
                                                             const float* __restrict__ w1, const float* __restrict__ bias,
                                                             half_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int lr = lane & 15, kq = lane >> 4;
  h4_t af[4];
  float br[4][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int ch = ((mt >> 1) & 1) * 32 + (lr >> 2) * 8 + (mt & 1) * 4 + (lr & 3);     // A row lr of tile mt
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int t = 4 * kq + j; af[mt][j] = t < 9 ? (half_t)w1[ch * 9 + t] : (half_t)0.f; }
    // D row i = 4 kq + r of tile mt is channel ((mt >> 1) & 1) * 32 + kq * 8 + (mt & 1) * 4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r) br[mt][r] = bias[((mt >> 1) & 1) * 32 + kq * 8 + (mt & 1) * 4 + r];
  }
  const int gpr = W >> 4;                                   // 16-pixel groups per image row
  const int ngroups = 2 * n * H * gpr;                      // < 2^31 (host check)
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
  const int nwaves = (int)gridDim.x * 4;
  int dy[4], dx[4];                                         // this lane's taps: (ky, kx) - 1, or far outside for the padding of K
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = 4 * kq + j;
    dy[j] = t < 9 ? t / 3 - 1 : -(1 << 20);
    dx[j] = t < 9 ? t % 3 - 1 : 0;
  }
  auto load_b = [&](int g, float (&bv)[4]) {
    const int rowi = g / gpr;                               // wave-uniform
    const int x = (g - rowi * gpr) * 16 + lr;
    const int img = rowi / H, y = rowi - img * H;
    const float* src = img < n ? xa + (int64_t)img * H * W : xb + (int64_t)(img - n) * H * W;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy = y + dy[j], ix = x + dx[j];
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const float v = src[ok ? iy * W + ix : 0];
      bv[j] = ok ? v : 0.f;
    }
  };
  float bv[4], bn[4];
  int g = wave;
  if (g < ngroups) load_b(g, bv);
  for (; g < ngroups; g += nwaves) {
    if (g + nwaves < ngroups) load_b(g + nwaves, bn);
    const h4_t bf = {(half_t)bv[0], (half_t)bv[1], (half_t)bv[2], (half_t)bv[3]};
    h8_t o[2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      f4_t acc = {br[mt][0], br[mt][1], br[mt][2], br[mt][3]};
      acc = __builtin_amdgcn_mfma_f32_16x16x16f16(af[mt], bf, acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[mt >> 1][(mt & 1) * 4 + r] = (half_t)fmaxf(acc[r], 0.f);
    }
    half_t* dst = out + ((int64_t)g * 16 + lr) * 64;
    *(h8_t*)(dst + kq * 8) = o[0];
    *(h8_t*)(dst + 32 + kq * 8) = o[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = bn[j];
  }
}

// NHWC fp16 2x2 / stride 2 max pooling, 8 channels per thread
__global__ void __launch_bounds__(256) maxpool2_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int nimg, int Ho, int Wo,
                                                       int C) {
  const int cg = C / 8;
  const int64_t total = (int64_t)nimg * Ho * Wo * cg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int g = (int)(i % cg);
    const int64_t p = i / cg;
    const int x = (int)(p % Wo);
    const int64_t t = p / Wo;
    const int y = (int)(t % Ho);
    const int64_t img = t / Ho;
    const half_t* s = in + (((img * 2 * Ho + 2 * y) * 2 * Wo + 2 * x) * (int64_t)C) + g * 8;
    const h8_t a = *(const h8_t*)s, b = *(const h8_t*)(s + C);
    const h8_t c = *(const h8_t*)(s + (int64_t)2 * Wo * C), d = *(const h8_t*)(s + (int64_t)2 * Wo * C + C);
    *(h8_t*)(out + p * C + g * 8) = __builtin_elementwise_max(__builtin_elementwise_max(a, b), __builtin_elementwise_max(c, d));
  }
}

__device__ __forceinline__ double blk_sum(double v, double* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// ---- Gram + perceptual term in one pass over a tap's feature maps (deterministic) ------------------------------------------------
// A workgroup owns one (c1 block, c2 block) tile of the Gram matrix (blocks of BLK = 2 WT channels, c1 block <= c2 block: the matrix
// is symmetric), one pixel range (split) and one image PAIR (output image i, target image i + n): per 32-pixel step it stages the
// 32 x BLK slices of both images (one slice on diagonal tiles), multiplies both Grams on the matrix cores (four waves, WT x WT each,
// transposing LDS reads as in the kernel above) and - on diagonal tiles, where every channel slice passes exactly once - adds up
// (F_o - F_t)^2 from the staging registers. Partial Grams go to P[pair][o|t][tile][split][BLK*BLK] with plain stores;
// gram2_diff_kernel adds the splits in a fixed order and reduces (G_o - G_t)^2 (off-diagonal tiles count twice). No atomics, no
// memset, the feature maps are read C / BLK (+1) / 2 times instead of 2 C / 64 times + once more for the perceptual term.
struct Gram2P {
  const char* F;      // [2n][HW][C] fp16
  float* P;           // partial Grams
  double* sq;         // [pairs][nblk][split] partial sums of (F_o - F_t)^2
  int C, HW, n, nblk, ntile, split, steps_per_split;
};
template <int WT>
__global__ void __launch_bounds__(256) gram2_kernel(Gram2P p) {
  constexpr int BLK = 2 * WT, ROWB = BLK * 2, LROW = ROWB + 32, SL = 32 * LROW;   // one staged slice: 32 pixels
  constexpr int CPR = ROWB / 16, RPP = 256 / CPR, NL = 32 / RPP;                  // 16-byte chunks per row, rows per pass, passes
  constexpr int MT = WT / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];                     // [2 stages][o c1, o c2, t c1, t c2][SL]
  __shared__ double shd[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bi = 0, bj = 0;
  { int t = blockIdx.x; for (bi = 0; t >= p.nblk - bi; ++bi) t -= p.nblk - bi; bj = bi + t; }   // tile index -> (bi <= bj)
  const bool diag = bi == bj;
  const int pair = blockIdx.z, sp = blockIdx.y;
  const char* Fo = p.F + (int64_t)pair * p.HW * p.C * 2;
  const char* Ft = p.F + (int64_t)(pair + p.n) * p.HW * p.C * 2;
  const int steps = (p.HW + 31) / 32;
  const int s0 = sp * p.steps_per_split, s1 = min(steps, s0 + p.steps_per_split);
  const int chunk = tid % CPR, rbase = tid / CPR;
  u4_t r[4][NL];      // [o c1, o c2, t c1, t c2]
  auto gload = [&](int st) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int pix = st * 32 + rbase + RPP * i;
      const bool ok = pix < p.HW;
      const int64_t o1 = ((int64_t)pix * p.C + bi * BLK + chunk * 8) * 2, o2 = ((int64_t)pix * p.C + bj * BLK + chunk * 8) * 2;
      const u4_t z = u4_t{0u, 0u, 0u, 0u};
      r[0][i] = ok ? *(const u4_t*)(Fo + o1) : z;
      r[2][i] = ok ? *(const u4_t*)(Ft + o1) : z;
      if (!diag) { r[1][i] = ok ? *(const u4_t*)(Fo + o2) : z; r[3][i] = ok ? *(const u4_t*)(Ft + o2) : z; }
    }
  };
  float sq = 0.f;
  auto lds_store = [&](int stage) {
    char* base = smem + stage * 4 * SL;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int off = (rbase + RPP * i) * LROW + chunk * 16;
      *(u4_t*)(base + 0 * SL + off) = r[0][i];
      *(u4_t*)(base + 2 * SL + off) = r[2][i];
      if (!diag) { *(u4_t*)(base + 1 * SL + off) = r[1][i]; *(u4_t*)(base + 3 * SL + off) = r[3][i]; }
      else {
        const h8_t a = __builtin_bit_cast(h8_t, r[0][i]), b = __builtin_bit_cast(h8_t, r[2][i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = (float)a[e] - (float)b[e]; sq = fmaf(d, d, sq); }
      }
    }
  };
  f4_t acc[2][MT][MT];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[q][i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
  auto frag = [&](const char* base, int ch) -> h8_t {      // 32 (k = pixels) x 16 (channels ch ..) as an MFMA operand
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    const char* lo_p = base + (8 * g + q) * LROW + (ch + 4 * pp) * 2;
    const char* hi_p = base + (8 * g + 4 + q) * LROW + (ch + 4 * pp) * 2;
    fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)lo_p);
    fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)hi_p);
    h4_t l4 = __builtin_bit_cast(h4_t, lo), h4 = __builtin_bit_cast(h4_t, hi);
    return h8_t{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
  };
  auto compute = [&](int stage) {
    const char* base = smem + stage * 4 * SL;
#pragma unroll
    for (int q = 0; q < 2; ++q) {       // output image, target image
      const char* sA = base + (2 * q) * SL;
      const char* sB = diag ? sA : sA + SL;
      h8_t af[MT], bf[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[mt] = frag(sA, wm * WT + mt * 16);
#pragma unroll
      for (int nt = 0; nt < MT; ++nt) bf[nt] = frag(sB, wn * WT + nt * 16);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < MT; ++nt) acc[q][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bf[nt], acc[q][mt][nt], 0, 0, 0);
    }
  };
  // (a second register set holding step st + 2 while st + 1 waits was measured slower: 54 -> 87 us average on the 128-channel taps)
  if (s0 < s1) {
    gload(s0);
    lds_store(0);
    __syncthreads();
    int stage = 0;
    for (int st = s0; st < s1; ++st) {
      const bool more = st + 1 < s1;
      if (more) gload(st + 1);
      compute(stage);
      if (more) lds_store(stage ^ 1);
      __syncthreads();
      stage ^= 1;
    }
  }
  // partial Grams: P[((pair * 2 + q) * ntile + tile) * split + sp][BLK][BLK]
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float* P = p.P + ((((int64_t)pair * 2 + q) * p.ntile + blockIdx.x) * p.split + sp) * (BLK * BLK);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = wm * WT + mt * 16 + (lane >> 4) * 4 + rr;
#pragma unroll
        for (int nt = 0; nt < MT; ++nt) P[row * BLK + wn * WT + nt * 16 + (lane & 15)] = acc[q][mt][nt][rr];
      }
  }
  if (diag) {
    double s = (double)sq;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) shd[wave] = s;
    __syncthreads();
    if (tid == 0) p.sq[((int64_t)pair * p.nblk + bi) * p.split + sp] = (shd[0] + shd[1]) + (shd[2] + shd[3]);
  }
}
// partial[block] = sum over the block's groups of 64 Gram elements (grid-stride) of w * ((sum_s P_o - sum_s P_t) * scale)^2: thread
// (element, part) adds a quarter of the splits (contiguous range, ascending), the four parts are added in ascending order; w = 2 on
// off-diagonal tiles (their mirror images are never computed). bb = BLK * BLK is a multiple of 64: a group lies inside one tile.
__global__ void __launch_bounds__(256) gram2_diff_kernel(const float* __restrict__ P, int n, int nblk, int ntile, int split, int bb,
                                                         float scale, double* __restrict__ partial) {
  __shared__ float red[2][3][64];
  __shared__ double sh[4];
  const int e64 = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int64_t per_pair = (int64_t)ntile * bb;
  const int64_t groups = (int64_t)n * per_pair / 64;
  const int per = (split + 3) / 4, k0 = part * per, k1 = min(split, k0 + per);
  double s = 0.0;
  for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {
    const int64_t i = g * 64 + e64;
    const int pair = (int)(i / per_pair);
    const int64_t rem = i - (int64_t)pair * per_pair;
    const int tile = (int)(rem / bb), e = (int)(rem - (int64_t)tile * bb);
    const float* po = P + ((((int64_t)pair * 2 + 0) * ntile + tile) * split) * bb + e;
    const float* pt = P + ((((int64_t)pair * 2 + 1) * ntile + tile) * split) * bb + e;
    float go = 0.f, gt = 0.f;
    for (int k = k0; k < k1; ++k) { go += po[(int64_t)k * bb]; gt += pt[(int64_t)k * bb]; }
    __syncthreads();                 // the previous group's sums have been read
    if (part > 0) { red[0][part - 1][e64] = go; red[1][part - 1][e64] = gt; }
    __syncthreads();
    if (part == 0) {
      go = ((go + red[0][0][e64]) + red[0][1][e64]) + red[0][2][e64];
      gt = ((gt + red[1][0][e64]) + red[1][1][e64]) + red[1][2][e64];
      int t = tile, bi = 0;
      for (; t >= nblk - bi; ++bi) t -= nblk - bi;
      const double d = (double)(go * scale) - (double)(gt * scale);
      s += (t == 0 ? 1.0 : 2.0) * d * d;
    }
  }
  s = blk_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
struct VggFinP { int nb[10]; double inv[10]; };
// per_tap[k] = inv[k] * sum of the nb[k] partials of term k (ascending), then out2 = (wp * sum of the 5 perceptual terms, ws * sum of the 5 style terms)
__global__ void __launch_bounds__(256) vgg_finish_kernel(const double* __restrict__ partial, VggFinP f, float* __restrict__ per_tap, float wp, float ws,
                                                         float* __restrict__ out2) {
  __shared__ double sh[4];
  __shared__ float terms[10];
  for (int k = 0; k < 10; ++k) {
    double s = 0.0;
    for (int i = threadIdx.x; i < f.nb[k]; i += 256) s += partial[k * 1024 + i];
    s = blk_sum(s, sh);
    if (threadIdx.x == 0) { terms[k] = (float)(s * f.inv[k]); per_tap[k] = terms[k]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float pp = 0.f, ss = 0.f;
    for (int t = 0; t < 5; ++t) { pp += terms[t]; ss += terms[5 + t]; }
    out2[0] = wp * pp;
    out2[1] = ws * ss;
  }
}
// NHWC fp16 feature map -> NCHW fp32 (parity / debugging)
__global__ void __launch_bounds__(256) nhwc_to_nchw_f32_kernel(const half_t* __restrict__ in, float* __restrict__ out, int nimg, int HW, int C) {
  const int64_t total = (int64_t)nimg * HW * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const int64_t t = i / HW;
    const int c = (int)(t % C);
    const int64_t img = t / C;
    out[i] = (float)in[(img * HW + p) * C + c];
  }
}

int nblk(int64_t work, int per, int cap) {
  int64_t b = (work + per - 1) / per;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

struct gi_vgg {
  gi_ctx* ctx;
  int H, W, max_pairs;
  int64_t woff[NCONV], boff[NCONV];   // float offsets into params
  int64_t param_floats;
  const float* params;
  // workspace carve-up
  char* ws;
  int64_t ws_bytes;
  half_t* act[2];
  half_t* wpk[NCONV];     // [1..12] packed fp16 weights
  float* w1;              // conv1_1 summed weights [64][9]
  float* gram;            // partial Gram tiles (gram2_kernel): gram_bytes
  int64_t gram_bytes;
  double* partial;        // [10 terms][1024] doubles
  VggFinP fin;            // per term: partial count and 1 / element count (filled by vgg_run)
  float* per_tap;         // 10 floats
  bool bound, synced;
};

namespace {

int64_t vgg_ws_layout(gi_vgg* v, char* base) {
  int64_t off = 0;
  auto take = [&](int64_t bytes) { char* p = base ? base + off : nullptr; off += gi_align_up(bytes, 256); return p; };
  const int64_t act_bytes = (int64_t)2 * v->max_pairs * v->H * v->W * 64 * 2;
  v->act[0] = (half_t*)take(act_bytes);
  v->act[1] = (half_t*)take(act_bytes);
  v->wpk[0] = nullptr;
  for (int i = 1; i < NCONV; ++i) v->wpk[i] = (half_t*)take((int64_t)kCout[i] * 9 * kCin[i] * 2);
  v->w1 = (float*)take(64 * 9 * 4);
  v->gram_bytes = (int64_t)2 * v->max_pairs * 10 * 65536;          // one split of the widest tap (ten 128 x 128 tiles per image)
  if (v->gram_bytes < (32ll << 20)) v->gram_bytes = 32ll << 20;
  v->gram = (float*)take(v->gram_bytes);
  v->partial = (double*)take(10 * 1024 * 8);
  v->per_tap = (float*)take(64);
  return off;
}

int vgg_run(gi_vgg* v, const float* xa, const float* xb, int n, int stop_tap, float* feat_out) {
  hipStream_t st = v->ctx->stream;
  const int nimg = 2 * n;
  int H = v->H, W = v->W, cur = 0;
  // (W % 16 == 0 and 2 n H W 64 < 2^31 are conditions of gi_vgg19_create)
  hipLaunchKernelGGL(vgg_conv1_mfma_kernel, dim3(nblk((int64_t)nimg * H * (W / 16), 4, 256 * 8)), dim3(256), 0, st, xa, xb, n, H, W, v->w1,
                     v->params + v->boff[0], v->act[0]);
  GI_LAUNCH_CHECK();
  for (int i = 0; i < NCONV; ++i) {
    bool pooled = false;
    if (i > 0) {
      IgemmArgs a = {};
      a.in = v->act[cur]; a.w = v->wpk[i]; a.out = v->act[cur ^ 1];
      a.bias = v->params + v->boff[i]; a.partials = nullptr; a.ws = nullptr; a.ws_bytes = 0;
      a.n = nimg; a.Hs = H; a.Ws = W;
      a.cin = kCin[i]; a.ldin = kCin[i]; a.coffin = 0;
      a.cout = kCout[i]; a.ldout = kCout[i]; a.coffout = 0;
      a.relu_in = 0; a.relu_cend = 0; a.act_out = GI_ACT_RELU; a.force_splitk = 0;
      // the pooled layers (conv1_2, conv2_2, conv3_4, conv4_4) are never taps: where the kernel can, it stores the 2x2 max pool of
      // its tile and the full-resolution map is never written
      a.pool2 = (kPoolAfter[i] && kTapAfter[i] < 0) ? 1 : 0;
      GI_TRY(op_igemm3(st, 2, a));
      cur ^= 1;
      pooled = a.pool_applied != 0;
    }
    const int C = kCout[i];
    const int tap = kTapAfter[i];
    if (tap >= 0) {
      const half_t* F = v->act[cur];
      const int64_t HW = (int64_t)H * W;
      if (feat_out && tap == stop_tap) {
        hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(nblk((int64_t)n * HW * C, 256, 4096)), dim3(256), 0, st, F, feat_out, n, (int)HW, C);
        GI_LAUNCH_CHECK();
        return GI_OK;
      }
      if (!feat_out) {
        // perceptual term mean over (n, C, H, W) of (F_o - F_t)^2 and style term mean over (n, C, C) of (G_o - G_t)^2, G = F^T F / (HW C):
        // one pass over the tap's feature maps (gram2_kernel) + the fixed-order reduction of its partial tiles
        const int BLK = C >= 128 ? 128 : 64, nb_ = C / BLK, ntile = nb_ * (nb_ + 1) / 2;
        const int steps = (int)((HW + 31) / 32);
        int split = (512 + n * ntile - 1) / (n * ntile);
        if (split > steps / 4) split = steps / 4;
        while (split > 1 && ((int64_t)2 * n * ntile * split * BLK * BLK * 4 > v->gram_bytes || n * nb_ * split > 1024)) --split;
        if (split < 1) split = 1;
        GI_REQUIRE((int64_t)2 * n * ntile * split * BLK * BLK * 4 <= v->gram_bytes && n * nb_ * split <= 1024, "vgg19: %d pairs exceed the Gram workspace", n);
        Gram2P gp;
        gp.F = (const char*)F; gp.P = v->gram; gp.sq = v->partial + tap * 1024;
        gp.C = C; gp.HW = (int)HW; gp.n = n; gp.nblk = nb_; gp.ntile = ntile;
        gp.steps_per_split = (steps + split - 1) / split;
        split = (steps + gp.steps_per_split - 1) / gp.steps_per_split;
        gp.split = split;
        if (BLK == 128) {
          static GiDevOnce attr;
          if (attr.first()) { GI_HIP(hipFuncSetAttribute((const void*)gram2_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 32 * 288)); }
          hipLaunchKernelGGL(gram2_kernel<64>, dim3(ntile, split, n), dim3(256), 2 * 4 * 32 * 288, st, gp);
        } else {
          hipLaunchKernelGGL(gram2_kernel<32>, dim3(ntile, split, n), dim3(256), 2 * 4 * 32 * 160, st, gp);
        }
        GI_LAUNCH_CHECK();
        v->fin.nb[tap] = n * nb_ * split;
        v->fin.inv[tap] = 1.0 / (double)((int64_t)n * HW * C);
        const int nb2 = nblk((int64_t)n * ntile * BLK * BLK / 64, 1, 1024);
        hipLaunchKernelGGL(gram2_diff_kernel, dim3(nb2), dim3(256), 0, st, (const float*)v->gram, n, nb_, ntile, split, BLK * BLK,
                           (float)(1.0 / ((double)HW * C)), v->partial + (5 + tap) * 1024);
        GI_LAUNCH_CHECK();
        v->fin.nb[5 + tap] = nb2;
        v->fin.inv[5 + tap] = 1.0 / (double)((int64_t)n * C * C);
      }
    }
    if (kPoolAfter[i] && pooled) {
      H /= 2;
      W /= 2;
    } else if (kPoolAfter[i]) {
      hipLaunchKernelGGL(maxpool2_kernel, dim3(nblk((int64_t)nimg * (H / 2) * (W / 2) * (C / 8), 256, 8192)), dim3(256), 0, st, v->act[cur],
                         v->act[cur ^ 1], nimg, H / 2, W / 2, C);
      GI_LAUNCH_CHECK();
      cur ^= 1;
      H /= 2;
      W /= 2;
    }
  }
  return GI_OK;
}

}  // namespace

extern "C" {

int gi_vgg19_create(gi_ctx* ctx, int H, int W, int max_pairs, gi_vgg** out) {
  GI_REQUIRE(ctx && out, "vgg19_create: null argument");
  GI_REQUIRE(H >= 16 && W >= 16 && H % 16 == 0 && W % 16 == 0 && max_pairs > 0, "vgg19_create: H=%d W=%d (multiples of 16) pairs=%d", H, W,
             max_pairs);
  GI_REQUIRE((int64_t)2 * max_pairs * H * W * 64 < (1ll << 31), "vgg19_create: 2*%d images of %dx%d exceed 32-bit activation offsets", max_pairs,
             H, W);
  gi_vgg* v = new (std::nothrow) gi_vgg();
  GI_REQUIRE(v, "vgg19_create: out of host memory");
  v->ctx = ctx; v->H = H; v->W = W; v->max_pairs = max_pairs;
  int64_t off = 0;
  for (int i = 0; i < NCONV; ++i) {
    v->woff[i] = off; off += (int64_t)kCout[i] * kCin[i] * 9;
    v->boff[i] = off; off += kCout[i];
  }
  v->param_floats = off;
  v->params = nullptr; v->ws = nullptr; v->bound = false; v->synced = false;
  v->ws_bytes = vgg_ws_layout(v, nullptr);
  *out = v;
  return GI_OK;
}
void gi_vgg19_destroy(gi_vgg* v) { delete v; }
int64_t gi_vgg19_param_floats(const gi_vgg* v) { return v ? v->param_floats : -1; }
int64_t gi_vgg19_workspace_bytes(const gi_vgg* v) { return v ? v->ws_bytes : -1; }
int gi_vgg19_num_tensors(const gi_vgg* v) { return v ? 2 * NCONV : -1; }

int gi_vgg19_tensor_desc(const gi_vgg* v, int index, char* name, int name_cap, int* shape4, int64_t* offset) {
  GI_REQUIRE(v && index >= 0 && index < 2 * NCONV && name && shape4 && offset, "vgg19_tensor_desc: bad argument");
  const int i = index / 2, is_bias = index & 1;
  snprintf(name, name_cap, "features.%d.%s", kFeatIdx[i], is_bias ? "bias" : "weight");
  if (is_bias) { shape4[0] = kCout[i]; shape4[1] = shape4[2] = shape4[3] = 0; *offset = v->boff[i]; }
  else { shape4[0] = kCout[i]; shape4[1] = kCin[i]; shape4[2] = shape4[3] = 3; *offset = v->woff[i]; }
  return GI_OK;
}

int gi_vgg19_bind(gi_vgg* v, const float* params, void* ws, int64_t ws_bytes) {
  GI_REQUIRE(v && params && ws, "vgg19_bind: null argument");
  GI_REQUIRE(ws_bytes >= v->ws_bytes && ((uintptr_t)ws & 255) == 0, "vgg19_bind: workspace %lld bytes (need %lld, 256-byte aligned)",
             (long long)ws_bytes, (long long)v->ws_bytes);
  v->params = params; v->ws = (char*)ws;
  vgg_ws_layout(v, v->ws);
  v->bound = true; v->synced = false;
  return GI_OK;
}

int gi_vgg19_sync_weights(gi_vgg* v) {
  GI_REQUIRE(v && v->bound, "vgg19_sync_weights: not bound");
  hipStream_t st = v->ctx->stream;
  hipLaunchKernelGGL(sum_cin_kernel, dim3(3), dim3(256), 0, st, v->params + v->woff[0], v->w1, 64, 3);
  GI_LAUNCH_CHECK();
  for (int i = 1; i < NCONV; ++i) {
    hipLaunchKernelGGL(pack3x3_kernel, dim3(nblk((int64_t)kCout[i] * 9 * kCin[i], 256, 2048)), dim3(256), 0, st, v->params + v->woff[i],
                       v->wpk[i], kCout[i], kCin[i]);
    GI_LAUNCH_CHECK();
  }
  v->synced = true;
  return GI_OK;
}

int gi_vgg19_perceptual_style(gi_vgg* v, const float* output, const float* target, int n, float weight_p, float weight_s, float* out2,
                              float* per_tap10) {
  GI_REQUIRE(v && v->bound && v->synced, "vgg19_perceptual_style: bind + sync_weights first");
  GI_REQUIRE(output && target && out2 && n > 0 && n <= v->max_pairs, "vgg19_perceptual_style: n=%d (max %d)", n, v->max_pairs);
  GI_TRY(vgg_run(v, output, target, n, -1, nullptr));
  hipLaunchKernelGGL(vgg_finish_kernel, dim3(1), dim3(256), 0, v->ctx->stream, (const double*)v->partial, v->fin, v->per_tap, weight_p, weight_s, out2);
  GI_LAUNCH_CHECK();
  if (per_tap10) GI_HIP(hipMemcpyAsync(per_tap10, v->per_tap, 10 * sizeof(float), hipMemcpyDeviceToDevice, v->ctx->stream));
  return GI_OK;
}

int gi_vgg19_features(gi_vgg* v, const float* x, int n, int tap, float* out_nchw) {
  GI_REQUIRE(v && v->bound && v->synced, "vgg19_features: bind + sync_weights first");
  GI_REQUIRE(x && out_nchw && n > 0 && n <= v->max_pairs && tap >= 0 && tap < 5, "vgg19_features: n=%d tap=%d", n, tap);
  // the runner works on 2n stacked images: feed x twice, return the first n
  GI_TRY(vgg_run(v, x, x, n, tap, out_nchw));
  return GI_OK;
}

}  // extern "C"
