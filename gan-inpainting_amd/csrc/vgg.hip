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
//   taps     perceptual term mean((F_o - F_t)^2) (fp32 partials, fixed-order finish) and the Gram matrices
//            F^T F / (H W C) as an MFMA GEMM with K = pixels (both operands pixel-major: transposing LDS
//            reads as in wgrad.hip), then mean((G_o - G_t)^2).
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

// out[img,y,x,0:64] = relu(b + sum_tap w1[.][tap] * x[img, y-1+ky, x-1+kx]); images [0,n) from xa, [n,2n) from xb
__global__ void __launch_bounds__(256) vgg_conv1_kernel(const float* __restrict__ xa, const float* __restrict__ xb, int n, int H, int W,
                                                        const float* __restrict__ w1, const float* __restrict__ bias,
                                                        half_t* __restrict__ out) {
  const int grp = threadIdx.x & 7;   // 8 channels each: their 72 weights and 8 biases stay in registers
  float wr[8][9], br[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    br[c] = bias[grp * 8 + c];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[c][k] = w1[(grp * 8 + c) * 9 + k];
  }
  const int64_t npix = (int64_t)2 * n * H * W;
  for (int64_t pix = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (int64_t)gridDim.x * 32) {
    const int x = (int)(pix % W);
    const int64_t t = pix / W;
    const int y = (int)(t % H);
    const int img = (int)(t / H);
    const float* src = (img < n ? xa + (int64_t)img * H * W : xb + (int64_t)(img - n) * H * W);
    float v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int iy = y - 1 + k / 3, ix = x - 1 + k % 3;
      v[k] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? src[(int64_t)iy * W + ix] : 0.f;
    }
    h8_t o;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float s = br[c];
#pragma unroll
      for (int k = 0; k < 9; ++k) s = fmaf(wr[c][k], v[k], s);
      o[c] = (half_t)(s > 0.f ? s : 0.f);
    }
    *(h8_t*)(out + pix * 64 + grp * 8) = o;
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

// ---- Gram: G[img][c1][c2] += scale * sum_p F[img][p][c1] * F[img][p][c2] --------------------------------
struct GramP {
  const char* F;     // [nimg][HW][C] fp16
  float* G;          // [nimg][C][C]
  int C, HW, tiles_per_split;
  float scale;
};
// 64 x 64 output tile, K tile = 64 pixels, 4 waves as 2x2 of 32x32
__global__ void __launch_bounds__(256) gram_kernel(GramP p) {
  constexpr int LROW = 128 + 32;       // 64 channels of fp16 + 32 B pad (conflict-free transposing reads)
  constexpr int OPB = 64 * LROW;
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * OPB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ct = p.C / 64;
  const int c1 = (blockIdx.x / ct) * 64, c2 = (blockIdx.x % ct) * 64;
  const int img = blockIdx.z;
  const char* F = p.F + (int64_t)img * p.HW * p.C * 2;
  const int t_total = (p.HW + 63) / 64;
  const int t_begin = blockIdx.y * p.tiles_per_split;
  const int t_end = min(t_total, t_begin + p.tiles_per_split);
  const int chunk = tid & 7, rbase = tid >> 3;   // 8 chunks of 16 B per 64-channel row, 32 rows per pass

  u4_t ra[2], rb[2];
  auto gload = [&](int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pix = t * 64 + rbase + 32 * i;
      u4_t va = u4_t{0u, 0u, 0u, 0u}, vb = u4_t{0u, 0u, 0u, 0u};
      if (pix < p.HW) {
        va = *(const u4_t*)(F + ((int64_t)pix * p.C + c1 + chunk * 8) * 2);
        vb = *(const u4_t*)(F + ((int64_t)pix * p.C + c2 + chunk * 8) * 2);
      }
      ra[i] = va;
      rb[i] = vb;
    }
  };
  auto lds_store = [&](int stage) {
    char* sA = smem + stage * 2 * OPB;
    char* sB = sA + OPB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = rbase + 32 * i;
      *(u4_t*)(sA + r * LROW + chunk * 16) = ra[i];
      *(u4_t*)(sB + r * LROW + chunk * 16) = rb[i];
    }
  };
  f4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  auto frag = [&](const char* base, int ch) -> h8_t {
    // ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(channel) block (see wgrad.hip)
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    const char* lo_p = base + (8 * g + q) * LROW + (ch + 4 * pp) * 2;
    const char* hi_p = base + (8 * g + 4 + q) * LROW + (ch + 4 * pp) * 2;
    fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)lo_p);
    fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)hi_p);
    h4_t l4 = __builtin_bit_cast(h4_t, lo), h4 = __builtin_bit_cast(h4_t, hi);
    return h8_t{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
  };
  auto compute = [&](int stage) {
    const char* sA = smem + stage * 2 * OPB;
    const char* sB = sA + OPB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h8_t af[2], bf[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) af[mt] = frag(sA + ks * 32 * LROW, wm * 32 + mt * 16);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bf[nt] = frag(sB + ks * 32 * LROW, wn * 32 + nt * 16);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
    }
  };
  if (t_begin < t_end) {
    gload(t_begin);
    lds_store(0);
    __syncthreads();
    int stage = 0;
    for (int t = t_begin; t < t_end; ++t) {
      const bool more = t + 1 < t_end;
      if (more) gload(t + 1);
      compute(stage);
      if (more) lds_store(stage ^ 1);
      __syncthreads();
      stage ^= 1;
    }
    float* G = p.G + (int64_t)img * p.C * p.C;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = c1 + wm * 32 + mt * 16 + (lane >> 4) * 4 + r;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int col = c2 + wn * 32 + nt * 16 + (lane & 15);
          atomicAdd(G + (int64_t)row * p.C + col, acc[mt][nt][r] * p.scale);
        }
      }
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
// partial[b] = sum over the first `half` elements of (a[i] - a[i + half])^2
template <typename T>
__global__ void __launch_bounds__(256) sqdiff_halves_kernel(const T* __restrict__ a, int64_t half, double* __restrict__ partial) {
  __shared__ double sh[4];
  double s = 0.0;
  if constexpr (std::is_same<T, half_t>::value) {
    const int64_t n8 = half / 8;   // half is a multiple of 8 (channels >= 64)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
      const h8_t x = *(const h8_t*)(a + i * 8), y = *(const h8_t*)(a + half + i * 8);
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = (float)x[e] - (float)y[e]; acc = fmaf(d, d, acc); }
      s += (double)acc;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < half; i += (int64_t)gridDim.x * 256) {
      const float d = a[i] - a[i + half];
      s += (double)d * d;
    }
  }
  s = blk_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) sqdiff_final_kernel(const double* __restrict__ partial, int nb, double inv_count, float* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  s = blk_sum(s, sh);
  if (threadIdx.x == 0) out[0] = (float)(s * inv_count);
}
// out2[0] = wp * sum_t per_tap[t] ; out2[1] = ws * sum_t per_tap[5 + t]   (loss.py:106-115)
__global__ void vgg_combine_kernel(const float* __restrict__ per_tap, float wp, float ws, float* __restrict__ out2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float p = 0.f, s = 0.f;
    for (int t = 0; t < 5; ++t) { p += per_tap[t]; s += per_tap[5 + t]; }
    out2[0] = wp * p;
    out2[1] = ws * s;
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
  float* gram;            // [2n][512][512] max
  double* partial;        // 1024 doubles
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
  v->gram = (float*)take((int64_t)2 * v->max_pairs * 512 * 512 * 4);
  v->partial = (double*)take(1024 * 8);
  v->per_tap = (float*)take(64);
  return off;
}

int vgg_run(gi_vgg* v, const float* xa, const float* xb, int n, int stop_tap, float* feat_out) {
  hipStream_t st = v->ctx->stream;
  const int nimg = 2 * n;
  int H = v->H, W = v->W, cur = 0;
  hipLaunchKernelGGL(vgg_conv1_kernel, dim3(nblk((int64_t)nimg * H * W, 32, 256 * 16)), dim3(256), 0, st, xa, xb, n, H, W, v->w1,
                     v->params + v->boff[0], v->act[0]);
  GI_LAUNCH_CHECK();
  for (int i = 0; i < NCONV; ++i) {
    if (i > 0) {
      IgemmArgs a = {};
      a.in = v->act[cur]; a.w = v->wpk[i]; a.out = v->act[cur ^ 1];
      a.bias = v->params + v->boff[i]; a.partials = nullptr; a.ws = nullptr; a.ws_bytes = 0;
      a.n = nimg; a.Hs = H; a.Ws = W;
      a.cin = kCin[i]; a.ldin = kCin[i]; a.coffin = 0;
      a.cout = kCout[i]; a.ldout = kCout[i]; a.coffout = 0;
      a.relu_in = 0; a.relu_cend = 0; a.act_out = GI_ACT_RELU; a.force_splitk = 0;
      GI_TRY(op_igemm3(st, 2, a));
      cur ^= 1;
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
        // perceptual term: mean over (n, C, H, W) of (F_o - F_t)^2
        const int64_t half = (int64_t)n * HW * C;
        const int nb = nblk(half / 8, 256 * 4, 1024);
        hipLaunchKernelGGL(sqdiff_halves_kernel<half_t>, dim3(nb), dim3(256), 0, st, F, half, v->partial);
        GI_LAUNCH_CHECK();
        hipLaunchKernelGGL(sqdiff_final_kernel, dim3(1), dim3(256), 0, st, v->partial, nb, 1.0 / (double)half, v->per_tap + tap);
        GI_LAUNCH_CHECK();
        // style term: Gram matrices of all 2n images, then mean over (n, C, C) of (G_o - G_t)^2
        GI_HIP(hipMemsetAsync(v->gram, 0, (size_t)nimg * C * C * 4, st));
        GramP gp;
        gp.F = (const char*)F; gp.G = v->gram; gp.C = C; gp.HW = (int)HW;
        gp.scale = (float)(1.0 / ((double)HW * C));
        const int tiles = (C / 64) * (C / 64) * nimg;
        const int ktiles = (int)((HW + 63) / 64);
        int split = (1024 + tiles - 1) / tiles;
        if (split > ktiles / 8) split = ktiles / 8;
        if (split < 1) split = 1;
        gp.tiles_per_split = (ktiles + split - 1) / split;
        split = (ktiles + gp.tiles_per_split - 1) / gp.tiles_per_split;
        hipLaunchKernelGGL(gram_kernel, dim3((C / 64) * (C / 64), split, nimg), dim3(256), 0, st, gp);
        GI_LAUNCH_CHECK();
        const int64_t ghalf = (int64_t)n * C * C;
        const int nb2 = nblk(ghalf, 256 * 4, 1024);
        hipLaunchKernelGGL(sqdiff_halves_kernel<float>, dim3(nb2), dim3(256), 0, st, (const float*)v->gram, ghalf, v->partial);
        GI_LAUNCH_CHECK();
        hipLaunchKernelGGL(sqdiff_final_kernel, dim3(1), dim3(256), 0, st, v->partial, nb2, 1.0 / (double)ghalf, v->per_tap + 5 + tap);
        GI_LAUNCH_CHECK();
      }
    }
    if (kPoolAfter[i]) {
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
  hipLaunchKernelGGL(vgg_combine_kernel, dim3(1), dim3(64), 0, v->ctx->stream, v->per_tap, weight_p, weight_s, out2);
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
