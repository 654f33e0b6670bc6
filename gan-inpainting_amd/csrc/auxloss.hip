// Auxiliary generator losses of BASELINE config 5: total variation
// (lib/models/loss.py:138-151 tv_loss) and the class-weighted cross entropy of the face-parsing term
// (nn.CrossEntropyLoss(weight=[0,1.2,0.7,0.7]), wgan_perceptual_style_faceparsing.py:67-68,212-213),
// each as a fused forward reduction + gradient kernel. HBM-bound elementwise work.
#include "common.h"

namespace {

__device__ __forceinline__ double bsum(double v, double* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// partial[b] = { sum (x[.,j]-x[.,j+1])^2 , sum (x[i,.]-x[i+1,.])^2 }
__global__ void __launch_bounds__(256) tv_partial_kernel(const float* __restrict__ x, int64_t total, int H, int W, double* __restrict__ partial) {
  __shared__ double sh[4];
  double sw = 0.0, shh = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xx = (int)(i % W);
    const int yy = (int)((i / W) % H);
    const float v = x[i];
    if (xx + 1 < W) { const float d = v - x[i + 1]; sw += (double)d * d; }
    if (yy + 1 < H) { const float d = v - x[i + W]; shh += (double)d * d; }
  }
  sw = bsum(sw, sh);
  shh = bsum(shh, sh);
  if (threadIdx.x == 0) { partial[blockIdx.x * 2] = sw; partial[blockIdx.x * 2 + 1] = shh; }
}
__global__ void __launch_bounds__(256) tv_final_kernel(const double* __restrict__ partial, int nb, double cnt_w, double cnt_h, float weight,
                                                       float* __restrict__ loss_out) {
  __shared__ double sh[4];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) { a += partial[i * 2]; b += partial[i * 2 + 1]; }
  a = bsum(a, sh);
  b = bsum(b, sh);
  if (threadIdx.x == 0) loss_out[0] = weight * ((float)(b / cnt_h) + (float)(a / cnt_w));   // tv_weight * (h_variance + w_variance)
}
__global__ void __launch_bounds__(256) tv_grad_kernel(const float* __restrict__ x, int64_t total, int H, int W, float gw, float gh,
                                                      float* __restrict__ grad) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xx = (int)(i % W);
    const int yy = (int)((i / W) % H);
    const float v = x[i];
    float g = 0.f;
    if (xx + 1 < W) g += gw * (v - x[i + 1]);
    if (xx > 0) g -= gw * (x[i - 1] - v);
    if (yy + 1 < H) g += gh * (v - x[i + W]);
    if (yy > 0) g -= gh * (x[i - W] - v);
    grad[i] = g;
  }
}

constexpr int CE_MAXK = 16;
struct CeW {
  float w[CE_MAXK];
};
// partial[b] = { sum_p w[y_p] * (logsumexp(z_p) - z_p[y_p]) , sum_p w[y_p] }
template <int K>
__global__ void __launch_bounds__(256) ce_partial_kernel(const float* __restrict__ z, const int64_t* __restrict__ y, int64_t hw, int64_t npix,
                                                         CeW cw, int ignore_index, double* __restrict__ partial) {
  __shared__ double sh[4];
  double sl = 0.0, sw = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const int64_t lab = y[i];
    if (lab == ignore_index || lab < 0 || lab >= K) continue;
    const int64_t n = i / hw, p = i - n * hw;
    const float* zp = z + n * K * hw + p;
    float v[K], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) { v[k] = zp[(int64_t)k * hw]; mx = fmaxf(mx, v[k]); }
    float se = 0.f, zl = 0.f, wl = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      se += expf(v[k] - mx);
      if (k == (int)lab) { zl = v[k]; wl = cw.w[k]; }
    }
    sl += (double)(wl * ((mx + logf(se)) - zl));
    sw += (double)wl;
  }
  sl = bsum(sl, sh);
  sw = bsum(sw, sh);
  if (threadIdx.x == 0) { partial[blockIdx.x * 2] = sl; partial[blockIdx.x * 2 + 1] = sw; }
}
__global__ void __launch_bounds__(256) ce_final_kernel(const double* __restrict__ partial, int nb, float* __restrict__ loss_out) {
  __shared__ double sh[4];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) { a += partial[i * 2]; b += partial[i * 2 + 1]; }
  a = bsum(a, sh);
  b = bsum(b, sh);
  if (threadIdx.x == 0) { loss_out[0] = (float)(a / b); loss_out[1] = (float)b; }
}
template <int K>
__global__ void __launch_bounds__(256) ce_grad_kernel(const float* __restrict__ z, const int64_t* __restrict__ y, int64_t hw, int64_t npix,
                                                      CeW cw, int ignore_index, const float* __restrict__ loss, float gscale,
                                                      float* __restrict__ grad) {
  const float inv = gscale / loss[1];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const int64_t lab = y[i];
    const int64_t n = i / hw, p = i - n * hw;
    const float* zp = z + n * K * hw + p;
    float* gp = grad + n * K * hw + p;
    if (lab == ignore_index || lab < 0 || lab >= K) {
#pragma unroll
      for (int k = 0; k < K; ++k) gp[(int64_t)k * hw] = 0.f;
      continue;
    }
    float v[K], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) { v[k] = zp[(int64_t)k * hw]; mx = fmaxf(mx, v[k]); }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { v[k] = expf(v[k] - mx); se += v[k]; }
    float wl = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) if (k == (int)lab) wl = cw.w[k];
    const float s = wl * inv / se;
#pragma unroll
    for (int k = 0; k < K; ++k) gp[(int64_t)k * hw] = s * v[k] - (k == (int)lab ? wl * inv : 0.f);
  }
}

int rblocks(int64_t count) {
  int64_t b = (count + 256 * 8 - 1) / (256 * 8);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

template <int K>
int ce_run(gi_ctx* ctx, const float* z, const int64_t* y, int n, int64_t hw, const CeW& cw, int ignore_index, float* loss_out, float* grad,
           float gscale, float* scratch) {
  const int64_t npix = (int64_t)n * hw;
  const int nb = rblocks(npix);
  hipLaunchKernelGGL(ce_partial_kernel<K>, dim3(nb), dim3(256), 0, ctx->stream, z, y, hw, npix, cw, ignore_index, (double*)scratch);
  GI_LAUNCH_CHECK();
  hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(256), 0, ctx->stream, (const double*)scratch, nb, loss_out);
  GI_LAUNCH_CHECK();
  if (grad) {
    hipLaunchKernelGGL(ce_grad_kernel<K>, dim3(nb), dim3(256), 0, ctx->stream, z, y, hw, npix, cw, ignore_index, loss_out, gscale, grad);
    GI_LAUNCH_CHECK();
  }
  return GI_OK;
}

}  // namespace

extern "C" {

int gi_loss_tv(gi_ctx* ctx, const float* img, int planes, int H, int W, float tv_weight, float* loss_out, float* grad, float gscale,
               float* scratch) {
  GI_REQUIRE(ctx && img && loss_out && scratch && planes > 0 && H > 1 && W > 1, "loss_tv: bad argument (planes=%d H=%d W=%d)", planes, H, W);
  GI_REQUIRE(((uintptr_t)scratch & 7) == 0, "loss_tv: scratch must be 8-byte aligned");
  const int64_t total = (int64_t)planes * H * W;
  const int nb = rblocks(total);
  const double cnt_w = (double)planes * H * (W - 1), cnt_h = (double)planes * (H - 1) * W;
  hipLaunchKernelGGL(tv_partial_kernel, dim3(nb), dim3(256), 0, ctx->stream, img, total, H, W, (double*)scratch);
  GI_LAUNCH_CHECK();
  hipLaunchKernelGGL(tv_final_kernel, dim3(1), dim3(256), 0, ctx->stream, (const double*)scratch, nb, cnt_w, cnt_h, tv_weight, loss_out);
  GI_LAUNCH_CHECK();
  if (grad) {
    hipLaunchKernelGGL(tv_grad_kernel, dim3(nb), dim3(256), 0, ctx->stream, img, total, H, W, (float)(2.0 * tv_weight * gscale / cnt_w),
                       (float)(2.0 * tv_weight * gscale / cnt_h), grad);
    GI_LAUNCH_CHECK();
  }
  return GI_OK;
}

int gi_loss_cross_entropy(gi_ctx* ctx, const float* logits, const int64_t* labels, int n, int num_classes, int64_t hw,
                          const float* class_weight_host, int ignore_index, float* loss_out, float* grad_logits, float gscale,
                          float* scratch) {
  GI_REQUIRE(ctx && logits && labels && loss_out && scratch && n > 0 && hw > 0, "loss_cross_entropy: bad argument");
  GI_REQUIRE(((uintptr_t)scratch & 7) == 0, "loss_cross_entropy: scratch must be 8-byte aligned");
  CeW cw;
  for (int k = 0; k < CE_MAXK; ++k) cw.w[k] = (k < num_classes) ? (class_weight_host ? class_weight_host[k] : 1.f) : 0.f;
  switch (num_classes) {
    case 2: return ce_run<2>(ctx, logits, labels, n, hw, cw, ignore_index, loss_out, grad_logits, gscale, scratch);
    case 3: return ce_run<3>(ctx, logits, labels, n, hw, cw, ignore_index, loss_out, grad_logits, gscale, scratch);
    case 4: return ce_run<4>(ctx, logits, labels, n, hw, cw, ignore_index, loss_out, grad_logits, gscale, scratch);
    case 8: return ce_run<8>(ctx, logits, labels, n, hw, cw, ignore_index, loss_out, grad_logits, gscale, scratch);
    case 16: return ce_run<16>(ctx, logits, labels, n, hw, cw, ignore_index, loss_out, grad_logits, gscale, scratch);
    default: break;
  }
  gi_set_error("loss_cross_entropy: num_classes=%d (supported: 2, 3, 4, 8, 16)", num_classes);
  return GI_ERR_INVALID;
}

}  // extern "C"
