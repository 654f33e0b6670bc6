// Evaluation-pass reductions: the reconstruction metrics of the reference's
// lib/models/evaluate.py:127-158 (composite + global/local L1 and RMSE per batch, accumulated over a
// loader) and the segmentation precision / recall / IoU of evaluate.py:179-224.
// Both are single-pass HBM-bound reductions; every sum is finished in a fixed order (deterministic).
#include "common.h"

namespace {

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// out = gen*m + ground*(1-m)            (evaluate.py:134-138; m already ceil-ed / flipped by gi_mask_apply)
// partial[b] = { sum|d|, sum d^2, sum|d*m|, sum (d*m)^2, count(m != 0) },  d = ground - out
__global__ void __launch_bounds__(256) eval_recon_partial_kernel(const float* __restrict__ ground, const float* __restrict__ gen,
                                                                 const float* __restrict__ m, int64_t count,
                                                                 float* __restrict__ out, double* __restrict__ partial) {
  __shared__ double sh[4];
  double s1 = 0.0, s2 = 0.0, l1 = 0.0, l2 = 0.0, cnt = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const float g = ground[i], mm = m[i];
    const float masked = g * (1.f - mm);
    const float o = __fadd_rn(__fmul_rn(gen[i], mm), masked);   // mul, then add: two roundings like `out * m + masked`
    if (out) out[i] = o;
    const float d = g - o;
    const float dl = __fsub_rn(__fmul_rn(o, mm), __fmul_rn(g, mm));   // loss(y*mask, yhat*mask): y = out, yhat = ground
    s1 += (double)fabsf(d);
    s2 += (double)d * d;
    l1 += (double)fabsf(dl);
    l2 += (double)dl * dl;
    cnt += (mm != 0.f) ? 1.0 : 0.0;
  }
  s1 = block_sum_d(s1, sh);
  s2 = block_sum_d(s2, sh);
  l1 = block_sum_d(l1, sh);
  l2 = block_sum_d(l2, sh);
  cnt = block_sum_d(cnt, sh);
  if (threadIdx.x == 0) {
    double* p = partial + (int64_t)blockIdx.x * 5;
    p[0] = s1; p[1] = s2; p[2] = l1; p[3] = l2; p[4] = cnt;
  }
}

// acc[0..3] += { rmse_global, l1_global, rmse_local, l1_local } of this batch ; acc[4] += 1 (batches)
__global__ void __launch_bounds__(256) eval_recon_final_kernel(const double* __restrict__ partial, int nblocks, double count,
                                                               float eps, float* __restrict__ acc, float* __restrict__ batch_out) {
  __shared__ double sh[4];
  double v[5] = {0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < nblocks; i += 256)
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] += partial[(int64_t)i * 5 + k];
#pragma unroll
  for (int k = 0; k < 5; ++k) v[k] = block_sum_d(v[k], sh);
  if (threadIdx.x == 0) {
    const float r[4] = {(float)sqrt(v[1] / count + (double)eps), (float)(v[0] / count),
                        (float)sqrt(v[3] / v[4] + (double)eps), (float)(v[2] / v[4])};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (acc) acc[k] += r[k];
      if (batch_out) batch_out[k] = r[k];
    }
    if (acc) acc[4] += 1.f;
  }
}

constexpr int SEG_MAXU = 16;
struct SegLabels {
  int u[SEG_MAXU];
};

// counts[n][u] = { |p & g|, |p|, |g| } with p = (argmax_k logits == u), g = (label == u)  (evaluate.py:193-207)
template <int NU>
__global__ void __launch_bounds__(256) seg_count_kernel(const int64_t* __restrict__ labels, const float* __restrict__ logits,
                                                        int K, int64_t hw, SegLabels L, int* __restrict__ counts) {
  __shared__ int sh[4][NU * 3];
  const int n = blockIdx.y;
  const int64_t* lab = labels + (int64_t)n * hw;
  const float* lg = logits + (int64_t)n * K * hw;
  int c[NU][3];
#pragma unroll
  for (int u = 0; u < NU; ++u) c[u][0] = c[u][1] = c[u][2] = 0;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < hw; p += (int64_t)gridDim.x * 256) {
    float best = lg[p];
    int arg = 0;
    for (int k = 1; k < K; ++k) {   // torch.argmax: first index of the maximum; NaN counts as the maximum
      const float v = lg[(int64_t)k * hw + p];
      if (v > best || (v != v && best == best)) { best = v; arg = k; }
    }
    const int64_t l = lab[p];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int pu = arg == L.u[u];
      const int gu = (l == (int64_t)L.u[u]) || (l == -1);   // the reference marks matches with -1 in place: a label of -1 matches every class
      c[u][0] += pu & gu;
      c[u][1] += pu;
      c[u][2] += gu;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int v = c[u][j];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) sh[wave][u * 3 + j] = v;
    }
  __syncthreads();
  if (threadIdx.x < NU * 3) {
    const int t = threadIdx.x;
    atomicAdd(counts + ((int64_t)n * NU) * 3 + t, sh[0][t] + sh[1][t] + sh[2][t] + sh[3][t]);   // integer: order-free
  }
}

// per_class[u] = { mean_n precision, mean_n recall, mean_n iou } ; across[j] = sum_u per_class[u][j] / nu
__global__ void __launch_bounds__(64) seg_final_kernel(const int* __restrict__ counts, int n, int nu, int stride,
                                                       float* __restrict__ per_class, float* __restrict__ across) {
  if (threadIdx.x != 0) return;
  const float eps = 1e-32f;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int u = 0; u < nu; ++u) {
    float sp = 0.f, sr = 0.f, si = 0.f;
    for (int s = 0; s < n; ++s) {
      const int* c = counts + ((int64_t)s * stride + u) * 3;
      const float inter = (float)c[0], pc = (float)c[1], gc = (float)c[2], uni = (float)(c[1] + c[2] - c[0]);
      si += inter / (uni + eps);
      sp += inter / (pc + eps);
      sr += inter / (gc + eps);
    }
    const float mp = sp / (float)n, mr = sr / (float)n, mi = si / (float)n;
    per_class[u * 3 + 0] = mp;
    per_class[u * 3 + 1] = mr;
    per_class[u * 3 + 2] = mi;
    acc[0] += mp;
    acc[1] += mr;
    acc[2] += mi;
  }
  for (int j = 0; j < 3; ++j) across[j] = acc[j] / (float)nu;
}

int eval_blocks(int64_t count) {
  int64_t b = (count + 256 * 8 - 1) / (256 * 8);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" {

int64_t gi_eval_recon_scratch_floats(int64_t count) { return count > 0 ? (int64_t)eval_blocks(count) * 10 : -1; }

int gi_eval_recon(gi_ctx* ctx, const float* ground, const float* gen, const float* mask_c, int64_t count, float eps,
                  float* inpainted_out, float* acc, float* batch_out, float* scratch) {
  GI_REQUIRE(ctx && ground && gen && mask_c && scratch && count > 0, "eval_recon: bad argument");
  GI_REQUIRE(((uintptr_t)scratch & 7) == 0, "eval_recon: scratch must be 8-byte aligned");
  const int nb = eval_blocks(count);
  hipLaunchKernelGGL(eval_recon_partial_kernel, dim3(nb), dim3(256), 0, ctx->stream, ground, gen, mask_c, count, inpainted_out,
                     (double*)scratch);
  GI_LAUNCH_CHECK();
  hipLaunchKernelGGL(eval_recon_final_kernel, dim3(1), dim3(256), 0, ctx->stream, (const double*)scratch, nb, (double)count, eps, acc,
                     batch_out);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int gi_seg_metrics(gi_ctx* ctx, const int64_t* labels, const float* logits, int n, int num_classes, int64_t hw,
                   const int* unique_labels_host, int nu, float* per_class, float* across, int* scratch_counts) {
  GI_REQUIRE(ctx && labels && logits && unique_labels_host && per_class && across && scratch_counts, "seg_metrics: null argument");
  GI_REQUIRE(n > 0 && n <= 65535 && num_classes > 0 && hw > 0 && nu > 0 && nu <= SEG_MAXU, "seg_metrics: n=%d classes=%d nu=%d (<=%d)", n,
             num_classes, nu, SEG_MAXU);
  SegLabels L;
  for (int u = 0; u < SEG_MAXU; ++u) L.u[u] = u < nu ? unique_labels_host[u] : INT32_MIN;
  const int NUc = nu <= 4 ? 4 : (nu <= 8 ? 8 : 16);
  GI_HIP(hipMemsetAsync(scratch_counts, 0, sizeof(int) * (size_t)n * NUc * 3, ctx->stream));
  int64_t bx = (hw + 256 * 4 - 1) / (256 * 4);
  if (bx > 64) bx = 64;
  const dim3 grid((unsigned)bx, n);
  if (NUc == 4)
    hipLaunchKernelGGL(seg_count_kernel<4>, grid, dim3(256), 0, ctx->stream, labels, logits, num_classes, hw, L, scratch_counts);
  else if (NUc == 8)
    hipLaunchKernelGGL(seg_count_kernel<8>, grid, dim3(256), 0, ctx->stream, labels, logits, num_classes, hw, L, scratch_counts);
  else
    hipLaunchKernelGGL(seg_count_kernel<16>, grid, dim3(256), 0, ctx->stream, labels, logits, num_classes, hw, L, scratch_counts);
  GI_LAUNCH_CHECK();
  // the padded classes (label INT32_MIN) count nothing; the finish reads the first `nu` of each NUc group
  hipLaunchKernelGGL(seg_final_kernel, dim3(1), dim3(64), 0, ctx->stream, scratch_counts, n, nu, NUc, per_class, across);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

}  // extern "C"
