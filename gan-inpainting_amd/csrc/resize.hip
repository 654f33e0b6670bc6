// Device input transform: transforms.Resize(size) + transforms.ToTensor() of the reference's
// loaders (train.py:69-72) on decoded 8-bit grey images (lib/data/dataset.py:6-12). torchvision resizes a PIL
// image with Pillow's antialiased bilinear resampler (libImaging/Resample.c): separable, 22-bit fixed-point
// coefficients, horizontal pass first with a uint8 intermediate, then vertical; ToTensor divides by 255.
// The coefficient tables are built on the host in double precision exactly as Pillow does (they depend on the
// geometry only) and live in caller-provided device memory; the two passes are integer kernels, bit-exact.
#include <math.h>
#include <vector>

#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

struct Coeffs {
  int ksize = 0;
  std::vector<int> bounds;   // [out][2] = (xmin, count)
  std::vector<int> kk;       // [out][ksize]
};

// Resample.c precompute_coeffs + normalize_coeffs_8bpc, bilinear filter (support 1), full box
Coeffs precompute(int in_size, int out_size) {
  Coeffs c;
  double scale = (double)in_size / out_size, filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  c.ksize = (int)ceil(support) * 2 + 1;
  c.bounds.assign((size_t)out_size * 2, 0);
  c.kk.assign((size_t)out_size * c.ksize, 0);
  std::vector<double> w(c.ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double t = (x + xmin - center + 0.5) * ss;
      if (t < 0.0) t = -t;
      const double v = t < 1.0 ? 1.0 - t : 0.0;
      w[x] = v;
      ww += v;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) w[x] /= ww;
      c.kk[(size_t)xx * c.ksize + x] = w[x] < 0 ? (int)(-0.5 + w[x] * (1 << PRECISION_BITS)) : (int)(0.5 + w[x] * (1 << PRECISION_BITS));
    }
    c.bounds[(size_t)xx * 2] = xmin;
    c.bounds[(size_t)xx * 2 + 1] = xmax;
  }
  return c;
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// dst[img][y][xo] = clip8(2^21 + sum_k src[img][y][xmin+k] * kk[xo][k])
__global__ void __launch_bounds__(256) resize_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int64_t rows, int in_w,
                                                       int out_w, const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const int64_t total = rows * out_w;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xo = (int)(i % out_w);
    const int64_t r = i / out_w;
    const int xmin = bounds[2 * xo], cnt = bounds[2 * xo + 1];
    const uint8_t* s = src + r * in_w + xmin;
    const int* k = kk + (int64_t)xo * ksize;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int x = 0; x < cnt; ++x) acc += (int)s[x] * k[x];
    dst[i] = (uint8_t)clip8(acc);
  }
}
// out[img][yo][x] = clip8(2^21 + sum_k tmp[img][ymin+k][x] * kk[yo][k]) / 255   (ToTensor)
template <typename OUT>
__global__ void __launch_bounds__(256) resize_v_kernel(const uint8_t* __restrict__ src, OUT* __restrict__ dst, int n, int in_h, int w, int out_h,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const int64_t total = (int64_t)n * out_h * w;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % w);
    const int64_t t = i / w;
    const int yo = (int)(t % out_h);
    const int64_t img = t / out_h;
    int v;
    if (bounds) {
      const int ymin = bounds[2 * yo], cnt = bounds[2 * yo + 1];
      const uint8_t* s = src + (img * in_h + ymin) * (int64_t)w + x;
      const int* k = kk + (int64_t)yo * ksize;
      int acc = 1 << (PRECISION_BITS - 1);
      for (int y = 0; y < cnt; ++y) acc += (int)s[(int64_t)y * w] * k[y];
      v = clip8(acc);
    } else {
      v = src[i];   // height unchanged: Pillow skips the vertical pass
    }
    if constexpr (std::is_same<OUT, float>::value) dst[i] = (float)v / 255.0f;
    else dst[i] = (uint8_t)v;
  }
}

struct Layout {
  int64_t off_bh, off_kh, off_bv, off_kv, total;
  int ks_h, ks_v;
};
Layout table_layout(int in_h, int in_w, int out_h, int out_w, const Coeffs* ch, const Coeffs* cv) {
  Layout L;
  L.ks_h = ch ? ch->ksize : 0;
  L.ks_v = cv ? cv->ksize : 0;
  int64_t o = 0;
  L.off_bh = o; o += (int64_t)out_w * 2 * 4;
  L.off_kh = o; o += (int64_t)out_w * L.ks_h * 4;
  L.off_bv = o; o += (int64_t)out_h * 2 * 4;
  L.off_kv = o; o += (int64_t)out_h * L.ks_v * 4;
  L.total = gi_align_up(o + 16, 256);
  return L;
}
int ksize_of(int in_size, int out_size) {
  double fs = (double)in_size / out_size;
  if (fs < 1.0) fs = 1.0;
  return (int)ceil(fs) * 2 + 1;
}
int nb(int64_t work) {
  int64_t b = (work + 255) / 256;
  if (b > 16384) b = 16384;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

// torchvision _compute_resized_output_size for Resize(int): smaller edge -> size, other int(size*long/short)
int gi_resize_output_size(int in_h, int in_w, int size, int* out_h, int* out_w) {
  GI_REQUIRE(in_h > 0 && in_w > 0 && size > 0 && out_h && out_w, "resize_output_size: bad argument");
  const int shortv = in_w <= in_h ? in_w : in_h, longv = in_w <= in_h ? in_h : in_w;
  const int new_long = (int)((double)size * longv / shortv);
  if (in_w <= in_h) { *out_w = size; *out_h = new_long; }
  else { *out_h = size; *out_w = new_long; }
  return GI_OK;
}

int64_t gi_resize_table_bytes(int in_h, int in_w, int out_h, int out_w) {
  if (in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return -1;
  int64_t o = (int64_t)out_w * 2 * 4 + (int64_t)out_w * ksize_of(in_w, out_w) * 4 + (int64_t)out_h * 2 * 4 +
              (int64_t)out_h * ksize_of(in_h, out_h) * 4;
  return gi_align_up(o + 16, 256);
}

// builds both coefficient tables on the host and copies them to tables_dev (synchronises the context stream once;
// the tables depend on the geometry only: build once per (in_h, in_w, out_h, out_w))
int gi_resize_build_tables(gi_ctx* ctx, int in_h, int in_w, int out_h, int out_w, void* tables_dev) {
  GI_REQUIRE(ctx && tables_dev && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0, "resize_build_tables: bad argument");
  const Coeffs ch = precompute(in_w, out_w), cv = precompute(in_h, out_h);
  const Layout L = table_layout(in_h, in_w, out_h, out_w, &ch, &cv);
  GI_REQUIRE(L.total <= gi_resize_table_bytes(in_h, in_w, out_h, out_w), "resize_build_tables: internal size mismatch");
  char* d = (char*)tables_dev;
  GI_HIP(hipMemcpyAsync(d + L.off_bh, ch.bounds.data(), ch.bounds.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  GI_HIP(hipMemcpyAsync(d + L.off_kh, ch.kk.data(), ch.kk.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  GI_HIP(hipMemcpyAsync(d + L.off_bv, cv.bounds.data(), cv.bounds.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  GI_HIP(hipMemcpyAsync(d + L.off_kv, cv.kk.data(), cv.kk.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  GI_HIP(hipStreamSynchronize(ctx->stream));   // the host vectors die at return
  return GI_OK;
}

// src: n images (in_h x in_w) uint8 on the device; dst_f32: (n,out_h,out_w) float32 in [0,1] (may be NULL);
// dst_u8 (may be NULL): the resized bytes. tmp: n*in_h*out_w bytes of device scratch.
int gi_resize_to_tensor(gi_ctx* ctx, const void* tables_dev, const uint8_t* src, int n, int in_h, int in_w, int out_h, int out_w,
                        float* dst_f32, uint8_t* dst_u8, uint8_t* tmp) {
  GI_REQUIRE(ctx && tables_dev && src && tmp && (dst_f32 || dst_u8) && n > 0, "resize_to_tensor: bad argument");
  Layout L = table_layout(in_h, in_w, out_h, out_w, nullptr, nullptr);
  L.ks_h = ksize_of(in_w, out_w);
  L.ks_v = ksize_of(in_h, out_h);
  L.off_kh = L.off_bh + (int64_t)out_w * 2 * 4;
  L.off_bv = L.off_kh + (int64_t)out_w * L.ks_h * 4;
  L.off_kv = L.off_bv + (int64_t)out_h * 2 * 4;
  const char* d = (const char*)tables_dev;
  const uint8_t* hsrc = src;
  if (out_w != in_w) {   // Pillow runs a pass only when that dimension changes
    hipLaunchKernelGGL(resize_h_kernel, dim3(nb((int64_t)n * in_h * out_w)), dim3(256), 0, ctx->stream, src, tmp, (int64_t)n * in_h, in_w, out_w,
                       (const int*)(d + L.off_bh), (const int*)(d + L.off_kh), L.ks_h);
    GI_LAUNCH_CHECK();
    hsrc = tmp;
  }
  const int* bv = out_h != in_h ? (const int*)(d + L.off_bv) : nullptr;
  const int* kv = (const int*)(d + L.off_kv);
  const int grid = nb((int64_t)n * out_h * out_w);
  if (dst_f32) {
    hipLaunchKernelGGL(resize_v_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, hsrc, dst_f32, n, in_h, out_w, out_h, bv, kv, L.ks_v);
    GI_LAUNCH_CHECK();
  }
  if (dst_u8) {
    hipLaunchKernelGGL(resize_v_kernel<uint8_t>, dim3(grid), dim3(256), 0, ctx->stream, hsrc, dst_u8, n, in_h, out_w, out_h, bv, kv, L.ks_v);
    GI_LAUNCH_CHECK();
  }
  return GI_OK;
}

}  // extern "C"
