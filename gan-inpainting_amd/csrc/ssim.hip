// SSIM metric (lib/pytorch_ssim/__init__.py:20-40, 68-76 of the reference): five depthwise Gaussian
// filterings (x, y, x^2, y^2, xy; window ws x ws = outer product of a normalised 1-D Gaussian, zero
// padding ws/2), the SSIM map, and its mean (over everything, or per sample).
//
// One fused kernel per (tile, plane): both image tiles + halo go to LDS once, the window is applied
// separably (horizontal pass LDS->LDS on the five quantities, vertical pass LDS->registers), the map
// never exists in memory; per-tile sums land in `scratch` and a fixed-order fp64 finish makes the
// result deterministic. HBM-bound by construction: 8 bytes read per pixel, nothing written.
#include "common.h"

namespace {

struct SsimWin {
  float g[32];
};

template <int TH, int TW>
__global__ void __launch_bounds__(256) ssim_tile_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H,
                                                        int W, int ws, SsimWin win, float* __restrict__ partial) {
  extern __shared__ float lds[];
  const int r = ws >> 1;
  const int IW = TW + 2 * r, IH = TH + 2 * r;
  float* sx = lds;
  float* sy = sx + IH * IW;
  float* hq = sy + IH * IW;  // [5][IH][TW]
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const int64_t plane = blockIdx.z;
  const float* p1 = img1 + plane * H * W;
  const float* p2 = img2 + plane * H * W;

  for (int i = tid; i < IH * IW; i += 256) {
    const int iy = i / IW, ix = i - iy * IW;
    const int gy = y0 - r + iy, gx = x0 - r + ix;
    float a = 0.f, b = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      a = p1[(int64_t)gy * W + gx];
      b = p2[(int64_t)gy * W + gx];
    }
    sx[i] = a;
    sy[i] = b;
  }
  __syncthreads();

  const int HS = IH * TW;
  for (int i = tid; i < HS; i += 256) {
    const int iy = i / TW, ix = i - iy * TW;
    const float* rx = sx + iy * IW + ix;
    const float* ry = sy + iy * IW + ix;
    float m1 = 0.f, m2 = 0.f, q11 = 0.f, q22 = 0.f, q12 = 0.f;
    for (int k = 0; k < ws; ++k) {
      const float w = win.g[k];
      const float a = rx[k], b = ry[k];
      m1 = fmaf(w, a, m1);
      m2 = fmaf(w, b, m2);
      q11 = fmaf(w, a * a, q11);
      q22 = fmaf(w, b * b, q22);
      q12 = fmaf(w, a * b, q12);
    }
    hq[i] = m1;
    hq[HS + i] = m2;
    hq[2 * HS + i] = q11;
    hq[3 * HS + i] = q22;
    hq[4 * HS + i] = q12;
  }
  __syncthreads();

  constexpr int RPT = TH * TW / 256;  // output rows per thread (a vertical strip)
  const int col = tid % TW, rg = tid / TW;
  float acc = 0.f;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  if (x0 + col < W) {
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
      const int oy = rg * RPT + rr;
      if (y0 + oy >= H) break;
      const float* base = hq + oy * TW + col;
      float m1 = 0.f, m2 = 0.f, q11 = 0.f, q22 = 0.f, q12 = 0.f;
      for (int k = 0; k < ws; ++k) {
        const float w = win.g[k];
        const float* b = base + k * TW;
        m1 = fmaf(w, b[0], m1);
        m2 = fmaf(w, b[HS], m2);
        q11 = fmaf(w, b[2 * HS], q11);
        q22 = fmaf(w, b[3 * HS], q22);
        q12 = fmaf(w, b[4 * HS], q12);
      }
      const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
      const float s1 = q11 - m11, s2 = q22 - m22, s12 = q12 - m12;
      acc += ((2.f * m12 + C1) * (2.f * s12 + C2)) / ((m11 + m22 + C1) * (s1 + s2 + C2));
    }
  }
  // block sum in a fixed order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  __syncthreads();
  if ((tid & 63) == 0) lds[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    const int64_t tiles = (int64_t)gridDim.x * gridDim.y;
    partial[plane * tiles + (int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
  }
}

// ---- window 11 (the reference's default and only call-site value), 16x64 output tile -------------
// 320 threads. Sliding windows in registers, so each LDS value is read once per pass instead of 11x:
//   pass V: a thread owns one column and 4 output rows (74 x 4 = 296 threads): reads 14 rows of x and
//           y, forms x^2, y^2, xy, keeps 4x5 running sums; results go to LDS as (mu1,mu2) / (E x^2,
//           E y^2) float2 planes and an E xy plane, column c at phys(c);
//   pass H: a thread owns 4 consecutive output columns of one row (256 threads): 14 reads per plane.
// (x,y) and (x^2,y^2) travel as float2 so the filtering runs on v_pk_fma_f32.
// phys(c) = (c & 3) * 20 + (c >> 2) puts the 4-strided columns that the lanes of pass H read together
// at consecutive addresses; row pitch 80 elements keeps the rows of a wave on distinct banks.
typedef float f2_t __attribute__((ext_vector_type(2)));
constexpr int S11_TH = 16, S11_TW = 64, S11_R = 5, S11_IW = S11_TW + 2 * S11_R, S11_IH = S11_TH + 2 * S11_R;
constexpr int S11_VP = 80;       // V row pitch (elements)
constexpr int S11_NT = 320;      // threads
constexpr int S11_NLD = (S11_IH * S11_IW + S11_NT - 1) / S11_NT;

__device__ __forceinline__ int s11_phys(int c) { return (c & 3) * 20 + (c >> 2); }

__global__ void __launch_bounds__(S11_NT) ssim11_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H,
                                                        int W, SsimWin win, float* __restrict__ partial) {
  __shared__ f2_t sxy[S11_IH * S11_IW];
  __shared__ f2_t Vm[S11_TH * S11_VP];
  __shared__ f2_t Vq[S11_TH * S11_VP];
  __shared__ float Vx[S11_TH * S11_VP];
  __shared__ float red[S11_NT / 64];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * S11_TW, y0 = blockIdx.y * S11_TH;
  const int64_t plane = blockIdx.z;
  const float* p1 = img1 + plane * H * W;
  const float* p2 = img2 + plane * H * W;
  float g[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) g[k] = win.g[k];

  {  // tile + halo of both images: all loads in flight before the first LDS write
    float va[S11_NLD], vb[S11_NLD];
#pragma unroll
    for (int t = 0; t < S11_NLD; ++t) {
      const int i = tid + t * S11_NT;
      const int iy = i / S11_IW, ix = i - iy * S11_IW;
      const int gy = y0 - S11_R + iy, gx = x0 - S11_R + ix;
      const bool ok = i < S11_IH * S11_IW && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const int64_t idx = ok ? (int64_t)gy * W + gx : 0;
      const float a = p1[idx], b = p2[idx];
      va[t] = ok ? a : 0.f;
      vb[t] = ok ? b : 0.f;
    }
#pragma unroll
    for (int t = 0; t < S11_NLD; ++t) {
      const int i = tid + t * S11_NT;
      if (i < S11_IH * S11_IW) sxy[i] = f2_t{va[t], vb[t]};
    }
  }
  __syncthreads();

  if (tid < 4 * S11_IW) {  // pass V: (column, quarter of the rows)
    const int col = tid % S11_IW, qr = tid / S11_IW;
    f2_t am[4], aq[4];
    float ax[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      am[o] = f2_t{0.f, 0.f};
      aq[o] = f2_t{0.f, 0.f};
      ax[o] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      const f2_t ab = sxy[(qr * 4 + j) * S11_IW + col];
      const f2_t sq = ab * ab;
      const float x = ab.x * ab.y;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int k = j - o;
        if (k >= 0 && k < 11) {
          const f2_t gk = f2_t{g[k], g[k]};
          am[o] = __builtin_elementwise_fma(gk, ab, am[o]);
          aq[o] = __builtin_elementwise_fma(gk, sq, aq[o]);
          ax[o] = fmaf(g[k], x, ax[o]);
        }
      }
    }
    const int pc = s11_phys(col);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      Vm[(qr * 4 + o) * S11_VP + pc] = am[o];
      Vq[(qr * 4 + o) * S11_VP + pc] = aq[o];
      Vx[(qr * 4 + o) * S11_VP + pc] = ax[o];
    }
  }
  __syncthreads();

  float sum = 0.f;
  if (tid < 256) {  // pass H: (row, run of 4 columns)
    const int row = tid >> 4, run = tid & 15;
    f2_t am[4], aq[4];
    float ax[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      am[o] = f2_t{0.f, 0.f};
      aq[o] = f2_t{0.f, 0.f};
      ax[o] = 0.f;
    }
    const int base = row * S11_VP + run;
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      const int off = base + (j & 3) * 20 + (j >> 2);
      const f2_t m = Vm[off], q = Vq[off];
      const float x = Vx[off];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int k = j - o;
        if (k >= 0 && k < 11) {
          const f2_t gk = f2_t{g[k], g[k]};
          am[o] = __builtin_elementwise_fma(gk, m, am[o]);
          aq[o] = __builtin_elementwise_fma(gk, q, aq[o]);
          ax[o] = fmaf(g[k], x, ax[o]);
        }
      }
    }
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    if (y0 + row < H) {
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        if (x0 + run * 4 + o < W) {
          const float m1 = am[o].x, m2 = am[o].y;
          const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
          const float s1 = aq[o].x - m11, s2 = aq[o].y - m22, s12 = ax[o] - m12;
          sum += ((2.f * m12 + C1) * (2.f * s12 + C2)) / ((m11 + m22 + C1) * (s1 + s2 + C2));
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  if (tid == 0) {
    const int64_t tiles = (int64_t)gridDim.x * gridDim.y;
    partial[plane * tiles + (int64_t)blockIdx.y * gridDim.x + blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) + red[4];
  }
}

// per_sample[s] = sum(partial[s*per .. (s+1)*per)) / pixels   (one block per sample, fp64, fixed order)
__global__ void __launch_bounds__(256) ssim_finish_kernel(const float* __restrict__ partial, int64_t per, double inv_pixels,
                                                          float* __restrict__ per_sample, double* __restrict__ sums) {
  __shared__ double red[256];
  const int s = blockIdx.x, tid = threadIdx.x;
  double a = 0.0;
  for (int64_t i = tid; i < per; i += 256) a += (double)partial[(int64_t)s * per + i];
  red[tid] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    sums[s] = red[0];
    if (per_sample) per_sample[s] = (float)(red[0] * inv_pixels);
  }
}

__global__ void __launch_bounds__(64) ssim_mean_kernel(const double* __restrict__ sums, int n, double inv_total, float* __restrict__ mean_out) {
  if (threadIdx.x == 0) {
    double a = 0.0;
    for (int i = 0; i < n; ++i) a += sums[i];
    mean_out[0] = (float)(a * inv_total);
  }
}

struct SsimPlan {
  int th, tw, tx, ty;
  int64_t tiles, partial_floats, total_floats;
  size_t lds;
};

bool ssim_plan(int n, int c, int H, int W, int ws, SsimPlan* p) {
  if (n <= 0 || c <= 0 || H <= 0 || W <= 0 || ws < 1 || ws > 31 || (ws & 1) == 0) return false;
  if (ws <= 11) { p->th = 16; p->tw = 64; } else { p->th = 8; p->tw = 32; }
  p->tx = (W + p->tw - 1) / p->tw;
  p->ty = (H + p->th - 1) / p->th;
  p->tiles = (int64_t)p->tx * p->ty;
  p->partial_floats = gi_align_up((int64_t)n * c * p->tiles, 2);
  p->total_floats = p->partial_floats + 2 * (int64_t)n;  // + n doubles
  const int r = ws / 2, IW = p->tw + 2 * r, IH = p->th + 2 * r;
  p->lds = (size_t)(2 * IH * IW + 5 * IH * p->tw) * sizeof(float);
  return (int64_t)n * c <= 65535 && p->ty <= 65535;
}

}  // namespace

extern "C" {

int64_t gi_ssim_scratch_floats(int n, int c, int H, int W, int window_size) {
  SsimPlan p;
  if (!ssim_plan(n, c, H, W, window_size, &p)) return -1;
  return p.total_floats;
}

int gi_ssim(gi_ctx* ctx, const float* img1, const float* img2, int n, int c, int H, int W, int window_size,
            const float* window_host, float* per_sample, float* mean_out, float* scratch) {
  SsimPlan p;
  GI_REQUIRE(ctx && img1 && img2 && scratch, "ssim: null argument");
  GI_REQUIRE(ssim_plan(n, c, H, W, window_size, &p), "ssim: n=%d c=%d H=%d W=%d window_size=%d (odd, <=31; n*c<=65535)", n, c,
             H, W, window_size);
  GI_REQUIRE(((uintptr_t)scratch & 7) == 0, "ssim: scratch must be 8-byte aligned");
  SsimWin win;
  for (int k = 0; k < 32; ++k) win.g[k] = 0.f;
  if (window_host) {
    for (int k = 0; k < window_size; ++k) win.g[k] = window_host[k];
  } else {  // gaussian(window_size, 1.5): fp64 exp -> fp32, fp32 sum, fp32 divide (__init__.py:10-12)
    float sum = 0.f;
    for (int k = 0; k < window_size; ++k) {
      const int d = k - window_size / 2;
      win.g[k] = (float)exp(-(double)(d * d) / (2.0 * 1.5 * 1.5));
      sum += win.g[k];
    }
    for (int k = 0; k < window_size; ++k) win.g[k] /= sum;
  }
  double* sums = (double*)(scratch + p.partial_floats);
  const dim3 grid(p.tx, p.ty, n * c);
  if (window_size == 11)
    hipLaunchKernelGGL(ssim11_kernel, grid, dim3(S11_NT), 0, ctx->stream, img1, img2, H, W, win, scratch);
  else if (p.th == 16)
    hipLaunchKernelGGL((ssim_tile_kernel<16, 64>), grid, dim3(256), p.lds, ctx->stream, img1, img2, H, W, window_size, win, scratch);
  else
    hipLaunchKernelGGL((ssim_tile_kernel<8, 32>), grid, dim3(256), p.lds, ctx->stream, img1, img2, H, W, window_size, win, scratch);
  GI_LAUNCH_CHECK();
  const double pixels = (double)c * H * W;
  hipLaunchKernelGGL(ssim_finish_kernel, dim3(n), dim3(256), 0, ctx->stream, scratch, (int64_t)c * p.tiles, 1.0 / pixels, per_sample,
                     sums);
  GI_LAUNCH_CHECK();
  if (mean_out) {
    hipLaunchKernelGGL(ssim_mean_kernel, dim3(1), dim3(64), 0, ctx->stream, sums, n, 1.0 / (pixels * n), mean_out);
    GI_LAUNCH_CHECK();
  }
  return GI_OK;
}

}  // extern "C"
