// Streaming-pass microbenchmark for the elementwise kernels (BatchNorm apply / backward apply shapes): R read streams and one
// write stream of 16-byte chunks, U chunks per thread in flight, grid = B blocks per CU, with and without a stride pattern
// (reads from one half of a 2c-channel concat buffer). Prints GB/s of total traffic.
// build: hipcc --offload-arch=gfx950 -O3 -o stream_bw tools/micro/stream_bw.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

template <int R, int U, bool NT>
__global__ void __launch_bounds__(256) pass_kernel(const char* a, const char* b, const char* c, char* out, long total, int cpp, int ldf) {
  // chunk i lives at (i / cpp) * cpp * ldf + (i % cpp) chunks: ldf = 1 dense, 2 = every other cpp-chunk group (half of a concat row)
  const long stride = (long)gridDim.x * 256;
  auto off = [&](long i) -> long { return ((i / cpp) * cpp * ldf + (i % cpp)) * 16; };
  for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += U * stride) {
    u4_t ra[U], rb[U], rc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < total) {
        if (NT) {
          ra[u] = __builtin_nontemporal_load((const u4_t*)(a + off(i)));
          if (R > 1) rb[u] = __builtin_nontemporal_load((const u4_t*)(b + i * 16));
          if (R > 2) rc[u] = __builtin_nontemporal_load((const u4_t*)(c + i * 16));
        } else {
          ra[u] = *(const u4_t*)(a + off(i));
          if (R > 1) rb[u] = *(const u4_t*)(b + i * 16);
          if (R > 2) rc[u] = *(const u4_t*)(c + i * 16);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < total) {
        h8_t x = __builtin_bit_cast(h8_t, ra[u]);
        if (R > 1) { h8_t y = __builtin_bit_cast(h8_t, rb[u]); x = x * (_Float16)0.5f + y; }
        if (R > 2) { h8_t z = __builtin_bit_cast(h8_t, rc[u]); x = x + z * (_Float16)0.25f; }
        if (NT) __builtin_nontemporal_store(__builtin_bit_cast(u4_t, x), (u4_t*)(out + i * 16));
        else *(u4_t*)(out + i * 16) = __builtin_bit_cast(u4_t, x);
      }
    }
  }
}

// write-only stream (the single-channel gather layers write 8x what they read): every thread stores U 16-byte chunks per trip
template <int U>
__global__ void __launch_bounds__(256) fill_kernel(char* out, long total, unsigned v) {
  const long stride = (long)gridDim.x * 256;
  const u4_t x = {v, v + 1, v + 2, v + 3};
  for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += U * stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < total) *(u4_t*)(out + i * 16) = x;
    }
  }
}

// FETCH_SIZE calibration for 64-byte sector reads: every group of four lanes reads the FIRST (or second) 64 bytes of a 128-byte line and
// skips the other half; the line's other half is never touched by this launch. If the memory side fetched whole 128-byte lines the pass
// would move 2x its useful bytes and take as long as reading everything; if it fetches 64-byte sectors it takes half as long.
__global__ void __launch_bounds__(256) half_line_kernel(const char* a, unsigned* sink, long lines, int half) {
  const long stride = (long)gridDim.x * 64;       // lines per trip (a thread quad per line)
  unsigned acc = 0;
  for (long l = (long)blockIdx.x * 64 + (threadIdx.x >> 2); l < lines; l += stride) {
    const u4_t v = *(const u4_t*)(a + l * 128 + half * 64 + (threadIdx.x & 3) * 16);
    acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int R, int U, bool NT>
float run(const char* a, const char* b, const char* c, char* out, long total, int cpp, int ldf, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pass_kernel<R, U, NT>), dim3(blocks), dim3(256), 0, 0, a, b, c, out, total, cpp, ldf);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((pass_kernel<R, U, NT>), dim3(blocks), dim3(256), 0, 0, a, b, c, out, total, cpp, ldf);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main(int argc, char** argv) {
  const long mb = argc > 1 ? atol(argv[1]) : 64;     // bytes per stream in MiB
  const long bytes = mb << 20, total = bytes / 16;
  char *a, *b, *c, *o;
  hipMalloc(&a, 2 * bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&o, bytes);
  hipMemset(a, 1, 2 * bytes); hipMemset(b, 2, bytes); hipMemset(c, 3, bytes); hipMemset(o, 0, bytes);
  // a spacer buffer the passes alternate with, so that nothing survives in the 256 MiB Infinity Cache between iterations
  printf("stream %ld MiB per tensor; columns: R reads + 1 write, U chunks in flight, blocks/CU, layout, policy -> us, GB/s total\n", mb);
  {   // half-line reads over a buffer far larger than the caches (1 GiB): useful bytes = 512 MiB per pass
    const long big = 1l << 30;
    char* g; unsigned* sink;
    hipMalloc(&g, big); hipMalloc(&sink, 64);
    hipMemset(g, 1, big);
    for (int half = 0; half < 2; ++half) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(half_line_kernel, dim3(256 * 16), dim3(256), 0, 0, g, sink, big / 128, half);
      hipEventRecord(e0);
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(half_line_kernel, dim3(256 * 16), dim3(256), 0, 0, g, sink, big / 128, (half + i) & 1);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 5;
      printf("half-line reads (64 B of every 128-B line, alternating halves per pass), 1 GiB buffer: %7.1f us  %7.0f GB/s of useful bytes\n", ms * 1e3, 0.5 * big / ms / 1e6);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((pass_kernel<1, 4, false>), dim3(256 * 8), dim3(256), 0, 0, g, g, g, g + (big / 2), big / 2 / 16, 16, 1);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((pass_kernel<1, 4, false>), dim3(256 * 8), dim3(256), 0, 0, g, g, g, g + (big / 2), big / 2 / 16, 16, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("full-line copy 512 MiB -> 512 MiB of the same buffer: %7.1f us  %7.0f GB/s total\n", ms * 1e3, (double)big / ms / 1e6);
    hipFree(g); hipFree(sink);
  }
  const int cpp = 16;   // 128 channels fp16
  for (int bpc : {2, 8, 32}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fill_kernel<2>, dim3(256 * bpc), dim3(256), 0, 0, a, 2 * total, 7u);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fill_kernel<2>, dim3(256 * bpc), dim3(256), 0, 0, a, 2 * total, 7u + i);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    printf("write only, %ld MiB, b/CU=%2d : %7.1f us %7.0f GB/s\n", 2 * mb, bpc, ms * 1e3, 2.0 * bytes / ms / 1e6);
  }
  for (int ldf = 1; ldf <= 2; ++ldf)
    for (int bpc : {2, 4, 8, 16}) {
      const int blocks = 256 * bpc;
#define ROW(R_, U_, NT_) do { float ms = run<R_, U_, NT_>(a, b, c, o, total, cpp, ldf, blocks, 20); \
      printf("R=%d U=%d b/CU=%2d ldf=%d %s : %7.1f us %7.0f GB/s\n", R_, U_, bpc, ldf, NT_ ? "nt" : "  ", ms * 1e3, (R_ + 1) * (double)bytes / ms / 1e6); } while (0)
      ROW(1, 1, false); ROW(1, 2, false); ROW(1, 4, false); ROW(1, 8, false);
      ROW(2, 1, false); ROW(2, 2, false); ROW(2, 4, false);
      ROW(3, 1, false); ROW(3, 2, false); ROW(3, 4, false);
      ROW(1, 4, true); ROW(2, 2, true); ROW(2, 4, true);
    }
  return 0;
}
