// Does an out-of-range lane of `buffer_load_dwordx4 ... lds` write zeros into LDS (gfx950)?  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const char* in, unsigned* out, int nbytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 512; i += 64) ((unsigned*)smem)[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, nbytes, 0x00020000);
  // even lanes in range, odd lanes far out of range
  unsigned voff = (threadIdx.x & 1) ? 0x7FFFFFF0u : threadIdx.x * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
  char* d; unsigned* o; unsigned h[512]; unsigned src[256];
  for (int i = 0; i < 256; ++i) src[i] = 0x1000 + i;
  hipMalloc(&d, 1024); hipMalloc(&o, 2048);
  hipMemcpy(d, src, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o, 1024);
  hipMemcpy(h, o, 2048, hipMemcpyDeviceToHost);
  int zero = 0, kept = 0, ok = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
    unsigned v = h[l * 4 + j];
    if (l & 1) { zero += v == 0; kept += v == 0xDEADBEEFu; } else ok += v == src[l * 4 + j];
  }
  printf("in-range words correct %d/128 ; out-of-range words: zero %d/128, untouched %d/128, first %08x\n", ok, zero, kept, h[4]);
  return 0;
}
