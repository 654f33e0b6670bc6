// BatchNorm from the exact accumulators (stat_acc.h): the pieces shared by the normalisation pass (elementwise.hip), the kernels that
// apply the affine map themselves (c1.hip) and the GEMM that normalises its own output (igemm7.hip, IgemmFold).
#pragma once
#include "common.h"
#include "stat_acc.h"

// Called by one thread per channel; `write` (one workgroup per channel and launch) also stores the vectors the backward needs and
// moves the running statistics, population by population in the order separate calls would.
struct BnAccP {
  const unsigned long long* acc;   // accumulator block (stat_acc.h)
  int reps;
  const float* gamma; const float* beta;
  float* rmean; float* rvar;
  float* scale; float* shift; float* smean; float* sinv;   // [groups] x out_stride floats apart
  double count;                    // pixels per population
  float momentum, eps;
  int groups, out_stride;
  unsigned long long* zero_next;   // the layer's OTHER accumulator region: cleared here for its next use (ping-pong), so
  int zero_words;                  // no memset launch is needed; nothing else touches it while this kernel runs
};
static inline void gi_fill_acc_params(BnAccP& fa, const BnAccArgs& b) {
  fa.acc = b.acc; fa.reps = b.reps > 0 ? b.reps : 1; fa.gamma = b.gamma; fa.beta = b.beta; fa.rmean = b.running_mean; fa.rvar = b.running_var;
  fa.scale = b.scale; fa.shift = b.shift; fa.smean = b.save_mean; fa.sinv = b.save_invstd;
  fa.count = (double)b.count; fa.momentum = b.momentum; fa.eps = b.eps; fa.groups = b.groups; fa.out_stride = b.out_stride;
  fa.zero_next = b.zero_next; fa.zero_words = b.zero_words;
}
__device__ __forceinline__ void zero_words64(unsigned long long* p, int n) {
  for (int i = threadIdx.x; i < n; i += 256) p[i] = 0ull;
}
// COHERENT: the accumulators were added to by OTHER workgroups of the running launch (igemm7's folded normalisation): every word is
// read with an agent-scope relaxed atomic load (global_load_dwordx2 sc1), the form the guide measures for counters that receive
// agent-scope atomic adds (MI355X_MICROARCH.md, hand-off table, third row). Same arithmetic either way.
template <bool COHERENT = false>
__device__ __forceinline__ void bn_from_acc(const BnAccP& a, int c, int ch, int j, bool write, float& sc_out, float& sh_out) {
  // the parameters are requested before the accumulator words are waited for (one memory round trip, not two)
  const float gam = a.gamma[ch], bet = a.beta[ch];
  float rm = 0.f, rv = 0.f;
  if (write) { rm = a.rmean[ch]; rv = a.rvar[ch]; }
  double t0, t1;
  gi_stat_read2<COHERENT>(a.acc, c, a.reps, j, ch, t0, t1);
  const double m = t0 / a.count;
  double v = t1 / a.count - m * m;
  if (v < 0.0) v = 0.0;
  const float mean = (float)m, var = (float)v;
  const float inv = 1.0f / sqrtf(var + a.eps);
  const float sc = gam * inv;
  sc_out = sc;
  sh_out = bet - mean * sc;
  if (write) {
    const float unbiased = a.count > 1.0 ? (float)(v * a.count / (a.count - 1.0)) : var;
    a.rmean[ch] = (1.f - a.momentum) * rm + a.momentum * mean;
    a.rvar[ch] = (1.f - a.momentum) * rv + a.momentum * unbiased;
    const int o = j * a.out_stride + ch;
    a.scale[o] = sc;
    a.shift[o] = sh_out;
    a.smean[o] = mean;
    a.sinv[o] = inv;
  }
}

// dropout keep-mask of element i under `seed` (splitmix64 of a counter), keep with probability 1 - p
__device__ __forceinline__ uint8_t dropout_keep(uint64_t seed, int64_t i, uint32_t thresh) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return ((uint32_t)(z >> 32) >= thresh) ? 1 : 0;
}
static inline uint32_t gi_dropout_thresh(float p) {   // keep when the 32-bit draw >= thresh
  const double t = (double)p * 4294967296.0;
  const uint32_t thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  return thresh ? thresh : 1u;
}
