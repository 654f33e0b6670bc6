// Exact, order-independent per-channel sums (BatchNorm batch statistics and their backward reductions).
//
// The producers (GEMM epilogues, reduction passes) hold one fp32 column sum per tile. Writing a partial row per
// tile and adding the rows up in a separate small launch costs a dependent launch per layer (a 2 us body behind
// ~3 us of launch gaps, 13 times per generator forward). Here every tile ADDS its fp32 value into a per-channel
// accumulator with 64-bit integer atomics: an fp32 number (24-bit mantissa) times 2^40 is an integer (fractions
// below 2^-40 are cut), kept as hi * 2^32 + lo with lo in [0, 2^32); the halves go to two words. Integer addition
// commutes and never rounds, so the totals do not depend on the order the tiles finish in: results are
// bit-reproducible, unlike fp32 atomics. Consumers convert back to double once.
//   bit 63 of lo = "a non-finite or absurd value was added" -> the total reads NaN
//   range: the hi word counts units of 2^-8, so a total must stay below 2^55 in magnitude (signed 64 bits); an addend is
//   accepted below 2^40 (1.1e12: a 256-row tile of fp16's largest squares) and a replica receives at most 2^13 addends
//   (tiles of a layer / replicas), i.e. |total| < 2^53; gi_stat_read returns NaN for a total beyond 2^54 all the same
//
// Layout of one accumulator block (structure of arrays): word[replica][group][w][c], w = {sum hi, sum lo, sumsq hi,
// sumsq lo} (or any two quantities), group = BatchNorm population (0 / 1). The 64 lanes of one atomic wave
// instruction are 64 consecutive channels of one w, i.e. 512 contiguous bytes = eight 64-byte requests at the
// memory-side atomic unit; with the words of a channel interleaved instead every request would carry two lanes, and
// requests to one line are served one after the other (~12 ns each: a layer of 2048 tiles queued for 100 us).
// Replicas (tile index mod R) bound the requests per line to a few hundred; the consumer adds the replicas up.
#pragma once
#include <hip/hip_runtime.h>

constexpr int GI_STAT_WORDS = 4;     // 64-bit words per channel, population and replica
constexpr int GI_STAT_MAXREP = 4;

// words of one block with `reps` replicas
__host__ __device__ __forceinline__ long long gi_stat_block_words(int c, int reps) { return (long long)reps * 2 * GI_STAT_WORDS * c; }
// address of word w of channel 0 (replica, group); channels are consecutive
__device__ __forceinline__ unsigned long long* gi_stat_ptr(unsigned long long* base, int c, int rep, int group, int w) {
  return base + ((long long)(rep * 2 + group) * GI_STAT_WORDS + w) * c;
}

// quantity q (0 / 1) of channel ch
__device__ __forceinline__ void gi_stat_add(unsigned long long* base, int c, int rep, int group, int q, int ch, float s) {
  unsigned long long* hi = gi_stat_ptr(base, c, rep, group, 2 * q) + ch;
  unsigned long long* lo = hi + c;
  if (!(fabsf(s) < 1.0995116e12f)) {     // inf, NaN, or beyond the fixed-point range (2^40)
    atomicOr(lo, 0x8000000000000000ull);
    return;
  }
  const double d = (double)s * 1099511627776.0;               // * 2^40, exact
  const double h = floor(d * 2.3283064365386963e-10);         // / 2^32
  const double l = d - h * 4294967296.0;                      // in [0, 2^32), exact
  atomicAdd(hi, (unsigned long long)(long long)h);            // two's complement: negative h wraps correctly
  atomicAdd(lo, (unsigned long long)l);
}

// Totals of channel ch over `reps` replicas. COHERENT: the adds came from other workgroups of the SAME launch; the words are then
// read with agent-scope relaxed atomic loads (sc1), not plain ones that this CU's L1 may serve.
// Every word is requested before any is used: the loads of all GI_STAT_MAXREP replicas go out unconditionally (replicas beyond
// `reps` re-read the last one and are masked out afterwards; a load under a run-time condition would be branched around and
// waited for one by one). The first form looped over the replicas with a wait per iteration - up to eight dependent memory round
// trips at the head of every normalisation pass, ~2 us each while other CUs stream.
template <bool COHERENT>
__device__ __forceinline__ unsigned long long gi_stat_word(const unsigned long long* p) {
  if constexpr (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
__device__ __forceinline__ double gi_stat_total(unsigned long long hi, unsigned long long lo, unsigned long long bad) {
  const long long h = (long long)hi;
  if (bad || h > (1ll << 62) || h < -(1ll << 62)) return __builtin_nan("");   // (2^62 units of 2^-8 = 2^54)
  return ((double)h * 4294967296.0 + (double)lo) * 9.094947017729282e-13;   // * 2^-40
}
// both quantities of a channel at once (sixteen independent loads)
template <bool COHERENT = false>
__device__ __forceinline__ void gi_stat_read2(const unsigned long long* base, int c, int reps, int group, int ch, double& q0, double& q1) {
  unsigned long long w[GI_STAT_MAXREP][GI_STAT_WORDS];
#pragma unroll
  for (int r = 0; r < GI_STAT_MAXREP; ++r) {
    const int rr = r < reps ? r : reps - 1;
    const unsigned long long* p = base + ((long long)(rr * 2 + group) * GI_STAT_WORDS) * c + ch;
#pragma unroll
    for (int k = 0; k < GI_STAT_WORDS; ++k) w[r][k] = gi_stat_word<COHERENT>(p + (long long)k * c);
  }
  unsigned long long hi0 = 0, lo0 = 0, hi1 = 0, lo1 = 0, bad0 = 0, bad1 = 0;
#pragma unroll
  for (int r = 0; r < GI_STAT_MAXREP; ++r) {
    const unsigned long long m = r < reps ? ~0ull : 0ull;
    hi0 += w[r][0] & m; lo0 += w[r][1] & m & 0x7FFFFFFFFFFFFFFFull; bad0 |= (w[r][1] & m) >> 63;
    hi1 += w[r][2] & m; lo1 += w[r][3] & m & 0x7FFFFFFFFFFFFFFFull; bad1 |= (w[r][3] & m) >> 63;
  }
  q0 = gi_stat_total(hi0, lo0, bad0);
  q1 = gi_stat_total(hi1, lo1, bad1);
}
// total of quantity q of channel ch
template <bool COHERENT = false>
__device__ __forceinline__ double gi_stat_read(const unsigned long long* base, int c, int reps, int group, int q, int ch) {
  unsigned long long w[GI_STAT_MAXREP][2];
#pragma unroll
  for (int r = 0; r < GI_STAT_MAXREP; ++r) {
    const int rr = r < reps ? r : reps - 1;
    const unsigned long long* p = base + ((long long)(rr * 2 + group) * GI_STAT_WORDS + 2 * q) * c + ch;
    w[r][0] = gi_stat_word<COHERENT>(p);
    w[r][1] = gi_stat_word<COHERENT>(p + c);
  }
  unsigned long long hi = 0, lo = 0, bad = 0;
#pragma unroll
  for (int r = 0; r < GI_STAT_MAXREP; ++r) {
    const unsigned long long m = r < reps ? ~0ull : 0ull;
    hi += w[r][0] & m; lo += w[r][1] & m & 0x7FFFFFFFFFFFFFFFull; bad |= (w[r][1] & m) >> 63;
  }
  return gi_stat_total(hi, lo, bad);
}
