// Exact, order-independent per-channel sums (BatchNorm batch statistics and their backward reductions).
//
// The producers (GEMM epilogues, reduction passes) hold one fp32 column sum per tile. Writing a partial row per
// tile and adding the rows up in a separate small launch costs a dependent launch per layer (a 2 us body behind
// ~3 us of launch gaps, 13 times per generator forward). Here every tile ADDS its fp32 value into a per-channel
// accumulator with 64-bit integer atomics: an fp32 number (24-bit mantissa) times 2^40 is an integer (fractions
// below 2^-40 are cut), kept as hi * 2^32 + lo with lo in [0, 2^32); the halves go to two words. Integer addition
// commutes and never rounds, so the totals do not depend on the order the tiles finish in: results are
// bit-reproducible, unlike fp32 atomics. Consumers convert back to double once.
//   words per quantity: {hi, lo}; bit 63 of lo = "a non-finite or absurd value was added" -> the total reads NaN
//   range: |value| < 2^43 per addend, up to 2^19 addends
#pragma once
#include <hip/hip_runtime.h>

constexpr int GI_STAT_WORDS = 4;   // per channel: {sum hi, sum lo, sumsq hi, sumsq lo} (or any two quantities)

__device__ __forceinline__ void gi_stat_add(unsigned long long* w2, float s) {
  if (!(fabsf(s) < 8.0e12f)) {     // inf, NaN, or beyond the fixed-point range
    atomicOr(w2 + 1, 0x8000000000000000ull);
    return;
  }
  const double d = (double)s * 1099511627776.0;               // * 2^40, exact
  const double h = floor(d * 2.3283064365386963e-10);         // / 2^32
  const double l = d - h * 4294967296.0;                      // in [0, 2^32), exact
  atomicAdd(w2, (unsigned long long)(long long)h);            // two's complement: negative h wraps correctly
  atomicAdd(w2 + 1, (unsigned long long)l);
}

__device__ __forceinline__ double gi_stat_read(const unsigned long long* w2) {
  const unsigned long long hi = w2[0], lo = w2[1];
  if (lo >> 63) return __builtin_nan("");
  return ((double)(long long)hi * 4294967296.0 + (double)lo) * 9.094947017729282e-13;   // * 2^-40
}
