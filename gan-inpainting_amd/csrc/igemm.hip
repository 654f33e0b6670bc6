// MFMA implicit-GEMM kernels for the 4x4 / stride-2 / pad-1 convolution family (gfx950).
//
//   CONV  : out[n,y,x,a]        = sum_{ky,kx,b} in[n,2y-1+ky,2x-1+kx,b] * W[a][ky][kx][b]
//           (Conv2d forward, lib/models/networks.py:285,337-348; ConvTranspose2d input-gradient)
//   PHASE : out[n,2y+py,2x+px,b] = sum_{ty,tx,a} in[n,y+py-ty,x+px-tx,a] * W[a][ky][kx][b],
//           ky = 1-py+2ty, kx = 1-px+2tx   (ConvTranspose2d forward, networks.py:293-309, as four
//           sub-pixel 2x2 convolutions; Conv2d input-gradient)
//
// GEMM rows are small-resolution pixels, columns are output channels, K = taps x input channels.
// NHWC activations, K-contiguous weights. A/B tiles are staged global -> registers -> LDS
// (128-byte K rows, XOR-swizzled 16-byte chunks for fp16; k-major padded rows for fp32),
// double-buffered with the next tile's global loads issued before the MFMA block.
// fp16: v_mfma_f32_16x16x32_f16, fp32: v_mfma_f32_32x32x2_f32 (exact fp32), fp32 accumulate.
// Epilogue: optional bias + activation, per-tile column sum / sum-of-squares (BatchNorm batch
// statistics, deterministic: no atomics), result staged through LDS and stored with 16-byte
// coalesced rows.
// Small-M layers split the reduction (split-K): every split stores its fp32 tile in the thread-linear register layout
// (16 bytes per lane, coalesced) and takes a ticket; the LAST arriver of a tile adds the splits up in split order (a
// fixed order: results do not depend on which workgroup happens to be last) and runs the normal epilogue - no finish
// launch, statistics straight from the final values. Without a ticket buffer: per-split buffers + splitk_finish_kernel.
#include <stdlib.h>

#include "common.h"
#include "stat_acc.h"

namespace {

struct KP {
  const char* in;
  const char* w;
  char* out;
  const float* bias;
  float* partials;
  unsigned long long* stat_acc; int stat_pg, stat_reps;   // IgemmArgs::stat_acc
  float* ws;
  unsigned* tickets;   // split-K fix-up by the last arriver of each tile (zero before and after every launch), or null
  int M, Hs, Ws;
  int cin, ldin, coffin;
  int cout, ldout, coffout;
  int Ktot, nk, splitk, kt_per_split;
  int64_t ws_stride;   // > 0: split ks stores its partial sums at ws + ks * ws_stride (plain stores); 0: atomics into one buffer
  int relu_in, act_out;
  int Hin, Win, Hout, Wout;
  int64_t in_elems;
};

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == GI_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == GI_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
  return v;
}

template <typename T>
__device__ __forceinline__ u4_t relu16(u4_t v) {
  if constexpr (std::is_same<T, half_t>::value) {
    // max(x, 0) on packed halves as signed 16-bit integers: negative halves have the sign bit set
    typedef short s8_t __attribute__((ext_vector_type(8)));
    s8_t h = __builtin_bit_cast(s8_t, v);
    const s8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    h = __builtin_elementwise_max(h, z);
    return __builtin_bit_cast(u4_t, h);
  } else {
    f4_t f = __builtin_bit_cast(f4_t, v);
    f4_t z = {0.f, 0.f, 0.f, 0.f};
    f = __builtin_elementwise_max(f, z);
    return __builtin_bit_cast(u4_t, f);
  }
}

template <typename T, int PHASE, int BM, int BN, int WGM, int WGN>
__global__ void __launch_bounds__(256, 2) igemm_kernel(KP p) {
  constexpr bool F16 = std::is_same<T, half_t>::value;
  constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int BK = 8 * EPC;               // 128-byte K rows
  constexpr int AP = BM / 32, BP = BN / 32;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TS = F16 ? 16 : 32;         // MFMA tile side
  constexpr int MT = WM / TS, NT = WN / TS;
  constexpr int LDA32 = BM + 1, LDB32 = BN + 1;  // fp32 k-major padded leading dims
  constexpr int STAGE_BYTES = F16 ? (BM + BN) * 128 : 32 * (LDA32 + LDB32) * 4;
  constexpr int A_BYTES = F16 ? BM * 128 : 32 * LDA32 * 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int chunk = tid & 7;
  const int rbase = tid >> 3;

  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int ks = blockIdx.z % p.splitk;
  const int ph = blockIdx.z / p.splitk;
  const int py = ph >> 1, px = ph & 1;

  const char* wptr = p.w + (PHASE ? (int64_t)ph * p.cout * p.Ktot * (int64_t)sizeof(T) : 0);

  // ---- per-row gather bases and tap-validity masks ----------------------------------------
  int abase[AP];
  unsigned amask[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = m0 + rbase + 32 * i;
    abase[i] = 0;
    amask[i] = 0;
    if (m < p.M) {
      const int x = m % p.Ws;
      const int t = m / p.Ws;
      const int y = t % p.Hs;
      const int n = t / p.Hs;
      if (PHASE) {
        const int y0 = y + py, x0 = x + px;
        abase[i] = ((n * p.Hs + y0) * p.Ws + x0) * p.ldin + p.coffin;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int iy = y0 - (tt >> 1), ix = x0 - (tt & 1);
          if (iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws) amask[i] |= 1u << tt;
        }
      } else {
        const int y0 = 2 * y - 1, x0 = 2 * x - 1;
        abase[i] = ((n * p.Hin + y0) * p.Win + x0) * p.ldin + p.coffin;
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) {
          const int iy = y0 + (tt >> 2), ix = x0 + (tt & 3);
          if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) amask[i] |= 1u << tt;
        }
      }
    }
  }

  const int kt_begin = ks * p.kt_per_split;
  const int kt_end = min(p.nk, kt_begin + p.kt_per_split);
  int tap = (kt_begin * BK) / p.cin;
  int c0 = (kt_begin * BK) % p.cin;

  // ---- loop-invariant addressing ------------------------------------------------------------
  // A: per-row byte offsets for the CURRENT tap (recomputed only when the tap changes); rows whose
  //    tap falls into the padding get an out-of-range offset: the buffer load returns zeros.
  // B: per-row byte offsets (constant); the K position lives in the scalar buffer base.
  // LDS: write / fragment-read addresses are lane constants + immediates.
  constexpr unsigned OOB = 0x7FFFFFF0u;
  constexpr int ES = (int)sizeof(T);
  const unsigned in_bytes = (unsigned)min((int64_t)0x7FFFFF00, (int64_t)p.in_elems * ES);
  const unsigned w_bytes = (unsigned)min((int64_t)0x7FFFFF00, (int64_t)p.cout * p.Ktot * ES);
  unsigned voffA[AP], voffB[BP];
  auto set_tap = [&]() {
    int toff;
    if (PHASE) toff = -((tap >> 1) * p.Win + (tap & 1)) * p.ldin;
    else toff = ((tap >> 2) * p.Win + (tap & 3)) * p.ldin;
    static_for<AP>([&](auto I) {
      constexpr int i = decltype(I)::value;
      voffA[i] = ((amask[i] >> tap) & 1u) ? (unsigned)((abase[i] + toff) * ES + chunk * 16) : OOB;
    });
  };
  static_for<BP>([&](auto I) {
    constexpr int i = decltype(I)::value;
    voffB[i] = (unsigned)(((n0 + rbase + 32 * i) * p.Ktot) * ES + chunk * 16);
  });
  set_tap();

  u4_t ra[AP], rb[BP];
  auto gload = [&](int kt) {
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (int64_t)c0 * ES), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(wptr + (int64_t)kt * (BK * ES)), 0, w_bytes, 0x00020000);
    static_for<AP>([&](auto I) {
      constexpr int i = decltype(I)::value;
      ra[i] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA[i], 0, 0));
    });
    static_for<BP>([&](auto I) {
      constexpr int i = decltype(I)::value;
      rb[i] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB[i], 0, 0));
    });
    c0 += BK;
    if (c0 >= p.cin) { c0 = 0; ++tap; set_tap(); }
  };

  // lane-constant LDS offsets
  const int wrow = rbase;                                            // rows rbase + 32 i share rbase & 7
  const int woff16 = wrow * 128 + ((chunk ^ (wrow & 7)) << 4);       // fp16 write offset (+ i*4096)
  const int lr = lane & 15, lq = lane >> 4;
  const int rdA0 = (wm * WM + lr) * 128 + (((0 + lq) ^ (lr & 7)) << 4);   // k2 = 0 (+ mt*2048)
  const int rdA1 = (wm * WM + lr) * 128 + (((4 + lq) ^ (lr & 7)) << 4);   // k2 = 1
  const int rdB0 = (wn * WN + lr) * 128 + (((0 + lq) ^ (lr & 7)) << 4);
  const int rdB1 = (wn * WN + lr) * 128 + (((4 + lq) ^ (lr & 7)) << 4);

  auto lds_store = [&](auto STG) {
    constexpr int stage = decltype(STG)::value;
    char* sA = smem + stage * STAGE_BYTES;
    char* sB = sA + A_BYTES;
    if (p.relu_in) {
      static_for<AP>([&](auto I) { ra[decltype(I)::value] = relu16<T>(ra[decltype(I)::value]); });
    }
    if constexpr (F16) {
      static_for<AP>([&](auto I) { *(u4_t*)(sA + woff16 + decltype(I)::value * 4096) = ra[decltype(I)::value]; });
      static_for<BP>([&](auto I) { *(u4_t*)(sB + woff16 + decltype(I)::value * 4096) = rb[decltype(I)::value]; });
    } else {
      float* fA = (float*)sA + chunk * 4 * LDA32 + rbase;
      float* fB = (float*)sB + chunk * 4 * LDB32 + rbase;
      static_for<AP>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const f4_t v = __builtin_bit_cast(f4_t, ra[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) fA[j * LDA32 + 32 * i] = v[j];
      });
      static_for<BP>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const f4_t v = __builtin_bit_cast(f4_t, rb[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) fB[j * LDB32 + 32 * i] = v[j];
      });
    }
  };

  using acc_t = typename std::conditional<F16, f4_t, f16_t>::type;
  acc_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < (F16 ? 4 : 16); ++r) acc[i][j][r] = 0.f;

  auto compute = [&](auto STG) {
    constexpr int stage = decltype(STG)::value;
    const char* sA = smem + stage * STAGE_BYTES;
    const char* sB = sA + A_BYTES;
    if constexpr (F16) {
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        h8_t af[MT], bf[NT];
        const int oa = k2 ? rdA1 : rdA0, ob = k2 ? rdB1 : rdB0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *(const h8_t*)(sA + oa + mt * 2048);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = *(const h8_t*)(sB + ob + nt * 2048);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
      }
    } else {
      const float* fA = (const float*)sA + (lane >> 5) * LDA32 + wm * WM + (lane & 31);
      const float* fB = (const float*)sB + (lane >> 5) * LDB32 + wn * WN + (lane & 31);
#pragma unroll 4
      for (int kk = 0; kk < 16; ++kk) {
        float af[MT], bf[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = fA[2 * kk * LDA32 + mt * 32];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = fB[2 * kk * LDB32 + nt * 32];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
      }
    }
  };

  // ---- main loop: one barrier per K tile, next tile's loads in flight during the MFMAs; the two
  //      LDS stages are separate code copies so every LDS address is a lane constant + immediate -----
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if (kt_begin < kt_end) {
    gload(kt_begin);
    lds_store(S0{});
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; kt += 2) {
      const bool more1 = (kt + 1 < kt_end);
      if (more1) gload(kt + 1);
      compute(S0{});
      if (more1) lds_store(S1{});
      __syncthreads();
      if (!more1) break;
      const bool more2 = (kt + 2 < kt_end);
      if (more2) gload(kt + 2);
      compute(S1{});
      if (more2) lds_store(S0{});
      __syncthreads();
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------
  auto elem_row = [&](int mt, int r) -> int {
    if constexpr (F16) return wm * WM + mt * 16 + (lane >> 4) * 4 + r;
    else return wm * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  };
  auto elem_col = [&](int nt) -> int {
    if constexpr (F16) return wn * WN + nt * 16 + (lane & 15);
    else return wn * WN + nt * 32 + (lane & 31);
  };
  auto out_pixel = [&](int m) -> int {
    if (!PHASE) return m;
    const int x = m % p.Ws;
    const int t = m / p.Ws;
    const int y = t % p.Hs;
    const int n = t / p.Hs;
    return (n * p.Hout + 2 * y + py) * p.Wout + 2 * x + px;
  };
  constexpr int NR = F16 ? 4 : 16;

  if (p.splitk > 1 && p.tickets) {
    // Coherence across the XCDs' L2 caches WITHOUT agent-scope fences (a fence is buffer_wbl2 + buffer_inv: it writes back
    // and invalidates the whole L2 under the other workgroups' weight streams): the tiles are stored and loaded with the
    // sc1 bit (what the compiler emits for agent-scope relaxed atomics: write-through / miss-always), the stores are
    // waited for (vmcnt) before the ticket is taken.
    constexpr int Q4 = NR / 4, NF = MT * NT * Q4;
    const int tile = (ph * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int ntile = gridDim.x * gridDim.y * (gridDim.z / p.splitk);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.ws, 0, 0x7FFFFF00u, 0x00020000);
    constexpr int TILE_BYTES = BM * BN * 4;
    const unsigned mine = (unsigned)(ks * ntile + tile) * TILE_BYTES + tid * 16;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < Q4; ++q) {
          const f4_t v = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsW, mine + ((mt * NT + nt) * Q4 + q) * 4096, 0, 16);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile has reached the coherent level
    __syncthreads();
    __shared__ int s_last;
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(p.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == (unsigned)p.splitk - 1u) ? 1 : 0;
      if (s_last) __hip_atomic_store(p.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // all arrived: ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[mt][nt][r] = 0.f;
    // split order, whoever is last; the tiles of TWO splits are requested before the first is added (igemm7.hip's tail: one split
    // per iteration is a chain of `splitk` dependent memory round trips). A split beyond the last: out-of-range offset, not added.
    constexpr int PF = 2;
    for (int k0 = 0; k0 < p.splitk; k0 += PF) {
      f4_t v[PF][NF];
#pragma unroll
      for (int f = 0; f < PF; ++f) {
        const unsigned src = k0 + f < p.splitk ? (unsigned)((k0 + f) * ntile + tile) * TILE_BYTES + tid * 16 : 0x80000000u;
#pragma unroll
        for (int i = 0; i < NF; ++i) v[f][i] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rsW, src + i * 4096, 0, 16));
      }
#pragma unroll
      for (int f = 0; f < PF; ++f) {
        if (k0 + f < p.splitk) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int q = 0; q < Q4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mt][nt][4 * q + j] += v[f][(mt * NT + nt) * Q4 + q][j];
        }
      }
    }
  } else if (p.splitk > 1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int m = m0 + elem_row(mt, r);
        if (m < p.M) {
          const int64_t o = (int64_t)out_pixel(m) * p.cout + n0;
          if (p.ws_stride > 0) {
            float* dst = p.ws + (int64_t)ks * p.ws_stride + o;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) dst[elem_col(nt)] = acc[mt][nt][r];
          } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) atomicAdd(p.ws + o + elem_col(nt), acc[mt][nt][r]);
          }
        }
      }
    return;
  }

  constexpr int SPAD = 16 / (int)sizeof(T);
  constexpr int SLD = BN + SPAD;
  T* stg = (T*)smem;
  float* red = (float*)(smem + (int64_t)BM * SLD * sizeof(T));  // [WGM][BN][2]

  const bool stats = p.partials || p.stat_acc;
  gi_with_act(p.act_out, [&](auto ACTc) {              // the activation as a compile-time constant (common.h): no test per element
    constexpr int ACT = decltype(ACTc)::value;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = elem_col(nt);
      const float b = p.bias ? p.bias[n0 + col] : 0.f;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          float v = acc[mt][nt][r] + b;
          s += v;
          q += v * v;
          stg[elem_row(mt, r) * SLD + col] = (T)gi_act_c<ACT>(v);
        }
      if (stats) {
        if constexpr (F16) {
          s += __shfl_xor(s, 16); q += __shfl_xor(q, 16);
          s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
          if (lane < 16) { red[(wm * BN + col) * 2] = s; red[(wm * BN + col) * 2 + 1] = q; }
        } else {
          s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
          if (lane < 32) { red[(wm * BN + col) * 2] = s; red[(wm * BN + col) * 2 + 1] = q; }
        }
      }
    }
  });
  __syncthreads();
  if ((p.partials || p.stat_acc) && tid < BN) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < WGM; ++i) { s += red[(i * BN + tid) * 2]; q += red[(i * BN + tid) * 2 + 1]; }
    if (p.stat_acc) {
      const int grp = (p.stat_pg > 0 && m0 >= p.stat_pg) ? 1 : 0, rep = (blockIdx.x + ph) & (p.stat_reps - 1);
      gi_stat_add(p.stat_acc, p.cout, rep, grp, 0, n0 + tid, s);
      gi_stat_add(p.stat_acc, p.cout, rep, grp, 1, n0 + tid, q);
    } else {
      const int64_t trow = (int64_t)blockIdx.x + (int64_t)gridDim.x * ph;
      p.partials[(trow * 2 + 0) * p.cout + n0 + tid] = s;
      p.partials[(trow * 2 + 1) * p.cout + n0 + tid] = q;
    }
  }
  constexpr int CPRO = BN / EPC;          // 16-byte chunks per output row
  constexpr int RPP = 256 / CPRO;         // rows per pass
  const int oc = tid % CPRO;
#pragma unroll 1
  for (int r = tid / CPRO; r < BM; r += RPP) {
    const int m = m0 + r;
    if (m < p.M) {
      const int64_t o = (int64_t)out_pixel(m) * p.ldout + p.coffout + n0 + oc * EPC;
      *(u4_t*)(p.out + o * (int64_t)sizeof(T)) = *(const u4_t*)((const char*)stg + ((int64_t)r * SLD + oc * EPC) * sizeof(T));
    }
  }
}

// split-K finish: ws (fp32, dense [pixels][cout]) -> bias/act -> T out (+ column statistics)
template <typename T>
__global__ void __launch_bounds__(256) splitk_finish_kernel(const float* ws, const float* bias, char* out,
                                                            float* partials, int64_t pixels, int cout,
                                                            int ldout, int coffout, int act, int rows_per_block,
                                                            int nsplit, int64_t ws_stride, unsigned long long* stat_acc,
                                                            int64_t stat_pg_out, int stat_reps) {
  __shared__ float red[2 * 256 * 4];
  const int Q = cout / 4;            // column quads
  const int RL = 256 / Q;            // row lanes (cout <= 1024)
  const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(pixels, r0 + rows_per_block);
  f4_t b = {0.f, 0.f, 0.f, 0.f};
  if (bias) b = *(const f4_t*)(bias + q * 4);
  f4_t s = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
  if (rl < RL) {
    for (int64_t r = r0 + rl; r < r1; r += RL) {
      f4_t v = *(const f4_t*)(ws + r * cout + q * 4) + b;
      for (int k = 1; k < nsplit; ++k) v += *(const f4_t*)(ws + (int64_t)k * ws_stride + r * cout + q * 4);   // fixed order
      s += v;
      sq += v * v;
      T o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (T)apply_act(v[j], act);
      T* dst = (T*)(out + (r * ldout + coffout + q * 4) * (int64_t)sizeof(T));
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[j] = o[j];
    }
  }
  if (!partials && !stat_acc) return;
  // reduce over row lanes
  for (int j = 0; j < 4; ++j) {
    red[(threadIdx.x) * 4 + j] = s[j];
    red[1024 + threadIdx.x * 4 + j] = sq[j];
  }
  __syncthreads();
  if (stat_acc) {   // one lane per channel: the lanes of an atomic instruction are consecutive words (stat_acc.h)
    // rows of one block never straddle two BatchNorm populations (rows_per_block divides stat_pg_out)
    const int grp = (stat_pg_out > 0 && r0 >= stat_pg_out) ? 1 : 0, rep = blockIdx.x & (stat_reps - 1);
    for (int ch = threadIdx.x; ch < cout; ch += 256) {
      float a = 0.f, b2 = 0.f;
      for (int i = 0; i < RL; ++i) { a += red[i * Q * 4 + ch]; b2 += red[1024 + i * Q * 4 + ch]; }
      gi_stat_add(stat_acc, cout, rep, grp, 0, ch, a);
      gi_stat_add(stat_acc, cout, rep, grp, 1, ch, b2);
    }
  } else if (rl == 0) {
    for (int i = 1; i < RL; ++i)
      for (int j = 0; j < 4; ++j) {
        s[j] += red[(i * Q + q) * 4 + j];
        sq[j] += red[1024 + (i * Q + q) * 4 + j];
      }
    float* ps = partials + ((int64_t)blockIdx.x * 2) * cout + q * 4;
    for (int j = 0; j < 4; ++j) { ps[j] = s[j]; ps[cout + j] = sq[j]; }
  }
}

template <typename T, int PHASE, int BM, int BN, int WGM, int WGN>
int launch_cfg(hipStream_t st, const KP& kp, dim3 grid) {
  constexpr bool F16 = std::is_same<T, half_t>::value;
  constexpr int STAGE_BYTES = F16 ? (BM + BN) * 128 : 32 * (BM + 1 + BN + 1) * 4;
  constexpr int EPI = BM * (BN + 16 / (int)sizeof(T)) * (int)sizeof(T) + WGM * BN * 2 * 4;
  constexpr int LDS = (2 * STAGE_BYTES > EPI ? 2 * STAGE_BYTES : EPI);
  auto kern = igemm_kernel<T, PHASE, BM, BN, WGM, WGN>;
  static GiDevOnce attr_set;
  if (attr_set.first()) {
    GI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), LDS, st, kp);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

template <typename T, int PHASE>
int run(hipStream_t st, IgemmArgs& a) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BK = 8 * EPC;
  GI_REQUIRE(a.cin % BK == 0, "igemm: cin=%d must be a multiple of %d", a.cin, BK);
  GI_REQUIRE(a.cout % 64 == 0, "igemm: cout=%d must be a multiple of 64", a.cout);
  GI_REQUIRE(a.ldin % EPC == 0 && a.coffin % EPC == 0 && a.ldout % EPC == 0 && a.coffout % EPC == 0,
             "igemm: leading dims / channel offsets must be 16-byte aligned");
  KP kp;
  kp.in = (const char*)a.in; kp.w = (const char*)a.w; kp.out = (char*)a.out;
  kp.bias = a.bias; kp.partials = a.stat_acc ? nullptr : a.partials; kp.ws = a.ws; kp.tickets = nullptr;
  kp.stat_acc = a.stat_acc; kp.stat_pg = a.stat_pg; kp.stat_reps = a.stat_reps > 0 ? a.stat_reps : 1;
  a.stat_used = a.stat_acc ? 1 : 0;
  kp.M = a.n * a.Hs * a.Ws; kp.Hs = a.Hs; kp.Ws = a.Ws;
  kp.cin = a.cin; kp.ldin = a.ldin; kp.coffin = a.coffin;
  kp.cout = a.cout; kp.ldout = a.ldout; kp.coffout = a.coffout;
  kp.Ktot = (PHASE ? 4 : 16) * a.cin;
  kp.nk = kp.Ktot / BK;
  kp.relu_in = a.relu_in; kp.act_out = a.act_out;
  if (PHASE) { kp.Hin = a.Hs; kp.Win = a.Ws; kp.Hout = 2 * a.Hs; kp.Wout = 2 * a.Ws; }
  else { kp.Hin = 2 * a.Hs; kp.Win = 2 * a.Ws; kp.Hout = a.Hs; kp.Wout = a.Ws; }
  const int64_t in_elems = (int64_t)a.n * kp.Hin * kp.Win * a.ldin;
  kp.in_elems = in_elems;
  const int64_t out_pixels = (int64_t)a.n * kp.Hout * kp.Wout;
  GI_REQUIRE(in_elems < (1ll << 31) && out_pixels * a.ldout < (1ll << 31), "igemm: tensor too large for 32-bit offsets");

  const bool wide = (a.cout % 128 == 0);
  int BM = wide ? 128 : 256, BN = wide ? 128 : 64;
  const int phases = PHASE ? 4 : 1;
  int mt = (kp.M + BM - 1) / BM, nt = a.cout / BN;
  int tiles = mt * nt * phases;
  int splitk = 1;
  // in-kernel fix-up: the last arriver reads splits x tile bytes on ONE CU (~100 GB/s), so the split count is capped
  // and very small M gets 128 x 64 tiles instead (twice the workgroups for the same tail)
  const int fix = gi_opt(GI_OPT_IGEMM_FIXUP);   // GI_IGEMM_FIXUP=0: finish-kernel path
  int fix_max = gi_tune("GI_IGEMM_FIX_MAXSPLIT", 8);
  if (fix_max < 2) fix_max = 8;
  bool fixup = fix && a.tickets && a.ws && a.force_splitk == 0 && tiles < 256 && kp.nk >= 8;
  bool half_n = false;
  if (fixup) {
    if (wide && tiles * fix_max < 256) { half_n = true; BN = 64; nt = a.cout / BN; tiles = mt * nt * phases; }
    splitk = (256 + tiles - 1) / tiles;
    if (splitk > fix_max) splitk = fix_max;
    if (splitk > kp.nk / 4) splitk = kp.nk / 4;
    if (splitk < 1) splitk = 1;
    if (tiles > GI_IGEMM_TICKETS || a.ws_bytes < (int64_t)splitk * tiles * BM * BN * 4) {
      fixup = false; splitk = 1;
      if (half_n) { half_n = false; BN = 128; nt = a.cout / BN; tiles = mt * nt * phases; }
    }
  }
  if (fixup) {
  } else if (a.force_splitk > 0) splitk = a.force_splitk;
  else if (tiles < 256 && kp.nk >= 8) {   // fewer workgroups than CUs: split the reduction
    const int target = gi_tune("GI_IGEMM_SPLIT_BLOCKS", 384);   // workgroups to aim for
    splitk = (target + tiles - 1) / tiles;
    if (splitk > kp.nk / 4) splitk = kp.nk / 4;
    if (splitk > 64) splitk = 64;
    if (splitk < 1) splitk = 1;
  }
  if (splitk > 1 && (a.ws == nullptr || a.ws_bytes < out_pixels * a.cout * 4)) splitk = 1;
  kp.kt_per_split = (kp.nk + splitk - 1) / splitk;
  splitk = (kp.nk + kp.kt_per_split - 1) / kp.kt_per_split;
  kp.splitk = splitk;
  kp.ws_stride = 0;
  if (splitk <= 1) fixup = false;
  if (fixup) kp.tickets = a.tickets;
  if (splitk > 1 && !fixup) {
    // scratch for one buffer per split: plain stores + a summing finish pass (deterministic, no memset);
    // otherwise fp32 atomics into a single zeroed buffer
    if (a.ws_bytes >= (int64_t)splitk * out_pixels * a.cout * 4) kp.ws_stride = out_pixels * a.cout;
    else GI_HIP(hipMemsetAsync(a.ws, 0, out_pixels * a.cout * 4, st));
    kp.partials = nullptr;
    kp.stat_acc = nullptr;
  }
  GI_REQUIRE(!a.stat_acc || a.stat_pg == 0 || a.stat_pg % 256 == 0, "igemm: stat_pg=%d must be a multiple of 256", a.stat_pg);
  dim3 grid(mt, nt, phases * splitk);
  if (half_n) GI_TRY((launch_cfg<T, PHASE, 128, 64, 2, 2>(st, kp, grid)));
  else if (wide) GI_TRY((launch_cfg<T, PHASE, 128, 128, 2, 2>(st, kp, grid)));
  else GI_TRY((launch_cfg<T, PHASE, 256, 64, 4, 1>(st, kp, grid)));
  gi_note_kernel(std::is_same<T, float>::value ? (fixup ? "igemm<f32,fixup>" : (splitk > 1 ? "igemm<f32,splitk>" : "igemm<f32>"))
                                               : (fixup ? "igemm<f16,fixup>" : (splitk > 1 ? "igemm<f16,splitk>" : "igemm<f16>")));
  a.ntiles_out = mt * phases;
  if (splitk > 1 && !fixup) {
    const int RL = 256 / (a.cout / 4) > 0 ? 256 / (a.cout / 4) : 1;   // row lanes per block
    int rpb = 64;
    while (rpb > RL && (out_pixels + rpb - 1) / rpb < 256) rpb >>= 1;   // fill the chip on small tensors
    if (rpb < RL) rpb = RL;
    const int blocks = (int)((out_pixels + rpb - 1) / rpb);
    GI_REQUIRE(a.cout <= 1024, "igemm split-K finish: cout=%d > 1024", a.cout);
    hipLaunchKernelGGL(splitk_finish_kernel<T>, dim3(blocks), dim3(256), 0, st, a.ws, a.bias, (char*)a.out,
                       a.partials, out_pixels, a.cout, a.ldout, a.coffout, a.act_out, rpb, kp.ws_stride > 0 ? splitk : 1,
                       kp.ws_stride, a.partials ? nullptr : a.stat_acc, (int64_t)a.stat_pg * (PHASE ? 4 : 1), kp.stat_reps);
    a.stat_used = (a.stat_acc && !a.partials) ? 1 : 0;
    GI_LAUNCH_CHECK();
    a.ntiles_out = blocks;
  }
  return GI_OK;
}

}  // namespace

int op_igemm3(hipStream_t st, int mode, IgemmArgs& a);   // igemm3.hip: 256xBN tiles, LDS-DMA ring
int op_igemm7(hipStream_t st, int mode, IgemmArgs& a);   // igemm7.hip: 128xBN tiles, 4-stage ring, split-K with in-kernel fix-up

int op_igemm(hipStream_t st, int dtype, int phase_mode, IgemmArgs& a) {
  // GI_IGEMM_VARIANT: 3 = LDS-DMA kernels (default), 1 = the register-staged kernel of this file only (what fp32 always runs)
  const int variant = gi_opt(GI_OPT_IGEMM_VARIANT);
  if (dtype == GI_F16 && a.force_splitk == 0 && variant >= 3) {
    int rc = op_igemm3(st, phase_mode, a);
    if (rc != GI_ERR_UNSUPPORTED) return rc;
    rc = op_igemm7(st, phase_mode, a);      // small-M layers: deep LDS-DMA ring + in-kernel split-K reduction
    if (rc != GI_ERR_UNSUPPORTED) return rc;
  }
  if (dtype == GI_F16) return phase_mode ? run<half_t, 1>(st, a) : run<half_t, 0>(st, a);
  if (dtype == GI_F32) return phase_mode ? run<float, 1>(st, a) : run<float, 0>(st, a);
  gi_set_error("igemm: bad dtype %d", dtype);
  return GI_ERR_INVALID;
}
