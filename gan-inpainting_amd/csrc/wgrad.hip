// Weight-gradient GEMM for the 4x4 / stride-2 / pad-1 family (gfx950):
//
//   dW[a][ky][kx][b] += scale * sum_{n,y,x} S[n,y,x,a] * L[n,2y-1+ky,2x-1+kx,b]
//
// S is the small-resolution tensor (Conv2d: output gradient; ConvTranspose2d: its input),
// L the large-resolution one (Conv2d: its input; ConvTranspose2d: output gradient), both NHWC.
// GEMM: M = a, N = (ky,kx,b), K = pixels. Both operands arrive pixel-major ([pixel][channel]),
// i.e. K-major, so the MFMA fragments are read from LDS with the gfx950 transposing read
// ds_read_b64_tr_b16 (fp16) or directly (fp32, one value per lane). dW is [a][16*b] row-major
// fp32 and is accumulated with global float atomics (split-K over pixel ranges fills the chip:
// the shallow layers have only a handful of output tiles but 10^5 pixels).
#include <stdlib.h>

#include "common.h"

namespace {

struct WP {
  const char* S;
  const char* L;
  float* dW;
  float* part;       // non-null: split ks writes its tile to part[ks][ca][16*cb] with plain stores (no atomics)
  int direct;        // 1: a single split owns every output element: dW += acc without atomics
  int P;             // pixels = n*Hs*Ws
  int lgWs, lgHs;    // log2 of Ws, Hs when both are powers of two, else -1 (divisions in the loader)
  int Hs, Ws, HL, WL;
  int ca, ldS, coffS;
  int cb, lgcb, ldL, coffL;
  int relu_S;
  float scale;
  int tiles_per_split;   // K tiles (BKP pixels) per split
};

template <typename T>
__device__ __forceinline__ u4_t relu16w(u4_t v) {
  if constexpr (std::is_same<T, half_t>::value) {
    h8_t h = __builtin_bit_cast(h8_t, v);
    h8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    h = __builtin_elementwise_max(h, z);
    return __builtin_bit_cast(u4_t, h);
  } else {
    f4_t f = __builtin_bit_cast(f4_t, v);
    f4_t z = {0.f, 0.f, 0.f, 0.f};
    f = __builtin_elementwise_max(f, z);
    return __builtin_bit_cast(u4_t, f);
  }
}

// tile: 128 (a) x 128 (tap,b) output, K tile = 32 pixels, 4 waves as 2x2 of 64x64
template <typename T>
__global__ void __launch_bounds__(256) wgrad_kernel(WP p) {
  constexpr bool F16 = std::is_same<T, half_t>::value;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BKP = F16 ? 64 : 32;            // pixels per K tile (fp16: 2 MFMA k-steps per barrier)
  constexpr int ROWB = 128 * (int)sizeof(T);    // bytes per tile row (128 channels)
  constexpr int LROW = F16 ? ROWB + 32 : ROWB;  // padded LDS row (fp16: +32 B -> conflict-free tr reads)
  constexpr int CPR = ROWB / 16;                // 16-byte chunks per row
  constexpr int RPP = 256 / CPR;                // rows per pass
  constexpr int NP = BKP / RPP;                 // passes
  constexpr int OPB = BKP * LROW;               // bytes per operand tile
  constexpr int TS = F16 ? 16 : 32;
  constexpr int MT = 64 / TS, NT = 64 / TS;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int chunk = tid % CPR, rbase = tid / CPR;

  const int a0 = blockIdx.x * 128;
  const int n0 = blockIdx.y * 128;     // column into (tap,b)
  const int ks = blockIdx.z;

  // this thread's L-side column chunk: fixed tap and channel offset
  const int col = n0 + chunk * EPC;
  const int tap = col >> p.lgcb;
  const int bch = col & (p.cb - 1);
  const int ky = tap >> 2, kx = tap & 3;

  const int t_begin = ks * p.tiles_per_split;
  const int t_total = (p.P + BKP - 1) / BKP;
  const int t_end = min(t_total, t_begin + p.tiles_per_split);

  u4_t rs[NP], rl[NP];
  auto gload = [&](int t) {
    static_for<NP>([&](auto I) {
      constexpr int i = decltype(I)::value;
      const int pix = t * BKP + rbase + RPP * i;
      u4_t vs = u4_t{0u, 0u, 0u, 0u}, vl = u4_t{0u, 0u, 0u, 0u};
      if (pix < p.P) {
        vs = *(const u4_t*)(p.S + ((int64_t)pix * p.ldS + p.coffS + a0 + chunk * EPC) * (int64_t)sizeof(T));
        int x, y, n;
        if (p.lgWs >= 0) {
          x = pix & (p.Ws - 1);
          y = (pix >> p.lgWs) & (p.Hs - 1);
          n = pix >> (p.lgWs + p.lgHs);
        } else {      // maps that are not powers of two (e.g. 192 x 192 inputs): three integer divisions per 16-byte load
          const int t = pix / p.Ws;
          x = pix - t * p.Ws;
          n = t / p.Hs;
          y = t - n * p.Hs;
        }
        const int iy = 2 * y - 1 + ky, ix = 2 * x - 1 + kx;
        if (iy >= 0 && iy < p.HL && ix >= 0 && ix < p.WL)
          vl = *(const u4_t*)(p.L + ((int64_t)((n * p.HL + iy) * p.WL + ix) * p.ldL + p.coffL + bch) * (int64_t)sizeof(T));
      }
      rs[i] = vs;
      rl[i] = vl;
    });
  };
  auto lds_store = [&](int stage) {
    char* sS = smem + stage * 2 * OPB;
    char* sL = sS + OPB;
    static_for<NP>([&](auto I) {
      constexpr int i = decltype(I)::value;
      const int r = rbase + RPP * i;
      u4_t v = rs[i];
      if (p.relu_S) v = relu16w<T>(v);
      *(u4_t*)(sS + r * LROW + chunk * 16) = v;
      *(u4_t*)(sL + r * LROW + chunk * 16) = rl[i];
    });
  };

  using acc_t = typename std::conditional<F16, f4_t, f16_t>::type;
  acc_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < (F16 ? 4 : 16); ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int stage) {
    const char* sS = smem + stage * 2 * OPB;
    const char* sL = sS + OPB;
    if constexpr (F16) {
      // ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(channel) block; lane 4q+p supplies the
      // address of block row q, columns 4p..4p+3; lane i receives column i of the 4 rows.
      const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
#pragma unroll
      for (int ks = 0; ks < BKP / 32; ++ks) {
      const char* sSk = sS + ks * 32 * LROW;
      const char* sLk = sL + ks * 32 * LROW;
      h8_t af[MT], bf[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int ch = wm * 64 + mt * 16 + 4 * pp;
        const char* a_lo = sSk + (8 * g + q) * LROW + ch * 2;
        const char* a_hi = sSk + (8 * g + 4 + q) * LROW + ch * 2;
        fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)a_lo);
        fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)a_hi);
        h4_t l4 = __builtin_bit_cast(h4_t, lo), h4 = __builtin_bit_cast(h4_t, hi);
        af[mt] = h8_t{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ch = wn * 64 + nt * 16 + 4 * pp;
        const char* b_lo = sLk + (8 * g + q) * LROW + ch * 2;
        const char* b_hi = sLk + (8 * g + 4 + q) * LROW + ch * 2;
        fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)b_lo);
        fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)b_hi);
        h4_t l4 = __builtin_bit_cast(h4_t, lo), h4 = __builtin_bit_cast(h4_t, hi);
        bf[nt] = h8_t{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
      }
    } else {
      const float* fS = (const float*)sS;
      const float* fL = (const float*)sL;
#pragma unroll 4
      for (int kk = 0; kk < 16; ++kk) {
        const int k = 2 * kk + (lane >> 5);
        float af[MT], bf[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = fS[k * 128 + wm * 64 + mt * 32 + (lane & 31)];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = fL[k * 128 + wn * 64 + nt * 32 + (lane & 31)];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
      }
    }
  };

  if (t_begin < t_end) {
    gload(t_begin);
    lds_store(0);
    __syncthreads();
    int stage = 0;
    for (int t = t_begin; t < t_end; ++t) {
      const bool more = (t + 1 < t_end);
      if (more) gload(t + 1);
      compute(stage);
      if (more) lds_store(stage ^ 1);
      __syncthreads();
      stage ^= 1;
    }
  }

  const int64_t ldw = (int64_t)16 * p.cb;
  constexpr int NR = F16 ? 4 : 16;
  // (the store mode is the same for every element: chosen once, not tested per element inside the unrolled loops)
  auto emit = [&](auto MODE) {
    constexpr int mode = decltype(MODE)::value;   // 0 partial tile, 1 direct accumulate, 2 float atomics
    float* base = mode == 0 ? p.part + (int64_t)ks * p.ca * ldw : p.dW;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        int row;
        if constexpr (F16) row = wm * 64 + mt * 16 + (lane >> 4) * 4 + r;
        else row = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          int c;
          if constexpr (F16) c = wn * 64 + nt * 16 + (lane & 15);
          else c = wn * 64 + nt * 32 + (lane & 31);
          const int64_t o = (int64_t)(a0 + row) * ldw + n0 + c;
          const float v = acc[mt][nt][r] * p.scale;
          if constexpr (mode == 0) base[o] = v;
          else if constexpr (mode == 1) base[o] += v;
          else atomicAdd(base + o, v);
        }
      }
  };
  if (p.part) { emit(std::integral_constant<int, 0>{}); return; }
  if constexpr (F16) {
    if (p.direct) {
      // One workgroup per output tile (no pixel-range split: the U-Net's innermost levels, 16.8 MB of gradient for <= 4 GFLOP):
      // dW += tile as 16-byte read-modify-writes of whole 256-byte rows through a wave-private LDS slab - a lane holds one COLUMN of
      // four rows per accumulator, so the direct form was 64 four-byte loads + 64 four-byte stores per lane on 64-byte segments
      // (d6 / d7 / u7 weight gradients: 28 - 37 us for 33.5 MB of traffic). Rows padded to 68 floats (two-way write conflicts: free).
      __syncthreads();                                  // every wave has read its last fragments: the staging buffers are free
      float* slab = (float*)smem + wave * (64 * 68);
      const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) slab[(mt * 16 + lq * 4 + r) * 68 + nt * 16 + lr] = acc[mt][nt][r] * p.scale;
      __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): the slab is wave-private, no workgroup barrier
      __builtin_amdgcn_wave_barrier();
      f4_t* rowp[16];
      f4_t cur[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {                    // all sixteen loads of dW first
        const int rr = i * 4 + lq;
        rowp[i] = (f4_t*)(p.dW + (int64_t)(a0 + wm * 64 + rr) * ldw + n0 + wn * 64 + lr * 4);
        cur[i] = *rowp[i];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f4_t v = *(const f4_t*)(slab + (i * 4 + lq) * 68 + lr * 4);
        *rowp[i] = cur[i] + v;
      }
      return;
    }
  }
  if (p.direct) emit(std::integral_constant<int, 1>{});
  else emit(std::integral_constant<int, 2>{});
}

// dW[i] += sum_k part[k][i] in a fixed order (deterministic): 64 float4 outputs per block, the splits divided among
// the four waves (contiguous k ranges), combined through LDS in wave order
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, int64_t count4, int split) {
  __shared__ f4_t red[3][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + tx;
  const int per = (split + 3) / 4, k0 = ty * per, k1 = min(split, k0 + per);
  f4_t s = f4_t{0.f, 0.f, 0.f, 0.f};
  if (i < count4)
    s = gi_ordered_sum_f4((const f4_t*)part + i, count4, k0, k1, s);   // eight loads in flight, the same order of additions
  if (ty > 0) red[ty - 1][tx] = s;
  __syncthreads();
  if (ty == 0 && i < count4) {
    s += red[0][tx];
    s += red[1][tx];
    s += red[2][tx];
    f4_t d = *(f4_t*)(dW + i * 4);
    d += s;
    *(f4_t*)(dW + i * 4) = d;
  }
}

int wgrad_split(int dtype, int n, int Hs, int Ws, int ca, int cb, int* tiles_per_split) {
  const int BKP = dtype == GI_F16 ? 64 : 32;
  const int P = n * Hs * Ws;
  const int tiles = (ca / 128) * (16 * cb / 128);
  const int ktiles = (P + BKP - 1) / BKP;
  const int target = gi_tune("GI_WGRAD_BLOCKS", 512);   // workgroups to aim for: 2 per CU in one wave; 1024 measured 10-28 % slower
  int split = (target + tiles - 1) / tiles;
  if (split > ktiles / 8) split = ktiles / 8;
  if (split < 1) split = 1;
  const int tps = (ktiles + split - 1) / split;
  if (tiles_per_split) *tiles_per_split = tps;
  return (ktiles + tps - 1) / tps;
}

template <typename T>
int run(hipStream_t st, const WgradArgs& a) {
  constexpr bool F16 = std::is_same<T, half_t>::value;
  constexpr int EPC = 16 / (int)sizeof(T);
  GI_REQUIRE(a.ca % 128 == 0, "wgrad: ca=%d must be a multiple of 128", a.ca);
  GI_REQUIRE(gi_is_pow2(a.cb) && a.cb >= EPC && (16 * a.cb) % 128 == 0, "wgrad: cb=%d must be a power of two >= 8", a.cb);
  GI_REQUIRE(a.ldS % EPC == 0 && a.coffS % EPC == 0 && a.ldL % EPC == 0 && a.coffL % EPC == 0,
             "wgrad: leading dims / channel offsets must be 16-byte aligned");
  WP p;
  p.S = (const char*)a.S; p.L = (const char*)a.L; p.dW = a.dW;
  p.P = a.n * a.Hs * a.Ws;
  p.Hs = a.Hs; p.Ws = a.Ws;
  const bool p2 = gi_is_pow2(a.Hs) && gi_is_pow2(a.Ws);
  p.lgHs = p2 ? gi_ilog2(a.Hs) : -1; p.lgWs = p2 ? gi_ilog2(a.Ws) : -1;
  p.HL = 2 * a.Hs; p.WL = 2 * a.Ws;
  p.ca = a.ca; p.ldS = a.ldS; p.coffS = a.coffS;
  p.cb = a.cb; p.lgcb = gi_ilog2(a.cb); p.ldL = a.ldL; p.coffL = a.coffL;
  p.relu_S = a.relu_S; p.scale = a.scale;
  GI_REQUIRE((int64_t)p.P * a.ldS < (1ll << 31) && (int64_t)a.n * p.HL * p.WL * a.ldL < (1ll << 31),
             "wgrad: tensor too large for 32-bit pixel math");
  int split = wgrad_split(F16 ? GI_F16 : GI_F32, a.n, a.Hs, a.Ws, a.ca, a.cb, &p.tiles_per_split);
  constexpr int BKP = F16 ? 64 : 32;
  const int64_t out_floats = (int64_t)a.ca * 16 * a.cb;
  p.part = nullptr;
  p.direct = split == 1 ? 1 : 0;
  if (split > 1 && a.scratch && a.scratch_bytes >= (int64_t)split * out_floats * 4) p.part = a.scratch;
  constexpr int LROW = F16 ? 256 + 32 : 512;
  constexpr int LDS = 2 * 2 * BKP * LROW;
  dim3 grid(a.ca / 128, 16 * a.cb / 128, split);
  static GiDevOnce attr_set;
  if (attr_set.first()) {
    GI_HIP(hipFuncSetAttribute((const void*)wgrad_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  }
  hipLaunchKernelGGL(wgrad_kernel<T>, grid, dim3(256), LDS, st, p);
  gi_note_kernel(F16 ? "wgrad<f16>" : "wgrad<f32>");
  GI_LAUNCH_CHECK();
  if (p.part) {
    const int64_t c4 = out_floats / 4;
    const int64_t nb = (c4 + 63) / 64;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const float*)p.part, a.dW, c4, split);
    GI_LAUNCH_CHECK();
  }
  return GI_OK;
}

}  // namespace

int op_wgrad2(hipStream_t st, const WgradArgs& a);                        // wgrad2.hip: LDS-DMA ring + shared halo (fp16)
int64_t op_wgrad2_scratch_bytes(int n, int Hs, int Ws, int ca, int cb);

int64_t op_wgrad_scratch_bytes(int dtype, int n, int Hs, int Ws, int ca, int cb) {
  if (ca % 128 != 0 || (16 * cb) % 128 != 0) return 0;
  const int split = wgrad_split(dtype, n, Hs, Ws, ca, cb, nullptr);
  int64_t b = split > 1 ? (int64_t)split * ca * 16 * cb * 4 : 0;
  if (dtype == GI_F16) { const int64_t b2 = op_wgrad2_scratch_bytes(n, Hs, Ws, ca, cb); if (b2 > b) b = b2; }
  return b;
}

int op_wgrad(hipStream_t st, int dtype, const WgradArgs& a) {
  const int use2 = gi_opt(GI_OPT_WGRAD2);   // GI_WGRAD2=0: always the register-staged kernel of this file
  if (dtype == GI_F16 && use2) {
    const int rc = op_wgrad2(st, a);
    if (rc != GI_ERR_UNSUPPORTED) return rc;
  }
  if (dtype == GI_F16) return run<half_t>(st, a);
  if (dtype == GI_F32) return run<float>(st, a);
  gi_set_error("wgrad: bad dtype %d", dtype);
  return GI_ERR_INVALID;
}
