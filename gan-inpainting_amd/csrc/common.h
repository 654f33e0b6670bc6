// Shared definitions for libganinpaint (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <type_traits>

#include "../../include/ganinpaint.h"

typedef _Float16 half_t;
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));  // one 16-byte chunk (native vector: SROA-friendly)
typedef float f16_t __attribute__((ext_vector_type(16)));

void gi_set_error(const char* fmt, ...);

// Run-time path options (api.hip): the value set through gi_set_option(), else the environment variable of the same name (read
// once per process), else the default. include/ganinpaint.h documents each; tests/test_options_gpu.py runs every alternative.
enum gi_opt_id {
  GI_OPT_IGEMM5 = 0, GI_OPT_IGEMM6, GI_OPT_IGEMM7, GI_OPT_IGEMM_FIXUP, GI_OPT_IGEMM_VARIANT, GI_OPT_BN_ACC, GI_OPT_FUSE_HEAD,
  GI_OPT_HEAD_FAST, GI_OPT_BN_BWD_FUSE, GI_OPT_BN_BWD_SMALL, GI_OPT_WGRAD2, GI_OPT_WGRAD3, GI_OPT_IGEMM8, GI_OPT_BN_FOLD, GI_OPT_C1_FUSED, GI_OPT_WGRAD_STREAM, GI_OPT_MASK_BITS, GI_OPT_C1W_FUSE, GI_OPT_IGEMM7_WAVES, GI_OPT_COUNT
};
int gi_opt(int id);
// name of the GEMM / weight-gradient kernel a dispatcher has just launched (gi_debug_last_kernel: tests assert which kernel
// served a shape)
void gi_note_kernel(const char* name);
void gi_note_fold();   // a GEMM launch took an IgemmFold (gi_debug_fold_count)
// tuning constants: fixed in the shipped library, read from the environment only by the ablation build (build.sh -DGI_ABLATION)
#ifdef GI_ABLATION
#include <stdlib.h>
static inline int gi_tune(const char* env, int dflt) { const char* e = getenv(env); return e ? atoi(e) : dflt; }
#else
#define gi_tune(env, dflt) (dflt)
#endif

// compile-time unrolled loop: f(std::integral_constant<int, i>) for i in [0, N). Register arrays
// indexed through it never fall back to scratch memory.
template <int N, typename F>
__host__ __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// ---- pieces shared by the GEMM epilogues (igemm3 / 5 / 6 / 7 / 8) -----------------------------------------------------------
// The output activation is the same for every element of a launch. Written as `act == RELU ? .. : act == LRELU ? ..` inside the
// unrolled accumulator loops, hipcc kept the test PER ELEMENT as scalar compares and branches (round 3's igemm8<1,true>: 256 s_cmp
// + 650 s_cbranch per wave in the epilogue, three of them taken per element on the BatchNorm layers' act = none path). gi_with_act
// runs the body once per launch with the activation as a compile-time constant instead.
template <int ACT>
__device__ __forceinline__ float gi_act_c(float v) {
  if constexpr (ACT == GI_ACT_RELU) return v > 0.f ? v : 0.f;
  else if constexpr (ACT == GI_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
  else return v;
}
template <typename F>
__device__ __forceinline__ void gi_with_act(int act, F&& f) {
  if (act == GI_ACT_NONE) f(std::integral_constant<int, GI_ACT_NONE>{});
  else if (act == GI_ACT_RELU) f(std::integral_constant<int, GI_ACT_RELU>{});
  else f(std::integral_constant<int, GI_ACT_LRELU>{});
}
// the same for a launch-constant flag (statistics wanted, bias present): the accumulator loops drop the work of the unused side
template <typename F>
__device__ __forceinline__ void gi_with_bool(bool b, F&& f) {
  if (b) f(std::true_type{});
  else f(std::false_type{});
}
// Sum over the 16 lanes of a DPP row (lane & 15 = the pixel row of a 16x16 MFMA tile), every lane receives the total. The same
// additions in the same order as the xor butterfly `v += __shfl_xor(v, 1 / 2 / 4 / 8)` it replaces (quad_perm = xor 1, xor 2; after
// those a quad holds one value, so row_half_mirror pairs quads like xor 4 and row_mirror pairs the 8-lane halves like xor 8) -
// bit-identical results - as four v_add_f32 with DPP operands instead of four ds_bpermute_b32 + waits + adds per value.
template <int CTRL>
__device__ __forceinline__ float gi_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float gi_row16_sum(float v) {
  v += gi_dpp<0xB1>(v);    // quad_perm:[1,0,3,2]
  v += gi_dpp<0x4E>(v);    // quad_perm:[2,3,0,1]
  v += gi_dpp<0x141>(v);   // row_half_mirror
  v += gi_dpp<0x140>(v);   // row_mirror
  return v;
}

// s += part[k0] + part[k0 + 1] + ... + part[k1 - 1] (elements `stride` f4_t apart), in that order, with EIGHT loads in flight: written
// as `for (k) s += part[k]`, hipcc waits for every load before it issues the next (the adds are a dependent chain and the loop is
// not unrolled), so a 16-way fixed-order sum was 16 dependent L2 / HBM round trips - the weight-gradient reduce launches of a batch
// ran 19 us each for 33 MB that way. Same additions in the same order: bit-identical results.
__device__ __forceinline__ f4_t gi_ordered_sum_f4(const f4_t* part, int64_t stride, int k0, int k1, f4_t s) {
  int k = k0;
  for (; k + 8 <= k1; k += 8) {
    f4_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = part[(int64_t)(k + j) * stride];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  // the last 1 .. 7 in groups of four, two and one (k0 / k1 are wave-uniform: plain branches; a first version loaded eight from
  // clamped indices and read the last partial up to seven times - conv4's 8-way sum, two per wave, moved four times the bytes)
  if (k + 4 <= k1) {
    f4_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = part[(int64_t)(k + j) * stride];
#pragma unroll
    for (int j = 0; j < 4; ++j) s += v[j];
    k += 4;
  }
  if (k + 2 <= k1) {
    const f4_t v0 = part[(int64_t)k * stride], v1 = part[(int64_t)(k + 1) * stride];
    s += v0;
    s += v1;
    k += 2;
  }
  if (k < k1) s += part[(int64_t)k * stride];
  return s;
}

// "done once per DEVICE" flag: hipFuncSetAttribute belongs to the device it is called on, so a process that drives several GPUs
// (not the data-parallel layout, which is one process per GPU) must repeat it per device
struct GiDevOnce {
  unsigned long long mask = 0;
  bool first() {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long b = 1ull << (d & 63);
    const unsigned long long old = __atomic_fetch_or(&mask, b, __ATOMIC_RELAXED);
    return (old & b) == 0;
  }
};

#define GI_HIP(expr)                                                                  \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      gi_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return GI_ERR_HIP;                                                              \
    }                                                                                 \
  } while (0)

#define GI_LAUNCH_CHECK()                                                             \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess) {                                                           \
      gi_set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
      return GI_ERR_HIP;                                                              \
    }                                                                                 \
  } while (0)

#define GI_REQUIRE(cond, ...)                                                         \
  do {                                                                                \
    if (!(cond)) {                                                                    \
      gi_set_error(__VA_ARGS__);                                                      \
      return GI_ERR_INVALID;                                                          \
    }                                                                                 \
  } while (0)

#define GI_TRY(expr)                \
  do {                              \
    int _s = (expr);                \
    if (_s != GI_OK) return _s;     \
  } while (0)

struct gi_ctx {
  int device;
  hipStream_t stream;
  unsigned* tickets = nullptr;   // IgemmArgs::tickets for the single-layer entry points (allocated on first use)
};

static inline size_t gi_dtype_size(int dtype) { return dtype == GI_F16 ? 2 : 4; }
static inline int64_t gi_align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
static inline int gi_ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static inline bool gi_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ------------------------------------------------------------------------------------------
// kernel launch wrappers (ops_*.hip). All tensors NHWC, element type T selected by dtype.
// ------------------------------------------------------------------------------------------

// Implicit-GEMM geometry shared by the strided conv ("CONV": gathers a 4x4 window of the
// large-resolution tensor) and the transposed conv ("PHASE": four 2x2 sub-pixel convolutions of
// the small-resolution tensor). Rows of the GEMM are always small-resolution pixels (n,y,x).
struct IgemmFold;       // below BnAccArgs
struct IgemmArgs {
  const void* in;      // CONV: large tensor (n,2Hs,2Ws,cin) ; PHASE: small tensor (n,Hs,Ws,cin)
  const void* w;       // CONV: T [cout][16*cin] ; PHASE: T [4][cout][4*cin]
  void* out;           // CONV: small tensor (n,Hs,Ws,cout) ; PHASE: large tensor (n,2Hs,2Ws,cout)
  const float* bias;   // [cout] or null
  float* partials;     // [tiles][2][cout] column sum / sum of squares, or null
  float* ws;           // fp32 split-K scratch (>= out pixels * cout floats) or null
  int64_t ws_bytes;
  int n, Hs, Ws;       // small-resolution geometry
  int cin, ldin, coffin;
  int cout, ldout, coffout;
  int relu_in;         // relu applied to `in` while staging
  int relu_cend;       // 0: all input channels; > 0: only channels [0, relu_cend) need the relu (hint; relu is
                       // idempotent, kernels may apply it to more channels)
  int act_out;         // gi_act on the result
  int force_splitk;    // 0 = heuristic, >0 forces that split
  unsigned* tickets;   // GI_IGEMM_TICKETS counters, zero before the first launch (every launch leaves them zero), or null:
                       // split-K layers then reduce inside the GEMM kernel (last arriver per tile) instead of a finish launch
  int ntiles_out;      // (returned) number of partial rows written
  // optional fused activation backward on the result (input-gradient GEMMs): out *= (mask > 0 ? 1 : mask_slope), with
  // mask the forward activation of the producing layer, laid out like out (pixels x channels, ld = ldmask). Kernels
  // that implement it set mask_applied = 1; otherwise the caller runs the separate pass.
  const void* mask; int ldmask, coffmask; float mask_slope;
  const void* add; int ldadd, coffadd;   // with mask: a second gradient added where mask > 0 before the slope (the skip
                                         // half of a U-Net concat gradient, which passed the parent's in-place ReLU)
  int mask_applied;
  // with mask: one 64-bit sign word per output pixel, bit c = [mask[p][coffmask + c] > 0] for the 64 channels from coffmask
  // (op_c1_gather writes them with the tensor), or null. A kernel that takes them (igemm8's dual-px tiles, cout = 64) never reads
  // `mask`: 8 bytes per pixel instead of 128, the slope applied to the fp32 accumulators (one rounding instead of two).
  const unsigned long long* mask_bits;
  // with mask_bits, the result being the gradient dz1 at the output of a single-channel 4x4 / s2 convolution (the networks' first
  // layers, 64 channels): that layer's weight gradient from the tile while it is in LDS, dW1[c][tap] = sum_p img[2Y-1+ky, 2X-1+kx] *
  // dz1[p][c] (img: the layer's fp32 input, n x 4Hs x 4Ws). Every workgroup writes its 64 x 16 sums * c1w_scale to
  // c1w_part[workgroup][1024] (>= c1w_part_floats); the caller adds the c1w_blocks rows in a fixed order (op_c1_wgrad_reduce).
  // c1w_skip_out: nothing else reads the result - it is not stored. Kernels that implement it set c1w_applied.
  const float* c1w_img; float* c1w_part; int64_t c1w_part_floats; float c1w_scale; int c1w_skip_out;
  int c1w_applied, c1w_blocks;
  // column statistics without partial rows (stat_acc.h): every tile adds its column sum / sum of squares into the exact
  // per-channel accumulator block stat_acc (layout: stat_acc.h; zeroed by the caller; stat_reps replicas, a power of two,
  // tile t adds to replica t mod stat_reps). stat_pg > 0: the GEMM rows (small-grid pixels) are two consecutive BatchNorm
  // populations of stat_pg rows each (a multiple of 256), rows >= stat_pg add to group 1. Takes precedence over `partials`.
  unsigned long long* stat_acc; int stat_pg; int stat_reps;
  // optional fused BatchNorm-backward reduction (input-gradient GEMMs whose result g is the gradient w.r.t. the output of
  // BatchNorm + LeakyReLU(bwd_slope) of a layer with the raw convolution output bwd_x, laid out like out with ld = bwd_ldx):
  // every tile adds sum dz and sum dz * xhat, dz = g * (fma(x, scale, shift) > 0 ? 1 : slope), xhat = (x - mean) * inv, to the
  // accumulator block bwd_acc (quantities 0 / 1, stat_acc.h; replicas / populations as stat_reps / stat_pg, here counted in
  // OUTPUT pixels). Kernels that implement it set bwd_applied = 1: the caller then skips the reduce pass of op_act_bn_bwd.
  // bwd_c > 0: only the output columns [bwd_c0, bwd_c0 + bwd_c) carry that gradient (the decoder half of a U-Net concat gradient;
  // bwd_c0 a multiple of 128): x, the vectors and the accumulators are indexed by column - bwd_c0 and hold bwd_c channels
  const void* bwd_x; int bwd_ldx;
  const float* bwd_scale; const float* bwd_shift; const float* bwd_mean; const float* bwd_inv; int bwd_stride;
  float bwd_slope; int bwd_c0, bwd_c;
  unsigned long long* bwd_acc; int bwd_reps; int64_t bwd_pg;
  int bwd_applied;
  // mode 2 (3x3 / s1) only: store the 2x2 / stride-2 max pool of the activated output instead of the output itself (`out` is then the
  // (n, Hs/2, Ws/2, cout) tensor). Kernels that implement it set pool_applied = 1; otherwise the caller pools in a separate pass.
  int pool2, pool_applied;
  int stat_used;   // (returned) 1: the statistics went to stat_acc; 0: to `partials` (split-K layers: their finish pass
                   // has few rows per block, the partial rows + finalize launch are cheaper there than 4 atomics per channel)
  // optional (with stat_acc): the normalisation pass that would follow, folded into this launch (IgemmFold below). A kernel that
  // takes it sets fold_applied = 1 and the caller launches no bn_apply for the layer.
  const IgemmFold* fold; int fold_applied;
};
constexpr int GI_IGEMM_TICKETS = 1024;
int op_igemm(hipStream_t st, int dtype, int phase_mode, IgemmArgs& a);

struct WgradArgs {
  const void* S;   // small tensor (n,Hs,Ws,ca), ld=ldS, channel offset coffS
  const void* L;   // large tensor (n,2Hs,2Ws,cb), ld=ldL, channel offset coffL
  float* dW;       // fp32 [ca][16][cb], accumulated (+=)
  int n, Hs, Ws;
  int ca, ldS, coffS;
  int cb, ldL, coffL;
  int relu_S;
  float scale;
  float* scratch;          // optional: op_wgrad_scratch_bytes() bytes; the pixel-range splits then write partial tiles
  int64_t scratch_bytes;   // with plain stores and a fixed-order pass adds them to dW (deterministic); else atomics
};
int op_wgrad(hipStream_t st, int dtype, const WgradArgs& a);
int64_t op_wgrad_scratch_bytes(int dtype, int n, int Hs, int Ws, int ca, int cb);

// single-channel-side kernels (generator first conv / last transposed conv, discriminator first
// conv): weights fp32 [c][16] (= [a][ky][kx][b] with b == 1)
// out[p][c] = act(sum_tap img[n,2y-1+ky,2x-1+kx] * w[c][tap])      img fp32 (n,2Hs,2Ws)
// bits != null: the kernels that can also write one 64-bit sign word per pixel (bit c = [out[p][c] > 0], c = 64 only) and say so
// in *bits_written
int op_c1_gather(hipStream_t st, int dtype, const float* img, const float* w, void* out, int n, int Hs,
                 int Ws, int c, int ldout, int coffout, int act_out, float in_scale, const float* bias = nullptr,
                 unsigned long long* bits = nullptr, int* bits_written = nullptr);
// img[n,Y,X] = post( bias + sum_{c,tap} relu?(X[p][c]) * w[c][tap] ) (overlap-add of the 4x4 taps)
// post: 0 none, 1 tanh. img fp32 (n,2Hs,2Ws); out_scale multiplies the result (loss-scale removal)
// col_scratch (fp16 path): >= n*Hs*Ws*16 halves; null selects the register-reduction kernel
// Fused producer for the upper half of a c1 kernel's input channels: relu(fma(x2, scale, shift)) of a dense raw tensor
// (the BatchNorm + ReLU of the last decoder level), instead of reading that half from X.
struct BnAccArgs;
struct C1Affine { const void* x2; int ld2; const float* scale; const float* shift;
                  const BnAccArgs* bn = nullptr; };   // non-null (op_c1_scatter): scale / shift are DERIVED by the kernel from the exact accumulators and stored there
// the 4-channel head of a narrow generator (UnetGenerator(1, 4, 7, ngf=32), fp16, 128 input channels): col GEMM with 64 rows + overlap-add
// with bias and tanh into (n,4,2Hs,2Ws) fp32 (out2: a second copy, may be null); input gradient of the head from the (n,4,2Hs,2Ws) gradient
bool op_c1_head4_ok(int dtype, int c, int out_c, int Ws, int ldx, int coffx);
int64_t op_c1_head4_col_bytes(int n, int Hs, int Ws);
int op_c1_head4_forward(hipStream_t st, const void* X, const float* w, const float* bias, float* out, float* out2, int n, int Hs, int Ws, int ldx,
                        int coffx, int relu_in, void* col_scratch);
int op_c1_head4_dgrad(hipStream_t st, const float* g, const float* w, void* out, int n, int Hs, int Ws, int ldout, int coffout);
// dW[i] += sum over `blocks` rows of part[block][count], fixed order (the second stage of op_c1_wgrad and of IgemmArgs::c1w_part)
// scratch (>= 64 * count floats, or null): a first stage over row ranges when there are many rows
int op_c1_wgrad_reduce(hipStream_t st, const float* part, float* dW, int count, int blocks, float* scratch = nullptr, int64_t scratch_floats = 0);
bool op_c1_affine_ok(int dtype, int c, int Ws, int ldx, int coffx);
int op_c1_scatter(hipStream_t st, int dtype, const void* X, const float* w, const float* bias, float* img,
                  int n, int Hs, int Ws, int c, int ldx, int coffx, int relu_in, int post, float out_scale,
                  void* col_scratch, float* img2 = nullptr, const C1Affine* aff = nullptr);
// dW[c][tap] += scale * sum_p relu?(X[p][c]) * img[n,2y-1+ky,2x-1+kx]
// scratch (optional, >= 512 * c * 16 floats): the workgroups' partial sums go there and are added in a fixed order (bit-reproducible)
int op_c1_wgrad(hipStream_t st, int dtype, const void* X, const float* img, float* dW, int n, int Hs, int Ws,
                int c, int ldx, int coffx, int relu_in, float scale, float img_scale, const C1Affine* aff = nullptr,
                float* scratch = nullptr, int64_t scratch_floats = 0);

// BatchNorm helpers -------------------------------------------------------------------------
// partials [rows][2][c] -> scale/shift (y = x*scale + shift), saved mean / invstd; train mode also
// updates running stats (momentum 0.1, unbiased variance), eval mode uses the running stats.
int op_bn_finalize(hipStream_t st, const float* partials, int rows, int c, int64_t count, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, float* scale, float* shift,
                   float* save_mean, float* save_invstd, int train, float momentum, float eps, int groups = 1,
                   int64_t part_stride = 0, int out_stride = 0);
// The same two steps from the exact accumulators the GEMM epilogues added to (IgemmArgs::stat_acc, stat_acc.h), train
// mode only. acc: [groups][c][4] 64-bit words; count = pixels per population; scale / shift / save_* as above, the
// populations' vectors out_stride floats apart.
struct BnAccArgs {
  const unsigned long long* acc; int reps;   // accumulator block (stat_acc.h) with `reps` replicas
  const float* gamma; const float* beta;
  float* running_mean; float* running_var;
  float* scale; float* shift; float* save_mean; float* save_invstd;
  int64_t count;
  float momentum, eps;
  int groups, out_stride;
  unsigned long long* zero_next; int zero_words;   // optional: a region the pass clears for the layer's next use (ping-pong)
};
int op_bn_finalize_acc(hipStream_t st, int c, const BnAccArgs& b);
// BatchNorm (train mode, one population) + activation + dropout of a small layer inside the GEMM that produces it (igemm7.hip): the
// workgroup that completes the LAST tile of a channel column (all M tiles and sub-pixel phases of an N tile; a second ticket per
// column, no grid barrier, no spin) derives that column's scale / shift from the exact accumulators, moves the running statistics,
// saves the backward's vectors, clears the column's share of the layer's other accumulator region and writes act(fma(x, scale,
// shift)) [* dropout] of the column into dst - the same fp32 expressions, the same fp16 rounding and the same dropout hash as
// bn_apply_kernel, so results are bit-identical to the separate pass. The raw tensor is still written (the backward reads it).
struct IgemmFold {
  BnAccArgs bn;                      // groups == 1; acc == IgemmArgs::stat_acc
  void* dst; int lddst, coffdst;     // (out pixels, lddst) + coffdst: the concat-buffer half
  int act;                           // gi_act
  uint8_t* drop_mask; float drop_scale; uint64_t drop_seed; float drop_p;   // as op_bn_apply_acc
};
// finalize + normalise + activation (+ dropout) in ONE pass: no launch between the GEMM and this one. drop_mask non-null
// with drop_p > 0: the keep-mask is drawn in the pass (seed drop_seed) and stored there; drop_p == 0: the mask is read.
// side (optional): an unrelated device copy of side_bytes (a multiple of 16) carried by the same launch: the pass is bound by its
// launch and first-load latencies, not by bandwidth, so up to a few MB ride along for free (the generator's saved input)
int op_bn_apply_acc(hipStream_t st, int dtype, const void* x, void* y, int64_t pixels, int c, int ldy, int coffy, int act,
                    uint8_t* drop_mask, float drop_scale, uint64_t drop_seed, float drop_p, const BnAccArgs& b,
                    const void* side_src = nullptr, void* side_dst = nullptr, int64_t side_bytes = 0);
// pg in (0, pixels): two BatchNorm populations, pixels >= pg use scale/shift + gstride
int op_bn_apply(hipStream_t st, int dtype, const void* x, void* y, int64_t pixels, int c, int ldy, int coffy,
                const float* scale, const float* shift, int act, const uint8_t* drop_mask, float drop_scale,
                int64_t pg = 0, int gstride = 0);
// column sum / sumsq of a dense (pixels,c) tensor -> partials [blocks][2][c]; returns rows in *rows_out
int op_col_stats(hipStream_t st, int dtype, const void* x, int64_t pixels, int c, float* partials, int* rows_out);

// backward through [drop] -> act -> [BN]:  dz = (g1 + g2*[y>0]) * slope(y) * drop_scale
//   slope(y): act==LRELU: y>0?1:0.2 ; RELU: [y>0] ; NONE: 1
// pass 1 (has_bn): partial sums of dz and dz*xhat ; pass 2: dx = scale_bn*(dz - mean_dz - xhat*mean_dzxhat)
struct ActBnBwdArgs {
  const void* g1; int ldg1, coffg1;   // may be null
  const void* g2; int ldg2, coffg2;   // may be null (masked by [y>0])
  const void* y;  int ldy, coffy;     // saved activation output (for the masks)
  const void* x;                      // raw pre-BN tensor, dense (ld=c); null when !has_bn
  void* dx;                           // dense (ld=c) result
  int64_t pixels; int c;
  int act; float drop_scale;
  int has_bn;
  int eval_bn;                        // BatchNorm ran on running statistics (frozen eval net): dx = gamma*inv*dz, no batch terms
  const float* gamma; const float* save_mean; const float* save_invstd;
  float* dgamma; float* dbeta;        // accumulated (+=) with 1/loss_scale when non-null
  float inv_loss_scale;
  float* partials;                    // scratch [blocks][2][c]
  float* sums;                        // scratch [groups][8][c]
  const float* fwd_scale; const float* fwd_shift;   // non-null (BatchNorm layers without dropout): the forward's affine map;
                                      // sign(act input) = sign(fma(x, scale, shift)) replaces the read of y
  int groups;                         // 2: the tensor is two consecutive BatchNorm populations (save_mean / save_invstd of
  int stat_stride;                    //    the second at +stat_stride floats), reduced separately in the same launches
  // non-null (has_bn): exact accumulator block (stat_acc.h, zeroed, acc_reps replicas): the reduce pass adds into it and the
  // apply pass derives its coefficients itself - two launches instead of three; the apply pass then clears zero_next.
  unsigned long long* acc; int acc_reps; unsigned long long* zero_next; int zero_words;
  int reduce_done;                    // 1: the sums are already in acc (IgemmArgs::bwd_acc, added by the producing GEMM's epilogue)
};
int op_act_bn_bwd(hipStream_t st, int dtype, const ActBnBwdArgs& a);
int op_bwd_rows_per_block(int64_t pixels);
// InstanceNorm2d(affine=False, track_running_stats=False) (get_norm_layer('instance'), networks.py:38-40): per (image, channel)
// statistics over hw pixels; x dense (n*hw, c); stats [n][c][2] = mean, inv (kept for the backward)
int op_in_forward(hipStream_t st, int dtype, const void* x, void* y, int n, int hw, int c, int ldy, int coffy, int act,
                  const uint8_t* drop_mask, float drop_scale, float eps, float* stats);
// backward through [dropout] -> activation -> InstanceNorm (ActBnBwdArgs: g1, g2, y, x, dx, pixels, c, act, drop_scale)
int op_act_in_bwd(hipStream_t st, int dtype, const ActBnBwdArgs& a, int n, int hw, const float* stats);
// dbias[ch] += scale * sum_pixels dz[p][ch]; partials: scratch for the column pass ((pixels/32 + 8) * 2 * c floats)
int op_bias_grad(hipStream_t st, int dtype, const void* dz, int64_t pixels, int c, float scale, float* dbias, float* partials);   // rows one reduce workgroup covers (groups = 2 needs pixels/2 to be a multiple)

// discriminator head: conv(512->1,k4,s1,p0) + Flatten + Linear(P,1) [+ sigmoid]
struct HeadArgs {
  const void* a4;      // (n,Hh,Wh,c) T dense
  const float* w5;     // [16][c]  (= [1][ky][kx][c])
  const float* wl;     // [P]
  const float* bl;     // [1]
  float* h;            // (n,P) conv output (saved)
  float* out;          // (n,1)
  int n, Hh, Wh, c, sigmoid;
  float* out2 = nullptr;   // optional second copy of out (the caller's y)
  // non-null: a4 is the RAW conv4 output and the head applies LeakyReLU(fma(x, scale4, shift4)) itself (BatchNorm
  // population of image i: i / n_per_group, its vectors at + population * gstride floats); op_head_affine_ok()
  const float* scale4 = nullptr; const float* shift4 = nullptr; int n_per_group = 0, gstride = 0;
  float* tbuf = nullptr; int64_t tbuf_bytes = 0;   // optional (n, Hh*Wh, 16) fp32 scratch: lets large maps be sliced over workgroups
  const BnAccArgs* bn = nullptr;   // with scale4: the vectors are DERIVED by the head's first kernel from conv4's exact accumulators and
                                   // stored at scale4 / shift4 (+ the backward's vectors): no finalize launch in front of the head
};
bool op_head_affine_ok(int dtype, int c);
int op_head_forward(hipStream_t st, int dtype, const HeadArgs& a);
struct HeadBwdArgs {
  const void* a4; const float* w5; const float* wl; const float* h; const float* out;
  const float* dy;     // (n,1) grad wrt out
  void* da4;           // (n,Hh,Wh,c) T, scaled by loss_scale
  float* dw5; float* dwl; float* dbl;   // accumulated, may be null (frozen)
  float* dh;           // scratch (n,P)
  int n, Hh, Wh, c, sigmoid;
  float loss_scale;
  float* scratch = nullptr; int64_t scratch_bytes = 0;   // weight-gradient partials (op_head_scratch_bytes)
  const float* scale4 = nullptr; const float* shift4 = nullptr; int n_per_group = 0, gstride = 0;   // as in HeadArgs
  // BatchNorm-backward reduction of the layer in front of the head (conv4), fused into the input-gradient blocks: with x = bwd_x (the
  // layer's raw output, dense, c channels), dz = da4 * (fma(x, scale, shift) > 0 ? 1 : bwd_slope), xhat = (x - mean) * inv, every
  // block adds sum dz and sum dz * xhat per channel to the exact accumulators bwd_acc (stat_acc.h; population of image nn =
  // nn / bwd_n_per_group, vectors bwd_stride floats apart). *bwd_applied = 1 when the kernel did it.
  const void* bwd_x = nullptr;
  const float* bwd_scale = nullptr; const float* bwd_shift = nullptr; const float* bwd_mean = nullptr; const float* bwd_inv = nullptr;
  int bwd_stride = 0, bwd_n_per_group = 0, bwd_reps = 1; float bwd_slope = 1.f;
  unsigned long long* bwd_acc = nullptr;
  int* bwd_applied = nullptr;
};
// (kernel argument of the fused reduction)
struct HeadBwdFuse {
  const char* x; const float* scale; const float* shift; const float* mean; const float* inv;
  int stride, n_per_group, reps; float slope;
  unsigned long long* acc;
};
int64_t op_head_scratch_bytes(int max_n, int Hh, int Wh);
int op_head_backward(hipStream_t st, int dtype, const HeadBwdArgs& a);

// weight packing / conversions
int op_pack_weights(hipStream_t st, int dtype, const float* w, int ca, int cb, void* w_packed, void* w_phase);
// the same for up to 16 layers in ONE launch (packed / phase may be null per layer)
struct PackJob { const float* w; void* packed; void* phase; int ca, cb, tile0, scale_on_b; const float* scale; };   // scale: per a (or b) factor, or null
struct PackJobs { PackJob j[16]; int n; };
int op_pack_weights_batch(hipStream_t st, int dtype, PackJobs& P);
int op_convert(hipStream_t st, int dtype, const float* src, void* dst, int64_t count);
int op_convert_back(hipStream_t st, int dtype, const void* src, float* dst, int64_t count);
int op_tanh_bwd(hipStream_t st, const float* dy, const float* y, float* dx, int64_t count, float scale);
int op_fill_dropout(hipStream_t st, uint8_t* mask_nhwc, int64_t count, uint64_t seed, float p);
int op_mask_nchw_to_nhwc(hipStream_t st, const uint8_t* src, uint8_t* dst, int n, int c, int hw, int to_nhwc);

// gradient-penalty helpers (elementwise.hip)
int op_mul_slope(hipStream_t st, int dtype, const void* tin, const void* a, void* tout, int64_t count);
int op_bn_tangent_inject(hipStream_t st, int dtype, const void* dta, const void* y, const void* tx, const void* x, void* dxp,
                         int64_t pixels, int c, const float* gamma, const float* mean, const float* inv, float* dgamma,
                         float* partials, float* sums);
int op_gp_direction(hipStream_t st, const float* g, int n, int64_t hw, float lam, float* sumsq, float* v, float* penalty);
