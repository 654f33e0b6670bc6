// C-ABI glue: error reporting, context, single-layer entry points (see include/ganinpaint.h).
#include <string.h>

#include <stdlib.h>

#include "common.h"
#include "stat_acc.h"

static thread_local char g_err[1024] = "";

void gi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
// state 0: unread, 1: from default / environment, 2: set by the caller. value / state are relaxed atomics: two host threads that
// drive separate contexts may read (and lazily initialise, to the same value) an option at the same time
struct OptDesc { const char* name; int dflt; int value; int state; };
inline int opt_load(const int& v) { return __atomic_load_n(&v, __ATOMIC_RELAXED); }
inline void opt_store(int& v, int x) { __atomic_store_n(&v, x, __ATOMIC_RELAXED); }
OptDesc g_opts[GI_OPT_COUNT] = {
  {"GI_IGEMM5", 7, 0, 0}, {"GI_IGEMM6", 1, 0, 0}, {"GI_IGEMM7", 1, 0, 0}, {"GI_IGEMM_FIXUP", 1, 0, 0}, {"GI_IGEMM_VARIANT", 3, 0, 0},
  {"GI_BN_ACC", 1, 0, 0}, {"GI_FUSE_HEAD", 1, 0, 0}, {"GI_HEAD_FAST", 1, 0, 0}, {"GI_BN_BWD_FUSE", 1, 0, 0}, {"GI_BN_BWD_SMALL", 512, 0, 0},
  {"GI_WGRAD2", 1, 0, 0}, {"GI_WGRAD3", 1, 0, 0}, {"GI_IGEMM8", 1, 0, 0}, {"GI_BN_FOLD", 0, 0, 0}, {"GI_C1_FUSED", 1, 0, 0},
  {"GI_WGRAD_STREAM", 1, 0, 0}, {"GI_MASK_BITS", 1, 0, 0}, {"GI_C1W_FUSE", 1, 0, 0}, {"GI_IGEMM7_WAVES", 8, 0, 0},
};
thread_local const char* g_last_kernel = "";   // per host thread: gi_debug_last_kernel never reports another thread's launch
int g_fold_count = 0;
}  // namespace
void gi_note_fold() { __atomic_fetch_add(&g_fold_count, 1, __ATOMIC_RELAXED); }

int gi_opt(int id) {
  OptDesc& o = g_opts[id];
  if (opt_load(o.state) == 0) {
    const char* e = getenv(o.name);
    opt_store(o.value, e ? atoi(e) : o.dflt);
    __atomic_store_n(&o.state, 1, __ATOMIC_RELEASE);
  }
  return opt_load(o.value);
}
void gi_note_kernel(const char* name) { g_last_kernel = name; }

extern "C" {

int gi_set_option(const char* name, int value) {
  GI_REQUIRE(name, "set_option: null name");
  for (int i = 0; i < GI_OPT_COUNT; ++i)
    if (strcmp(name, g_opts[i].name) == 0) {
      if (value < 0) opt_store(g_opts[i].state, 0);      // back to the environment / default
      else { opt_store(g_opts[i].value, value); __atomic_store_n(&g_opts[i].state, 2, __ATOMIC_RELEASE); }
      return GI_OK;
    }
  gi_set_error("set_option: unknown option '%s'", name);
  return GI_ERR_INVALID;
}
int gi_get_option(const char* name, int* value) {
  GI_REQUIRE(name && value, "get_option: null argument");
  for (int i = 0; i < GI_OPT_COUNT; ++i)
    if (strcmp(name, g_opts[i].name) == 0) { *value = gi_opt(i); return GI_OK; }
  gi_set_error("get_option: unknown option '%s'", name);
  return GI_ERR_INVALID;
}
const char* gi_debug_last_kernel(void) { return g_last_kernel; }
int gi_debug_fold_count(void) { return __atomic_load_n(&g_fold_count, __ATOMIC_RELAXED); }

const char* gi_last_error(void) { return g_err; }
int gi_version(void) { return 100; }

int gi_ctx_create(int device_id, void* hip_stream, gi_ctx** out) {
  GI_REQUIRE(out != nullptr, "ctx_create: out is null");
  int count = 0;
  GI_HIP(hipGetDeviceCount(&count));
  GI_REQUIRE(device_id >= 0 && device_id < count, "ctx_create: device %d not present (%d visible)", device_id, count);
  GI_HIP(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  GI_HIP(hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    gi_set_error("ctx_create: device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
    return GI_ERR_UNSUPPORTED;
  }
  gi_ctx* c = new gi_ctx();
  c->device = device_id;
  c->stream = (hipStream_t)hip_stream;
  *out = c;
  return GI_OK;
}
int gi_ctx_destroy(gi_ctx* ctx) {
  if (ctx && ctx->tickets) (void)hipFree(ctx->tickets);
  delete ctx;
  return GI_OK;
}
int gi_ctx_sync(gi_ctx* ctx) {
  GI_REQUIRE(ctx, "ctx_sync: null");
  GI_HIP(hipStreamSynchronize(ctx->stream));
  return GI_OK;
}

static unsigned* ctx_tickets(gi_ctx* ctx) {
  if (!ctx->tickets) {
    if (hipMalloc((void**)&ctx->tickets, GI_IGEMM_TICKETS * 4) != hipSuccess) { ctx->tickets = nullptr; return nullptr; }
    if (hipMemset(ctx->tickets, 0, GI_IGEMM_TICKETS * 4) != hipSuccess) { (void)hipFree(ctx->tickets); ctx->tickets = nullptr; }
  }
  return ctx->tickets;
}

int gi_conv_s2_forward(gi_ctx* ctx, int dtype, const void* in, const void* w_packed, void* out, int n, int H, int W, int cb, int ldin,
                       int ca, int ldout, int relu_in, int act_out, float* ws, int64_t ws_bytes) {
  GI_REQUIRE(ctx && in && w_packed && out, "conv_s2_forward: null pointer");
  GI_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv_s2_forward: H=%d W=%d must be even", H, W);
  IgemmArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in; a.w = w_packed; a.out = out; a.ws = ws; a.ws_bytes = ws_bytes;
  a.tickets = ws ? ctx_tickets(ctx) : nullptr;
  a.n = n; a.Hs = H / 2; a.Ws = W / 2;
  a.cin = cb; a.ldin = ldin; a.cout = ca; a.ldout = ldout;
  a.relu_in = relu_in; a.act_out = act_out;
  return op_igemm(ctx->stream, dtype, 0, a);
}

int gi_convT_s2_forward(gi_ctx* ctx, int dtype, const void* in, const void* w_phase, void* out, int n, int H, int W, int ca, int ldin,
                        int cb, int ldout, int relu_in, int act_out, float* ws, int64_t ws_bytes) {
  GI_REQUIRE(ctx && in && w_phase && out, "convT_s2_forward: null pointer");
  IgemmArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in; a.w = w_phase; a.out = out; a.ws = ws; a.ws_bytes = ws_bytes;
  a.tickets = ws ? ctx_tickets(ctx) : nullptr;
  a.n = n; a.Hs = H; a.Ws = W;
  a.cin = ca; a.ldin = ldin; a.cout = cb; a.ldout = ldout;
  a.relu_in = relu_in; a.act_out = act_out;
  return op_igemm(ctx->stream, dtype, 1, a);
}

// the same two layers with the optional fused epilogues of IgemmArgs (what the networks pass; include/ganinpaint.h: gi_igemm_ex)
static int apply_ex(IgemmArgs& a, const gi_igemm_ex* ex) {
  if (!ex) return GI_OK;
  // relu_cend is a HINT (IgemmArgs::relu_cend): kernels apply the input ReLU per 32- or 64-channel chunk up to it, the generic
  // kernel to every channel - all agree only when the channels from relu_cend on are non-negative already and it is chunk-aligned
  GI_REQUIRE(ex->relu_cend >= 0 && ex->relu_cend % 64 == 0 && ex->relu_cend <= a.cin, "igemm_ex: relu_cend=%d must be a multiple of 64 within the %d input channels",
             ex->relu_cend, a.cin);
  a.relu_cend = ex->relu_cend;
  a.mask = ex->mask; a.ldmask = ex->ldmask; a.mask_slope = ex->mask_slope;
  a.add = ex->add; a.ldadd = ex->ldadd;
  a.mask_bits = ex->mask ? ex->mask_bits : nullptr;
  a.stat_acc = ex->stat_acc; a.stat_reps = ex->stat_reps; a.stat_pg = ex->stat_pg;
  a.partials = ex->partials;
  a.bwd_x = ex->bwd_x; a.bwd_ldx = ex->bwd_ldx; a.bwd_scale = ex->bwd_scale; a.bwd_shift = ex->bwd_shift; a.bwd_mean = ex->bwd_mean;
  a.bwd_inv = ex->bwd_inv; a.bwd_stride = ex->bwd_stride; a.bwd_slope = ex->bwd_slope; a.bwd_acc = ex->bwd_acc; a.bwd_reps = ex->bwd_reps;
  a.bwd_pg = ex->bwd_pg;
  a.bwd_c0 = ex->bwd_c0; a.bwd_c = ex->bwd_c;
  return GI_OK;
}
static void return_ex(const IgemmArgs& a, gi_igemm_ex* ex) {
  if (!ex) return;
  ex->mask_applied = a.mask_applied; ex->bwd_applied = a.bwd_applied; ex->stat_used = a.stat_used; ex->ntiles_out = a.ntiles_out;
}

int gi_conv_s2_forward_ex(gi_ctx* ctx, int dtype, const void* in, const void* w_packed, void* out, int n, int H, int W, int cb, int ldin,
                          int ca, int ldout, int relu_in, int act_out, float* ws, int64_t ws_bytes, gi_igemm_ex* ex) {
  GI_REQUIRE(ctx && in && w_packed && out, "conv_s2_forward_ex: null pointer");
  GI_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv_s2_forward_ex: H=%d W=%d must be even", H, W);
  IgemmArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in; a.w = w_packed; a.out = out; a.ws = ws; a.ws_bytes = ws_bytes;
  a.tickets = ws ? ctx_tickets(ctx) : nullptr;
  a.n = n; a.Hs = H / 2; a.Ws = W / 2;
  a.cin = cb; a.ldin = ldin; a.cout = ca; a.ldout = ldout;
  a.relu_in = relu_in; a.act_out = act_out;
  GI_TRY(apply_ex(a, ex));
  const int rc = op_igemm(ctx->stream, dtype, 0, a);
  return_ex(a, ex);
  return rc;
}

int gi_convT_s2_forward_ex(gi_ctx* ctx, int dtype, const void* in, const void* w_phase, void* out, int n, int H, int W, int ca, int ldin,
                           int cb, int ldout, int relu_in, int act_out, float* ws, int64_t ws_bytes, gi_igemm_ex* ex) {
  GI_REQUIRE(ctx && in && w_phase && out, "convT_s2_forward_ex: null pointer");
  IgemmArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in; a.w = w_phase; a.out = out; a.ws = ws; a.ws_bytes = ws_bytes;
  a.tickets = ws ? ctx_tickets(ctx) : nullptr;
  a.n = n; a.Hs = H; a.Ws = W;
  a.cin = ca; a.ldin = ldin; a.cout = cb; a.ldout = ldout;
  a.relu_in = relu_in; a.act_out = act_out;
  GI_TRY(apply_ex(a, ex));
  const int rc = op_igemm(ctx->stream, dtype, 1, a);
  return_ex(a, ex);
  return rc;
}

int64_t gi_stat_acc_words(int c) { return gi_stat_block_words(c, GI_STAT_MAXREP); }

namespace {
__global__ void __launch_bounds__(256) stat_acc_read_kernel(const unsigned long long* acc, int c, int reps, int group, double* out) {
  for (int ch = blockIdx.x * 256 + threadIdx.x; ch < c; ch += gridDim.x * 256) {
    out[ch] = gi_stat_read(acc, c, reps, group, 0, ch);
    out[c + ch] = gi_stat_read(acc, c, reps, group, 1, ch);
  }
}
}  // namespace

int gi_stat_acc_read(gi_ctx* ctx, const unsigned long long* acc, int c, int reps, int group, double* out_dev) {
  GI_REQUIRE(ctx && acc && out_dev && c > 0 && reps >= 1 && reps <= GI_STAT_MAXREP && (group == 0 || group == 1), "stat_acc_read: bad argument");
  hipLaunchKernelGGL(stat_acc_read_kernel, dim3((c + 255) / 256), dim3(256), 0, ctx->stream, acc, c, reps, group, out_dev);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

int gi_wgrad_s2(gi_ctx* ctx, int dtype, const void* S, const void* L, float* dW, int n, int Hs, int Ws, int ca, int ldS, int cb, int ldL,
                int relu_S, float scale) {
  GI_REQUIRE(ctx && S && L && dW, "wgrad_s2: null pointer");
  WgradArgs a;
  a.S = S; a.L = L; a.dW = dW; a.n = n; a.Hs = Hs; a.Ws = Ws;
  a.ca = ca; a.ldS = ldS; a.coffS = 0; a.cb = cb; a.ldL = ldL; a.coffL = 0;
  a.relu_S = relu_S; a.scale = scale;
  a.scratch = nullptr; a.scratch_bytes = 0;      // atomics across the pixel-range splits
  return op_wgrad(ctx->stream, dtype, a);
}

int64_t gi_wgrad_s2_scratch_bytes(int dtype, int n, int Hs, int Ws, int ca, int cb) {
  return op_wgrad_scratch_bytes(dtype, n, Hs, Ws, ca, cb);
}

int gi_wgrad_s2_ws(gi_ctx* ctx, int dtype, const void* S, const void* L, float* dW, int n, int Hs, int Ws, int ca, int ldS, int cb, int ldL,
                   int relu_S, float scale, float* scratch, int64_t scratch_bytes) {
  GI_REQUIRE(ctx && S && L && dW, "wgrad_s2_ws: null pointer");
  WgradArgs a;
  a.S = S; a.L = L; a.dW = dW; a.n = n; a.Hs = Hs; a.Ws = Ws;
  a.ca = ca; a.ldS = ldS; a.coffS = 0; a.cb = cb; a.ldL = ldL; a.coffL = 0;
  a.relu_S = relu_S; a.scale = scale;
  a.scratch = scratch; a.scratch_bytes = scratch_bytes;
  return op_wgrad(ctx->stream, dtype, a);
}


int gi_pack_weights(gi_ctx* ctx, int dtype, const float* w, int ca, int cb, void* w_packed, void* w_phase) {
  GI_REQUIRE(ctx && w, "pack_weights: null pointer");
  return op_pack_weights(ctx->stream, dtype, w, ca, cb, w_packed, w_phase);
}
int gi_convert(gi_ctx* ctx, int dtype, const float* src, void* dst, int64_t count) {
  GI_REQUIRE(ctx && src && dst, "convert: null pointer");
  return op_convert(ctx->stream, dtype, src, dst, count);
}
int gi_convert_back(gi_ctx* ctx, int dtype, const void* src, float* dst, int64_t count) {
  GI_REQUIRE(ctx && src && dst, "convert_back: null pointer");
  return op_convert_back(ctx->stream, dtype, src, dst, count);
}

int gi_time_convT_s2(gi_ctx* ctx, int dtype, const void* in, const void* w_phase, void* out, int n, int H, int W, int ca, int ldin, int cb,
                     int ldout, int iters, float* ms_out_host) {
  GI_REQUIRE(ctx && ms_out_host && iters > 0, "time_convT_s2: bad argument");
  hipEvent_t e0, e1;
  GI_HIP(hipEventCreate(&e0));
  GI_HIP(hipEventCreate(&e1));
  const int relu = gi_tune("GI_TIME_RELU", 1);   // (ablation build: 0 times the variant without the fused input ReLU)
  // warm-up
  GI_TRY(gi_convT_s2_forward(ctx, dtype, in, w_phase, out, n, H, W, ca, ldin, cb, ldout, relu, GI_ACT_NONE, nullptr, 0));
  GI_HIP(hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < iters; ++i)
    GI_TRY(gi_convT_s2_forward(ctx, dtype, in, w_phase, out, n, H, W, ca, ldin, cb, ldout, relu, GI_ACT_NONE, nullptr, 0));
  GI_HIP(hipEventRecord(e1, ctx->stream));
  GI_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  GI_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out_host = ms / iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return GI_OK;
}

/* the same for the strided convolution (gi_conv_s2_forward): the critic's conv2 / the generator's d2 family */
int gi_time_conv_s2(gi_ctx* ctx, int dtype, const void* in, const void* w_packed, void* out, int n, int H, int W, int cb, int ldin, int ca,
                    int ldout, int iters, float* ms_out_host) {
  GI_REQUIRE(ctx && ms_out_host && iters > 0, "time_conv_s2: bad argument");
  hipEvent_t e0, e1;
  GI_HIP(hipEventCreate(&e0));
  GI_HIP(hipEventCreate(&e1));
  GI_TRY(gi_conv_s2_forward(ctx, dtype, in, w_packed, out, n, H, W, cb, ldin, ca, ldout, 0, GI_ACT_NONE, nullptr, 0));
  GI_HIP(hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < iters; ++i)
    GI_TRY(gi_conv_s2_forward(ctx, dtype, in, w_packed, out, n, H, W, cb, ldin, ca, ldout, 0, GI_ACT_NONE, nullptr, 0));
  GI_HIP(hipEventRecord(e1, ctx->stream));
  GI_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  GI_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out_host = ms / iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return GI_OK;
}

}  // extern "C"
