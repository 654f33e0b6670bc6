// C-ABI glue: error reporting, context, single-layer entry points (see include/ganinpaint.h).
#include <string.h>

#include <stdlib.h>

#include "common.h"

static thread_local char g_err[1024] = "";

void gi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

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
  const char* er = getenv("GI_TIME_RELU");     // tools: 0 times the variant without the fused input ReLU (input-gradient use)
  const int relu = er ? atoi(er) : 1;
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

}  // extern "C"
