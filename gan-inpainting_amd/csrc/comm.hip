// Data-parallel gradient exchange behind the C-ABI: SUM all-reduce of a network's flat fp32 gradient
// buffer over RCCL (xGMI), asynchronous on a caller-given HIP stream. The reference is single-device (train.py:44-46):
// there is no collective to mirror; this is the exchange step BASELINE.json's north_star names.
// librccl is loaded on first use (dlopen), so that a host without it can still load libganinpaint.so.
#include <dlfcn.h>
#include <string.h>

#include <rccl/rccl.h>

#include "common.h"

struct gi_net;
float* gi_net_grads_ptr(gi_net* net, int64_t* floats);   // net.hip

struct gi_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipEvent_t done = nullptr;
};

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) {
      r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
      r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
      r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
      r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
      if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce) r.handle = h;
    }
  }
  return r.handle ? &r : nullptr;
}

#define GI_NCCL(expr)                                                                                   \
  do {                                                                                                  \
    ncclResult_t _r = (expr);                                                                           \
    if (_r != ncclSuccess) {                                                                            \
      gi_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, R->GetErrorString ? R->GetErrorString(_r) : "rccl error"); \
      return GI_ERR_HIP;                                                                                \
    }                                                                                                   \
  } while (0)

}  // namespace

extern "C" {

int gi_comm_unique_id(char* id128_host) {
  GI_REQUIRE(id128_host, "comm_unique_id: null");
  Rccl* R = rccl();
  if (!R) { gi_set_error("comm: librccl.so.1 could not be loaded (%s)", dlerror()); return GI_ERR_UNSUPPORTED; }
  ncclUniqueId id;
  GI_NCCL(R->GetUniqueId(&id));
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id128_host, id.internal, 128);
  return GI_OK;
}

int gi_comm_create(const char* id128_host, int rank, int world, int device_id, gi_comm** out) {
  GI_REQUIRE(id128_host && out && world >= 1 && rank >= 0 && rank < world, "comm_create: rank=%d world=%d", rank, world);
  Rccl* R = rccl();
  if (!R) { gi_set_error("comm: librccl.so.1 could not be loaded (%s)", dlerror()); return GI_ERR_UNSUPPORTED; }
  GI_HIP(hipSetDevice(device_id));
  gi_comm* c = new gi_comm();
  c->rank = rank; c->world = world; c->device = device_id;
  ncclUniqueId id;
  memcpy(id.internal, id128_host, 128);
  ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    gi_set_error("comm_create: ncclCommInitRank(rank %d of %d) -> %s", rank, world, R->GetErrorString ? R->GetErrorString(r) : "rccl error");
    delete c;
    return GI_ERR_HIP;
  }
  if (hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
    R->CommDestroy(c->comm);
    delete c;
    gi_set_error("comm_create: hipEventCreate failed");
    return GI_ERR_HIP;
  }
  *out = c;
  return GI_OK;
}

int gi_comm_destroy(gi_comm* comm) {
  if (!comm) return GI_OK;
  Rccl* R = rccl();
  if (comm->done) (void)hipEventDestroy(comm->done);
  if (R && comm->comm) R->CommDestroy(comm->comm);
  delete comm;
  return GI_OK;
}

int gi_allreduce_sum_f32(gi_comm* comm, float* buf, int64_t count, void* hip_stream) {
  GI_REQUIRE(comm && buf && count > 0, "allreduce: bad argument");
  Rccl* R = rccl();
  GI_REQUIRE(R != nullptr, "allreduce: RCCL not loaded");
  GI_NCCL(R->AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, comm->comm, (hipStream_t)hip_stream));
  return GI_OK;
}

int gi_net_allreduce_grads_async(gi_net* net, gi_comm* comm, int64_t begin, int64_t end, void* comm_stream) {
  int64_t n = 0;
  float* g = gi_net_grads_ptr(net, &n);
  GI_REQUIRE(g != nullptr, "allreduce_grads: net not bound");
  if (end < 0) end = n;
  GI_REQUIRE(begin >= 0 && begin < end && end <= n, "allreduce_grads: range [%lld, %lld) of %lld floats", (long long)begin, (long long)end, (long long)n);
  return gi_allreduce_sum_f32(comm, g + begin, end - begin, comm_stream);
}

int gi_allreduce_wait(gi_comm* comm, void* comm_stream, void* compute_stream) {
  GI_REQUIRE(comm, "allreduce_wait: null");
  GI_HIP(hipEventRecord(comm->done, (hipStream_t)comm_stream));
  GI_HIP(hipStreamWaitEvent((hipStream_t)compute_stream, comm->done, 0));
  return GI_OK;
}

}  // extern "C"
