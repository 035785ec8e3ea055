// Data-parallel gradient exchange with the communicator owned by the library (sdm_dp_init /
// sdm_dp_allreduce_bucket / sdm_dp_finalize).  The reference trains on one device (train_diffusion.py:49); the exchange is
// north_star's requirement: plain data parallelism, one process per GPU, RCCL over xGMI, a bucket's all-reduce enqueued
// the moment the backward has finished its blocks so that it overlaps the rest of the backward and the optimizer.
//
// RCCL is bound at run time (dlopen of librccl.so.1: inside a PyTorch process that is the copy torch has already mapped,
// so the process holds ONE RCCL), never at link time: libsdm_hip.so loads on a machine without RCCL and every sdm_dp_*
// call then fails with SDM_ERR_COMM and a text.  RCCL results are returned as SDM_ERR_COMM + sdm_last_error() carrying
// ncclGetErrorString / ncclGetLastError; asynchronous failures of the network / a lost peer through sdm_dp_async_error.
//
// Two exchange forms per bucket (fp32 gradient slice `buf`, in place):
//   wire f32 : one ncclAllReduce(sum) on the caller's stream -- a ring / tree inside RCCL.
//   wire bf16: the "mesh" exchange for the fully connected xGMI topology (7 point-to-point links per GPU, no switch):
//              1. pack: the slice is cut into `world` shards; every shard is rounded to bf16 into the send buffer
//              2. all-to-all of the shards (grouped ncclSend / ncclRecv: W - 1 concurrent point-to-point transfers, one
//                 per link, each (n / W) * 2 bytes)
//              3. reduce: rank r sums the W received copies of shard r in FP32 (its own contribution taken from the fp32
//                 slice, unrounded), rounds the sum to bf16 once
//              4. all-gather of the reduced shards (bf16), unpack to fp32 in place
//              Bytes on each link: 2 * (n / W) * 2 against a ring's 2 * (W - 1) / W * n * 4 through ONE link pair per hop;
//              every element is rounded twice (once before, once after the sum) whatever W is, the sum itself is fp32.
//              Deterministic: the sum runs over ranks in rank order on every rank's shard.
//   `average` divides by the world size inside the unpack (bf16) or with ncclAvg (f32).
// Everything is enqueued on the caller's stream; nothing synchronises.  Not capturable into a hipGraph (RCCL's own rule
// for grouped point-to-point on this version is not relied upon): the trainer calls it eagerly between graph segments.
#include <dlfcn.h>

#include "sdm_common.h"

#define SDM_DP_ID_BYTES 128

namespace {

typedef int ncclResult_t;                  // ncclSuccess == 0, ncclInProgress == 7
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[SDM_DP_ID_BYTES]; };
enum { kNcclSum = 0, kNcclAvg = 4, kNcclFloat32 = 7, kNcclBfloat16 = 9, kNcclInProgress = 7 };

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  const char* (*GetLastError)(ncclComm_t) = nullptr;
  int (*GetVersion)(int*) = nullptr;
};
Rccl g_rccl;
char g_rccl_why[256] = "";

template <typename F>
bool bind(F& f, const char* name) {
  f = (F)dlsym(g_rccl.h, name);
  if (!f) snprintf(g_rccl_why, sizeof g_rccl_why, "librccl: symbol %s missing", name);
  return f != nullptr;
}

bool rccl_load() {
  if (g_rccl.h) return true;
  if (g_rccl_why[0]) return false;      // tried before
  const char* over = getenv("SDM_RCCL_LIB");
  const char* names[] = {over, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    if (!n || !n[0]) continue;
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    snprintf(g_rccl_why, sizeof g_rccl_why, "librccl.so.1 not loadable (%s)", dlerror());
    return false;
  }
  g_rccl.h = h;
  const bool ok = bind(g_rccl.GetUniqueId, "ncclGetUniqueId") && bind(g_rccl.CommInitRank, "ncclCommInitRank") &&
                  bind(g_rccl.CommDestroy, "ncclCommDestroy") && bind(g_rccl.CommAbort, "ncclCommAbort") &&
                  bind(g_rccl.CommGetAsyncError, "ncclCommGetAsyncError") && bind(g_rccl.AllReduce, "ncclAllReduce") &&
                  bind(g_rccl.AllGather, "ncclAllGather") && bind(g_rccl.Send, "ncclSend") && bind(g_rccl.Recv, "ncclRecv") &&
                  bind(g_rccl.GroupStart, "ncclGroupStart") && bind(g_rccl.GroupEnd, "ncclGroupEnd") &&
                  bind(g_rccl.GetErrorString, "ncclGetErrorString") && bind(g_rccl.GetLastError, "ncclGetLastError") &&
                  bind(g_rccl.GetVersion, "ncclGetVersion");
  if (!ok) {
    g_rccl.h = nullptr;
    return false;
  }
  return true;
}

}  // namespace

struct sdm_dp_comm {
  unsigned magic;
  ncclComm_t comm;
  int rank, world, device;
  // bf16-wire staging (device): send = `world` packed shards, recv = `world` received copies; grown on demand OUTSIDE the
  // exchange (sdm_dp_reserve) or lazily at the first bucket of a size (a hipMalloc: synchronises once, never again)
  bf16_t* send;
  bf16_t* recv;
  size_t cap_elems;     // shard capacity * world
};
#define SDM_DP_MAGIC 0x5d3dc0deu

static int comm_error(sdm_dp_comm* c, const char* what, ncclResult_t r) {
  const char* last = (c && c->comm) ? g_rccl.GetLastError(c->comm) : nullptr;
  return sdm_set_error(SDM_ERR_COMM, "%s: %s%s%s", what, g_rccl.GetErrorString(r), (last && last[0]) ? " | " : "",
                       (last && last[0]) ? last : "");
}
#define DP_NCCL(c, call, what)                        \
  do {                                                \
    ncclResult_t r_ = (call);                         \
    if (r_ != 0) return comm_error((c), (what), r_);  \
  } while (0)
#define DP_COMM(c)                                                                                     \
  do {                                                                                                 \
    if (!(c) || (c)->magic != SDM_DP_MAGIC) return sdm_set_error(SDM_ERR_ARG, "sdm_dp: not a live communicator"); \
  } while (0)

// ---- pack / reduce / unpack kernels of the bf16-wire exchange.  HBM-bound streaming passes: 16 bytes per lane and access,
// grid-stride, bounds from the caller.  n = elements of the bucket, shard = ceil(n / world) rounded up to 8.
__global__ __launch_bounds__(256) void dp_pack_kernel(const float* __restrict__ g, bf16_t* __restrict__ send, long long n,
                                                      long long shard, int world, int vec) {
  // send[w][i] = bf16(g[w * shard + i]) (zero beyond n): 8 elements per thread
  const long long total8 = shard * world / 8;
  for (long long q = blockIdx.x * 256ll + threadIdx.x; q < total8; q += (long long)gridDim.x * 256) {
    const long long e = q * 8;
    u16x8 o;
    if (vec && e + 8 <= n) {      // vec: the bucket starts on a 16-byte boundary
      const f32x4 a = *(const f32x4*)(g + e), b = *(const f32x4*)(g + e + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { o[i] = f2bf(a[i]); o[4 + i] = f2bf(b[i]); }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = e + i < n ? f2bf(g[e + i]) : (bf16_t)0;
    }
    *(u16x8*)(send + e) = o;
  }
}

__global__ __launch_bounds__(256) void dp_reduce_kernel(const float* __restrict__ g, const bf16_t* __restrict__ recv,
                                                        bf16_t* __restrict__ out, long long n, long long shard, int world,
                                                        int rank) {
  // out[i] = bf16( sum over ranks w (rank order) of contribution w to element rank * shard + i ), own contribution in fp32
  const long long base = (long long)rank * shard;
  for (long long q = blockIdx.x * 256ll + threadIdx.x; q < shard / 8; q += (long long)gridDim.x * 256) {
    const long long i = q * 8;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    for (int w = 0; w < world; ++w) {
      if (w == rank) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += base + i + k < n ? g[base + i + k] : 0.f;
      } else {
        const u16x8 v = *(const u16x8*)(recv + (long long)w * shard + i);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += bf2f(v[k]);
      }
    }
    u16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = f2bf(s[k]);
    *(u16x8*)(out + i) = o;
  }
}

__global__ __launch_bounds__(256) void dp_unpack_kernel(const bf16_t* __restrict__ gathered, float* __restrict__ g, long long n,
                                                        float mul, int vec) {
  for (long long q = blockIdx.x * 256ll + threadIdx.x; q < (n + 7) / 8; q += (long long)gridDim.x * 256) {
    const long long e = q * 8;
    const u16x8 v = *(const u16x8*)(gathered + e);
    if (vec && e + 8 <= n) {
      f32x4 a, b;
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = bf2f(v[i]) * mul; b[i] = bf2f(v[4 + i]) * mul; }
      *(f32x4*)(g + e) = a;
      *(f32x4*)(g + e + 4) = b;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (e + i < n) g[e + i] = bf2f(v[i]) * mul;
    }
  }
}

static long long dp_shard(long long n, int world) {
  const long long s = (n + world - 1) / world;
  return (s + 7) / 8 * 8;
}

static int dp_grow(sdm_dp_comm* c, long long n) {
  const size_t need = (size_t)dp_shard(n, c->world) * c->world;
  if (need <= c->cap_elems) return SDM_OK;
  if (c->send) (void)hipFree(c->send);
  if (c->recv) (void)hipFree(c->recv);
  c->send = c->recv = nullptr;
  c->cap_elems = 0;
  if (hipMalloc((void**)&c->send, need * sizeof(bf16_t)) != hipSuccess || hipMalloc((void**)&c->recv, need * sizeof(bf16_t)) != hipSuccess) {
    (void)hipGetLastError();
    return sdm_set_error(SDM_ERR_LAUNCH, "sdm_dp: cannot allocate 2 x %zu bytes of exchange staging", need * sizeof(bf16_t));
  }
  c->cap_elems = need;
  return SDM_OK;
}

extern "C" {

int sdm_dp_available(void) {
  if (!rccl_load()) return sdm_set_error(SDM_ERR_COMM, "%s", g_rccl_why);
  int v = 0;
  if (g_rccl.GetVersion(&v) != 0) return sdm_set_error(SDM_ERR_COMM, "ncclGetVersion failed");
  return v;
}

int sdm_dp_unique_id(void* out, size_t bytes) {
  SDM_CHECK_ARG(out && bytes == SDM_DP_ID_BYTES, "sdm_dp_unique_id: the id buffer is %d bytes", SDM_DP_ID_BYTES);
  if (!rccl_load()) return sdm_set_error(SDM_ERR_COMM, "%s", g_rccl_why);
  ncclUniqueId id;
  DP_NCCL(nullptr, g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(out, id.internal, SDM_DP_ID_BYTES);
  return SDM_OK;
}

int sdm_dp_init(const void* id, size_t bytes, int rank, int world, sdm_dp_comm_t* out) {
  SDM_CHECK_ARG(out, "sdm_dp_init: null result pointer");
  *out = nullptr;
  SDM_CHECK_ARG(id && bytes == SDM_DP_ID_BYTES, "sdm_dp_init: the id buffer is %d bytes", SDM_DP_ID_BYTES);
  SDM_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "sdm_dp_init: rank %d of %d", rank, world);
  if (!rccl_load()) return sdm_set_error(SDM_ERR_COMM, "%s", g_rccl_why);
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return sdm_set_error(SDM_ERR_LAUNCH, "sdm_dp_init: no current HIP device");
  ncclUniqueId uid;
  memcpy(uid.internal, id, SDM_DP_ID_BYTES);
  sdm_dp_comm* c = (sdm_dp_comm*)calloc(1, sizeof(sdm_dp_comm));
  if (!c) return sdm_set_error(SDM_ERR_LAUNCH, "sdm_dp_init: out of host memory");
  c->rank = rank; c->world = world; c->device = dev;
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
  if (r != 0) {
    const int rc = comm_error(nullptr, "ncclCommInitRank", r);
    free(c);
    return rc;
  }
  c->magic = SDM_DP_MAGIC;
  *out = c;
  return SDM_OK;
}

int sdm_dp_rank(sdm_dp_comm_t c) { DP_COMM(c); return c->rank; }
int sdm_dp_world(sdm_dp_comm_t c) { DP_COMM(c); return c->world; }

int sdm_dp_reserve(sdm_dp_comm_t c, int64_t max_count) {
  DP_COMM(c);
  SDM_CHECK_ARG(max_count >= 0, "sdm_dp_reserve: negative count");
  return dp_grow(c, max_count);
}

int sdm_dp_allreduce_bucket(sdm_dp_comm_t c, float* buf, int64_t count, int wire_bf16, int average, sdm_stream_t stream) {
  DP_COMM(c);
  SDM_CHECK_ARG(buf && count >= 0, "sdm_dp_allreduce_bucket: null buffer / negative count");
  SDM_CHECK_ARG((((uintptr_t)buf) & 3) == 0, "sdm_dp_allreduce_bucket: misaligned fp32 buffer");
  const int vec = aligned16(buf);      // bucket boundaries fall on arbitrary elements of the flat gradient buffer
  if (count == 0) return SDM_OK;
  hipStream_t st = (hipStream_t)stream;
  sdm_plan_taint("sdm_dp_allreduce_bucket");
  if (!wire_bf16) {
    DP_NCCL(c, g_rccl.AllReduce(buf, buf, (size_t)count, kNcclFloat32, average ? kNcclAvg : kNcclSum, c->comm, st), "ncclAllReduce");
    return SDM_OK;
  }
  const int W = c->world;
  const long long shard = dp_shard(count, W);
  if (int rc = dp_grow(c, count)) return rc;
  const int blocks = (int)((shard * W / 8 + 255) / 256 < 2048 ? (shard * W / 8 + 255) / 256 : 2048);
  SDM_LAUNCH(dp_pack_kernel, dim3(blocks), dim3(256), 0, st, (const float*)buf, c->send, (long long)count, shard, W, vec);
  SDM_CHECK_LAUNCH("dp_pack");
  if (W > 1) {
    DP_NCCL(c, g_rccl.GroupStart(), "ncclGroupStart");
    for (int k = 1; k < W; ++k) {
      // peer order staggered by rank: at any position of the list every rank talks to a different peer
      const int to = (c->rank + k) % W, from = (c->rank - k + W) % W;
      ncclResult_t r1 = g_rccl.Send(c->send + (long long)to * shard, (size_t)shard, kNcclBfloat16, to, c->comm, st);
      ncclResult_t r2 = r1 == 0 ? g_rccl.Recv(c->recv + (long long)from * shard, (size_t)shard, kNcclBfloat16, from, c->comm, st) : r1;
      if (r2 != 0) {
        (void)g_rccl.GroupEnd();
        return comm_error(c, "ncclSend/ncclRecv", r2);
      }
    }
    DP_NCCL(c, g_rccl.GroupEnd(), "ncclGroupEnd");
  }
  const int rblocks = (int)((shard / 8 + 255) / 256 < 2048 ? (shard / 8 + 255) / 256 : 2048);
  // the reduced shard goes to send[rank] (its packed copy is dead: a rank never sends its own shard)
  SDM_LAUNCH(dp_reduce_kernel, dim3(rblocks), dim3(256), 0, st, (const float*)buf, (const bf16_t*)c->recv,
             c->send + (long long)c->rank * shard, (long long)count, shard, W, c->rank);
  SDM_CHECK_LAUNCH("dp_reduce");
  const bf16_t* gathered = c->send;
  if (W > 1) {
    DP_NCCL(c, g_rccl.AllGather(c->send + (long long)c->rank * shard, c->recv, (size_t)shard, kNcclBfloat16, c->comm, st), "ncclAllGather");
    gathered = c->recv;
  }
  SDM_LAUNCH(dp_unpack_kernel, dim3(blocks), dim3(256), 0, st, gathered, buf, (long long)count, average ? 1.0f / (float)W : 1.0f, vec);
  SDM_CHECK_LAUNCH("dp_unpack");
  return SDM_OK;
}

int sdm_dp_async_error(sdm_dp_comm_t c) {
  DP_COMM(c);
  ncclResult_t async = 0;
  DP_NCCL(c, g_rccl.CommGetAsyncError(c->comm, &async), "ncclCommGetAsyncError");
  if (async != 0 && async != kNcclInProgress) return comm_error(c, "asynchronous RCCL error", async);
  return SDM_OK;
}

int sdm_dp_finalize(sdm_dp_comm_t c) {
  DP_COMM(c);
  c->magic = 0;
  if (c->send) (void)hipFree(c->send);
  if (c->recv) (void)hipFree(c->recv);
  ncclResult_t r = g_rccl.CommDestroy(c->comm);
  const int rc = r == 0 ? SDM_OK : comm_error(nullptr, "ncclCommDestroy", r);
  free(c);
  return rc;
}

}  // extern "C"
