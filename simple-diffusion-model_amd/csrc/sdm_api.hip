// Library-level entry points: version, error text, device probe.
#include "sdm_common.h"

static thread_local char g_err[512] = "";

int sdm_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

unsigned long long g_sdm_launches = 0;

extern "C" int sdm_version(void) { return 400; }

// ---- device-side failure word -------------------------------------------------------------------------------------
volatile unsigned* g_sdm_errw_host = nullptr;
unsigned* g_sdm_errw_dev = nullptr;

static const char* deverr_text(unsigned code) {
  switch (code) {
    case SDM_DEVERR_GN_COOP_TIMEOUT: return "sdm_gn_apply_coop: a slab gave up waiting for the other slabs of its image (outputs are NaN)";
    case SDM_DEVERR_GN_BWD_COOP_TIMEOUT: return "sdm_gn_bwd_coop: a slab gave up waiting for the other slabs of its image (outputs are NaN)";
    case SDM_DEVERR_DP_TIMEOUT: return "sdm_dp: a device-side wait gave up";
    default: return "unknown device-side failure";
  }
}
int sdm_report_device_error(const char* where) {
  const unsigned code = *g_sdm_errw_host;
  *g_sdm_errw_host = 0;                      // reported once
  return sdm_set_error(SDM_ERR_DEVICE, "device-side failure %u seen at %s: %s", code, where, deverr_text(code));
}
// Allocates the word (idempotent).  Must run outside stream capture (hipHostMalloc); the host language calls it once per
// process before its first launch.
extern "C" int sdm_device_error_init(void) {
  if (g_sdm_errw_host) return SDM_OK;
  void* h = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) return sdm_set_error(SDM_ERR_LAUNCH, "device_error_init: hipHostMalloc failed");
  memset(h, 0, 64);
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return sdm_set_error(SDM_ERR_LAUNCH, "device_error_init: no device pointer"); }
  g_sdm_errw_dev = (unsigned*)d;
  g_sdm_errw_host = (volatile unsigned*)h;
  return SDM_OK;
}
// 0, or SDM_ERR_DEVICE (sdm_last_error names the kernel) when a kernel has reported a failure since the last report.
// Does not synchronise: call it after a synchronisation point to be sure that everything enqueued has been looked at.
extern "C" int sdm_device_error_poll(void) {
  if (g_sdm_errw_host && *g_sdm_errw_host) return sdm_report_device_error("sdm_device_error_poll");
  return SDM_OK;
}

extern "C" unsigned long long sdm_launch_count(void) { return g_sdm_launches; }

extern "C" const char* sdm_last_error(void) { return g_err; }

extern "C" int sdm_device_info(int* out4) {
  SDM_CHECK_ARG(out4 != nullptr, "device_info: null");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return sdm_set_error(SDM_ERR_LAUNCH, "device_info: no HIP device");
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return sdm_set_error(SDM_ERR_LAUNCH, "device_info: props failed");
  out4[0] = p.multiProcessorCount;
  out4[1] = (int)p.maxSharedMemoryPerMultiProcessor;
  out4[2] = p.warpSize;
  out4[3] = strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
  return SDM_OK;
}

// ---- launch plans -------------------------------------------------------------------------------------------------
#include <algorithm>
#include <vector>
namespace {
struct PlanRec { const void* fn; dim3 grid, block; unsigned lds; int nargs; size_t blob_off; unsigned arg_off[32]; };
struct Plan { std::vector<PlanRec> recs; std::vector<unsigned char> blob; bool tainted = false; char why[64] = ""; bool live = false; };
std::vector<Plan*> g_plans;
Plan* g_rec = nullptr;
}  // namespace

bool sdm_plan_recording() { return g_rec != nullptr; }
void sdm_plan_taint(const char* why) {
  if (g_rec && !g_rec->tainted) {
    g_rec->tainted = true;
    snprintf(g_rec->why, sizeof(g_rec->why), "%s", why);
  }
}
void sdm_plan_push(const void* fn, dim3 grid, dim3 block, size_t lds, void** args, const size_t* sizes, int nargs) {
  if (!g_rec) return;
  if (nargs > 32) { sdm_plan_taint("more than 32 kernel arguments"); return; }
  PlanRec r;
  r.fn = fn; r.grid = grid; r.block = block; r.lds = (unsigned)lds; r.nargs = nargs;
  size_t off = (g_rec->blob.size() + 15) & ~(size_t)15;
  r.blob_off = off;
  size_t need = off;
  for (int i = 0; i < nargs; ++i) {
    need = (need + 15) & ~(size_t)15;             // every argument 16-byte aligned inside the blob
    r.arg_off[i] = (unsigned)(need - off);
    need += sizes[i];
  }
  g_rec->blob.resize(need);
  for (int i = 0; i < nargs; ++i) memcpy(g_rec->blob.data() + off + r.arg_off[i], args[i], sizes[i]);
  g_rec->recs.push_back(r);
}

extern "C" int sdm_plan_begin(void) {
  SDM_CHECK_ARG(g_rec == nullptr, "plan_begin: a plan is already being recorded");
  g_rec = new Plan();
  return SDM_OK;
}
// Ends the recording.  Returns the plan's handle (>= 1), or 0 when something ran that a plan cannot replay (the
// recording is dropped; sdm_last_error() names it) -- the caller then keeps replaying its hipGraph.
extern "C" int sdm_plan_end(void) {
  SDM_CHECK_ARG(g_rec != nullptr, "plan_end: no plan is being recorded");
  Plan* p = g_rec;
  g_rec = nullptr;
  if (p->tainted || p->recs.empty()) {
    (void)sdm_set_error(0, "plan dropped: %s", p->tainted ? p->why : "no launches");
    delete p;
    return 0;
  }
  p->live = true;
  for (size_t i = 0; i < g_plans.size(); ++i)
    if (g_plans[i] == nullptr) { g_plans[i] = p; return (int)i + 1; }
  g_plans.push_back(p);
  return (int)g_plans.size();
}
extern "C" int sdm_plan_launches(int handle) {
  SDM_CHECK_ARG(handle >= 1 && handle <= (int)g_plans.size() && g_plans[handle - 1], "plan_launches: bad handle");
  return (int)g_plans[handle - 1]->recs.size();
}
extern "C" int sdm_plan_run(int handle, int repeat, sdm_stream_t stream) {
  SDM_CHECK_ARG(handle >= 1 && handle <= (int)g_plans.size() && g_plans[handle - 1] && repeat >= 1, "plan_run: bad handle");
  Plan* p = g_plans[handle - 1];
  void* ptrs[32];
  for (int rep = 0; rep < repeat; ++rep)
    for (const PlanRec& r : p->recs) {
      unsigned char* base = p->blob.data() + r.blob_off;
      for (int i = 0; i < r.nargs; ++i) ptrs[i] = base + r.arg_off[i];
      if (hipLaunchKernel(r.fn, r.grid, r.block, ptrs, r.lds, (hipStream_t)stream) != hipSuccess)
        return sdm_set_error(SDM_ERR_LAUNCH, "plan_run: %s", hipGetErrorString(hipGetLastError()));
    }
  g_sdm_launches += (unsigned long long)repeat * p->recs.size();
  return SDM_OK;
}
extern "C" int sdm_plan_free(int handle) {
  SDM_CHECK_ARG(handle >= 1 && handle <= (int)g_plans.size() && g_plans[handle - 1], "plan_free: bad handle");
  delete g_plans[handle - 1];
  g_plans[handle - 1] = nullptr;
  return SDM_OK;
}

// sizes of the ABI structs, so a binding can assert its own layout (0: igemm desc, 1: wgrad desc, 2: rows,
// 3: linear job, 4: adam job)
// ring of timing-free events for sdm_stream_order (a wait captures the record that precedes it, so re-recording an
// event later does not disturb earlier waits)
#define SDM_ORDER_EVENTS 256
static hipEvent_t g_order_ev[SDM_ORDER_EVENTS];
static int g_order_next = -1;
static int g_order_dev = -1;

extern "C" int sdm_stream_order(sdm_stream_t from, sdm_stream_t to) {
  sdm_plan_taint("sdm_stream_order (a launch plan replays ONE stream's launches)");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return sdm_set_error(SDM_ERR_LAUNCH, "stream_order: hipGetDevice failed");
  if (g_order_next < 0 || dev != g_order_dev) {      // (re)build the ring on the current device
    for (int i = 0; i < SDM_ORDER_EVENTS; ++i)
      if (hipEventCreateWithFlags(&g_order_ev[i], hipEventDisableTiming) != hipSuccess)
        return sdm_set_error(SDM_ERR_LAUNCH, "stream_order: hipEventCreate failed");
    g_order_next = 0;
    g_order_dev = dev;
  }
  hipEvent_t ev = g_order_ev[g_order_next];
  g_order_next = (g_order_next + 1) % SDM_ORDER_EVENTS;
  if (hipEventRecord(ev, (hipStream_t)from) != hipSuccess) return sdm_set_error(SDM_ERR_LAUNCH, "stream_order: record failed");
  if (hipStreamWaitEvent((hipStream_t)to, ev, 0) != hipSuccess) return sdm_set_error(SDM_ERR_LAUNCH, "stream_order: wait failed");
  return SDM_OK;
}

extern "C" size_t sdm_struct_size(int which) {
  switch (which) {
    case 0: return sizeof(sdm_igemm_desc);
    case 1: return sizeof(sdm_wgrad_desc);
    case 2: return sizeof(sdm_rows);
    case 3: return sizeof(sdm_linear_job);
    case 4: return sizeof(sdm_adam_job);
    default: return 0;
  }
}

// ---- per-launch event timing ----------------------------------------------------------------
#include <stdlib.h>
#include <vector>
namespace {
struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; int M, N, K, taps, splitk; };
std::vector<ProfRec> g_prof;
int g_prof_cap = 0;
bool g_prof_armed = false;
hipEvent_t g_pending = nullptr;
}  // namespace

bool sdm_prof_on() { return g_prof_armed && (int)g_prof.size() < g_prof_cap; }
void sdm_prof_pre(hipStream_t st) {
  if (!sdm_prof_on()) return;
  if (hipEventCreate(&g_pending) != hipSuccess) { g_pending = nullptr; return; }
  (void)hipEventRecord(g_pending, st);
}
static int g_pM, g_pN, g_pK, g_pT, g_pS;
static double g_pB;
void sdm_prof_shape(int M, int N, int K, int taps, int splitk, double bytes) {
  g_pM = M; g_pN = N; g_pK = K; g_pT = taps; g_pS = splitk; g_pB = bytes;
}
void sdm_prof_post(hipStream_t st, int cls, double flops) {
  if (!g_pending) return;
  ProfRec r;
  r.a = g_pending;
  g_pending = nullptr;
  if (hipEventCreate(&r.b) != hipSuccess) { (void)hipEventDestroy(r.a); return; }
  (void)hipEventRecord(r.b, st);
  r.cls = cls;
  r.flops = flops;
  r.bytes = g_pB;
  r.M = g_pM; r.N = g_pN; r.K = g_pK; r.taps = g_pT; r.splitk = g_pS;
  g_prof.push_back(r);
}

extern "C" int sdm_prof_begin(int capacity) {
  SDM_CHECK_ARG(capacity > 0, "prof_begin: capacity");
  for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_prof.clear();
  g_prof.reserve(capacity);
  g_prof_cap = capacity;
  g_prof_armed = true;
  return SDM_OK;
}

// What an event pair adds to the launch it brackets: the same pair around an EMPTY kernel, median of `reps` launches on the
// given stream (the stream is synchronised).  The result still contains the empty kernel's own ~1-2 us.
__global__ void sdm_prof_null_kernel() {}
extern "C" double sdm_prof_bracket_ms(int reps, sdm_stream_t s) {
  if (reps < 1) reps = 1;
  if (reps > 1024) reps = 1024;
  hipStream_t st = (hipStream_t)s;
  std::vector<hipEvent_t> ev(2 * (size_t)reps);
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) return -1.0;
  for (int i = 0; i < 8; ++i) sdm_prof_null_kernel<<<dim3(1), dim3(64), 0, st>>>();
  for (int i = 0; i < reps; ++i) {
    (void)hipEventRecord(ev[2 * i], st);
    sdm_prof_null_kernel<<<dim3(1), dim3(64), 0, st>>>();
    (void)hipEventRecord(ev[2 * i + 1], st);
  }
  (void)hipStreamSynchronize(st);
  std::vector<float> t((size_t)reps, 0.f);
  for (int i = 0; i < reps; ++i) (void)hipEventElapsedTime(&t[i], ev[2 * i], ev[2 * i + 1]);
  for (auto& e : ev) (void)hipEventDestroy(e);
  std::sort(t.begin(), t.end());
  return (double)t[reps / 2];
}

extern "C" int sdm_prof_end(double* ms, long long* launches, double* flops, double* bytes) {
  SDM_CHECK_ARG(ms && launches && flops && bytes, "prof_end: null");
  g_prof_armed = false;
  for (int c = 0; c < SDM_PROF_CLASSES; ++c) { ms[c] = 0; launches[c] = 0; flops[c] = 0; bytes[c] = 0; }
  const char* dump = getenv("SDM_PROF_DUMP");     // optional per-launch CSV (tuning aid)
  FILE* df = dump ? fopen(dump, "w") : nullptr;
  if (df) fprintf(df, "cls,M,N,K,taps,splitk,us,tflops\n");
  for (auto& r : g_prof) {
    if (hipEventSynchronize(r.b) != hipSuccess) return sdm_set_error(SDM_ERR_LAUNCH, "prof_end: event sync failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess && r.cls >= 0 && r.cls < SDM_PROF_CLASSES) {
      ms[r.cls] += t;
      launches[r.cls] += 1;
      flops[r.cls] += r.flops;
      bytes[r.cls] += r.bytes;
      if (df) fprintf(df, "%d,%d,%d,%d,%d,%d,%.1f,%.1f\n", r.cls, r.M, r.N, r.K, r.taps, r.splitk, t * 1e3, r.flops / (t * 1e-3) / 1e12);
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  if (df) fclose(df);
  g_prof.clear();
  return SDM_OK;
}
