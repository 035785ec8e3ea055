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

extern "C" int sdm_version(void) { return 300; }

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

// sizes of the ABI structs, so a binding can assert its own layout (0: igemm desc, 1: wgrad desc, 2: rows,
// 3: linear job, 4: adam job)
// ring of timing-free events for sdm_stream_order (a wait captures the record that precedes it, so re-recording an
// event later does not disturb earlier waits)
#define SDM_ORDER_EVENTS 256
static hipEvent_t g_order_ev[SDM_ORDER_EVENTS];
static int g_order_next = -1;
static int g_order_dev = -1;

extern "C" int sdm_stream_order(sdm_stream_t from, sdm_stream_t to) {
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
