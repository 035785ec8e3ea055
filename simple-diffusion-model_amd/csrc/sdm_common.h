// Shared device/host helpers for libsdm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdm_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// bf16 storage helpers (bit patterns travel as unsigned short)
typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {   // round to nearest even (v_cvt_pk_bf16_f32 on gfx950)
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

int sdm_set_error(int code, const char* fmt, ...);
void sdm_plan_taint(const char* why);

#define SDM_CHECK_ARG(cond, ...)                         \
  do {                                                   \
    if (!(cond)) return sdm_set_error(SDM_ERR_ARG, __VA_ARGS__); \
  } while (0)

extern unsigned long long g_sdm_launches;   // kernel launches issued by the library (sdm_launch_count)

// Device-side failure word (sdm_device_error_init): one host-mapped, pinned 32-bit word.  A kernel whose bounded spin gave
// up stores a non-zero code there with a system-scope atomic; every later launch function of the library sees it on the
// host WITHOUT synchronising (a plain volatile read of pinned memory) and returns SDM_ERR_DEVICE through the ordinary
// error convention.  Null until sdm_device_error_init has run (kernels then skip the store).
extern volatile unsigned* g_sdm_errw_host;
extern unsigned* g_sdm_errw_dev;
int sdm_report_device_error(const char* where);
#define SDM_DEVERR_GN_COOP_TIMEOUT 1u
#define SDM_DEVERR_GN_BWD_COOP_TIMEOUT 2u
#define SDM_DEVERR_DP_TIMEOUT 3u

#define SDM_CHECK_LAUNCH(name)                                                         \
  do {                                                                                 \
    ++g_sdm_launches;                                                                  \
    hipError_t e_ = hipGetLastError();                                                 \
    if (e_ != hipSuccess) return sdm_set_error(SDM_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
    if (g_sdm_errw_host && *g_sdm_errw_host) return sdm_report_device_error(name);     \
  } while (0)

#define SDM_MEMSET0(ptr, bytes, st)                                              \
  do {                                                                           \
    sdm_plan_taint("hipMemsetAsync");                                            \
    if (hipMemsetAsync((ptr), 0, (bytes), (st)) != hipSuccess)                   \
      return sdm_set_error(SDM_ERR_LAUNCH, "hipMemsetAsync failed");             \
  } while (0)

// SDM_FAST_SIGMOID (defined by the bf16 translation units before this header): v_rcp_f32 (1 ulp) instead of the IEEE
// division sequence -- the result is rounded to bf16 right after; the fp32 path keeps the exact quotient.
#ifdef SDM_FAST_SIGMOID
__device__ __forceinline__ float sdm_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
#else
__device__ __forceinline__ float sdm_sigmoid(float z) { return 1.0f / (1.0f + __expf(-z)); }
#endif
__device__ __forceinline__ float sdm_silu(float z) { return z * sdm_sigmoid(z); }
__device__ __forceinline__ float sdm_dsilu(float z) {
  float s = sdm_sigmoid(z);
  return s * (1.0f + z * (1.0f - s));
}
__device__ __forceinline__ float sdm_act(float v, int act) {
  if (act == SDM_ACT_SILU) return sdm_silu(v);
  if (act == SDM_ACT_TANH) return tanhf(v);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// 64-bit element index / 32-bit extent in the pointwise kernels: the index fits 31 bits at every benchmark shape, and a
// 32-bit unsigned division is ~25 instructions against ~130 for the 64-bit sequence -- in the one-element-per-thread
// finishing / GroupNorm / affine passes of a sampling step the division WAS most of the kernel.
__device__ __forceinline__ long long sdm_idiv(long long a, long long b) {
  return (((unsigned long long)a | (unsigned long long)b) >> 31) == 0 ? (long long)((unsigned)a / (unsigned)b) : a / b;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- launch plan (sdm_plan_begin / sdm_plan_end / sdm_plan_run) -------------------------------------------------------
// Every kernel launch of the library goes through SDM_LAUNCH.  While a plan is being recorded the launch is ALSO appended
// to the plan (function, geometry, a copy of the argument values); sdm_plan_run then re-issues the whole sequence from C
// with one call from the host language -- the ~10 us of Python + ctypes per launch of an eagerly launched step (85
// launches per N=1 denoise step) become ~2 us of hipLaunchKernel each, and a replayed step costs what an eager one does
// on the GPU (hipGraph replay measured ~4 % slower than eager launching at N=1, tools/sample_timing.py).
bool sdm_plan_recording();
void sdm_plan_push(const void* fn, dim3 grid, dim3 block, size_t lds, void** args, const size_t* sizes, int nargs);
void sdm_plan_taint(const char* why);          // something ran that the plan cannot replay (a memset, ...)

template <typename T>
struct sdm_identity { typedef T type; };

template <typename... P>
static inline void sdm_launch(void (*kern)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t st,
                              typename sdm_identity<P>::type... a) {
  void* ptrs[sizeof...(P) > 0 ? sizeof...(P) : 1] = {(void*)&a...};
  if (sdm_plan_recording()) {
    const size_t sizes[sizeof...(P) > 0 ? sizeof...(P) : 1] = {sizeof(P)...};
    sdm_plan_push((const void*)kern, grid, block, lds, ptrs, sizes, (int)sizeof...(P));
  }
  (void)hipLaunchKernel((const void*)kern, grid, block, ptrs, lds, st);
}
#define SDM_LAUNCH(kern, grid, block, lds, st, ...) sdm_launch(kern, grid, block, lds, st, __VA_ARGS__)

// fused GroupNorm tails keep a small map's values in registers between their two sweeps (SDM_GN_HOLD=0: re-read, A/B)
inline int sdm_gn_hold() {
  static const char* e = getenv("SDM_GN_HOLD");
  return !(e && e[0] == '0');
}

// ---- optional per-launch event timing (sdm_prof_begin / sdm_prof_end) ----
bool sdm_prof_on();
void sdm_prof_pre(hipStream_t st);
void sdm_prof_post(hipStream_t st, int cls, double flops);
void sdm_prof_shape(int M, int N, int K, int taps, int splitk, double bytes);
