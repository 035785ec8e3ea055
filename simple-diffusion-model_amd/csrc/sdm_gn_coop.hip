// Fused conv-block tail (see gn_apply_fused_kernel in sdm_pointwise.hip) for the layers where ONE workgroup per (image,
// group) is the wrong shape: at 64x64 / 32x32 a GroupNorm group is 4-8 channels of 1024-4096 pixel rows, i.e. 16-32 bytes
// out of every 512-byte row, read twice by 32 workgroups (profiles/r03_sample_n1_kernel_stats.csv: 23-33 us per launch
// at N=1, a sixth of the whole denoise step together).
//
// Here a workgroup owns a SLAB of whole rows (coalesced 16-byte accesses, all channels), keeps its elements in registers
// (one sweep over memory instead of two), and the slabs of an image meet ONCE to exchange the per-group sums:
//   partial sums (fp32 in the lane, fp64 from the wave on) -> LDS -> fp64 atomic adds into sums[image][group][2]
//   -> every adding lane drains its atomics (s_waitcnt vmcnt(0)) -> workgroup barrier -> one lane bumps arrive[image]
//   -> one lane polls arrive[image] (agent-scope relaxed loads, s_sleep) until every slab of the image has arrived
//   -> the sums are read back with agent-scope atomic loads (8-byte atomics on both sides: no cache holds them)
//   -> apply from registers, store; the last slab to DEPART zeroes the sums and both counters for the next launch.
// Placement-independent: nothing depends on which CU / XCD a slab runs on.  Residency: the host sizes the grid against the
// occupancy query (half of occupancy x CU count of THIS device; anything larger is declined and the caller runs the
// one-workgroup-per-group kernel), and an image's slabs are consecutive block ids, so on an otherwise idle device every slab is
// resident and the poll ends.  That argument is per grid, not per device -- two cooperative launches on two streams can each
// hold part of their slabs resident -- so EVERY poll is bounded: it gives up after `timeout` ticks of the 100 MHz
// s_memrealtime clock (2 s by default; legitimate waits are kernels of other streams draining, i.e. milliseconds).  A slab
// that gave up writes NaN instead of results, reports SDM_DEVERR_* through the host-mapped failure word
// (sdm_device_error_init) and still departs, so the last slab leaves the exchange state zero as after a good launch.
#include <stdlib.h>

#include "sdm_common.h"

#define GC_T 1024
#define GC_MAXV 4          // float4 / bf16x4 elements a lane keeps
#define GC_MAXG 64         // groups

struct GnCoopArgs {
  const void* src; long long ld_src; int raw; float alpha; const float* bias; int act;
  int HW, C, groups; float eps;
  const float* gamma; const float* beta; const float* mod; int mod_ns;
  const void* x0; long long ldx0; const void* x1; long long ldx1; int csplit;
  void* out; long long ldo;
  int slab_rows, lc4, lq4;       // rows per slab; log2(C/4); log2(channel quads per group)
  double* sums;                  // [images][GC_MAXG][2], zero between launches
  unsigned* ctr;                 // [images][2] arrive / depart, zero between launches
  unsigned* errw;                // host-mapped failure word (may be null)
  unsigned long long timeout;    // s_memrealtime ticks (100 MHz) a slab waits for the others
  unsigned expect_extra;         // test hook: arrivals expected beyond the grid's slabs (forces the give-up path)
};

// One lane waits until *ctr >= need; false when `timeout` ticks passed first.  Relaxed agent-scope loads (never cached).
__device__ __forceinline__ bool gc_wait_arrivals(unsigned* ctr, unsigned need, unsigned long long timeout) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
    if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) return false;
    __builtin_amdgcn_s_sleep(2);
  }
}

template <bool BF>
__global__ __launch_bounds__(GC_T) void gn_apply_coop_kernel(const GnCoopArgs a) {
  __shared__ double ls[2][GC_MAXG];
  __shared__ float st[2][GC_MAXG];
  __shared__ int s_ok;
  const int tid = threadIdx.x, lane = tid & 63;
  const int n = blockIdx.y, nsl = gridDim.x;
  const int C4 = a.C >> 2, q4 = 1 << a.lq4;
  const int r0 = blockIdx.x * a.slab_rows;
  const int nrows = min(a.slab_rows, a.HW - r0);
  const long long row0 = (long long)n * a.HW + r0;
  const int quad = tid & (C4 - 1);             // the same for every element of this lane (1024 % C4 == 0)
  const int c = quad * 4, g = quad >> a.lq4;
  const int rstep = GC_T >> a.lc4;             // rows between two elements of a lane
  const int rfirst = tid >> a.lc4;
  if (tid < 2 * GC_MAXG) (&ls[0][0])[tid] = 0.0;
  __syncthreads();

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 b4 = (a.raw && a.bias) ? *(const f32x4*)(a.bias + c) : zero4;
  f32x4 v[GC_MAXV];
  float p1 = 0.f, p2 = 0.f;
#pragma unroll
  for (int k = 0; k < GC_MAXV; ++k) {
    const int r = rfirst + k * rstep;
    v[k] = zero4;
    if (r < nrows) {
      const long long idx = (row0 + r) * a.ld_src + c;
      if (a.raw) {
        f32x4 t = *(const f32x4*)((const float*)a.src + idx);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          t[i] = sdm_act(t[i] * a.alpha + b4[i], a.act);
          if (BF) t[i] = bf2f(f2bf(t[i]));           // statistics of the values the next layer sees
        }
        v[k] = t;
        *(f32x4*)((float*)a.src + idx) = zero4;        // the split-K scratch goes back zero-filled
      } else if (BF) {
        const u16x4 u = *(const u16x4*)((const bf16_t*)a.src + idx);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[k][i] = bf2f(u[i]);
      } else {
        v[k] = *(const f32x4*)((const float*)a.src + idx);
      }
      p1 += (v[k].x + v[k].y) + (v[k].z + v[k].w);
      p2 += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
    }
  }
  // lanes of one wave that share a group: the q4 neighbouring quads, and (C4 < 64) the lanes C4 apart
  for (int o = 1; o < q4; o <<= 1) {
    p1 += __shfl_xor(p1, o, 64);
    p2 += __shfl_xor(p2, o, 64);
  }
  for (int o = C4; o < 64; o <<= 1) {
    p1 += __shfl_xor(p1, o, 64);
    p2 += __shfl_xor(p2, o, 64);
  }
  if ((lane & (q4 - 1)) == 0 && (C4 >= 64 || lane < C4)) {
    atomicAdd(&ls[0][g], (double)p1);
    atomicAdd(&ls[1][g], (double)p2);
  }
  __syncthreads();
  double* gs = a.sums + (long long)n * GC_MAXG * 2;
  unsigned* ctr = a.ctr + n * 2;
  if (tid < a.groups) {
    atomicAdd(gs + tid * 2, ls[0][tid]);
    atomicAdd(gs + tid * 2 + 1, ls[1][tid]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this lane's adds are performed before the arrival below
  }
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok = gc_wait_arrivals(ctr, (unsigned)nsl + a.expect_extra, a.timeout);
    if (!ok && a.errw) __hip_atomic_store(a.errw, SDM_DEVERR_GN_COOP_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    s_ok = ok;
  }
  __syncthreads();
  const bool ok = s_ok != 0;
  if (tid < a.groups) {
    const double s1 = __hip_atomic_load(gs + tid * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double s2 = __hip_atomic_load(gs + tid * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double total = (double)a.HW * (a.C / a.groups);
    const double mean = s1 / total;
    double var = s2 / total - mean * mean;
    if (var < 0.0) var = 0.0;
    st[0][tid] = (float)mean;
    st[1][tid] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
  __syncthreads();
  const float mu = st[0][g], rstd = st[1][g];
  const f32x4 ga = *(const f32x4*)(a.gamma + c), be = *(const f32x4*)(a.beta + c);
  f32x4 sc, sh;
  if (a.mod) {
    const f32x4 sm = *(const f32x4*)(a.mod + (long long)n * a.mod_ns + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // the two-step form of sdm_gn_affine + sdm_affine_residual: scale*v + shift
      sc[i] = rstd * ga[i] * sm[i];
      sh[i] = (be[i] - mu * rstd * ga[i]) * sm[i] + sm[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sc[i] = rstd * ga[i];
      sh[i] = be[i] - mu * rstd * ga[i];
    }
  }
#pragma unroll
  for (int k = 0; k < GC_MAXV; ++k) {
    const int r = rfirst + k * rstep;
    if (r >= nrows) continue;
    const long long row = row0 + r;
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = ok ? v[k][i] * sc[i] + sh[i] : __builtin_nanf("");     // a failed exchange is visible
    if (a.x0) {
      if (BF) {
        const u16x4 u = (c < a.csplit) ? *(const u16x4*)((const bf16_t*)a.x0 + row * a.ldx0 + c)
                                       : *(const u16x4*)((const bf16_t*)a.x1 + row * a.ldx1 + (c - a.csplit));
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] += bf2f(u[i]);
      } else {
        o += (c < a.csplit) ? *(const f32x4*)((const float*)a.x0 + row * a.ldx0 + c)
                            : *(const f32x4*)((const float*)a.x1 + row * a.ldx1 + (c - a.csplit));
      }
    }
    if (BF) {
      u16x4 u;
#pragma unroll
      for (int i = 0; i < 4; ++i) u[i] = f2bf(o[i]);
      *(u16x4*)((bf16_t*)a.out + row * a.ldo + c) = u;
    } else {
      *(f32x4*)((float*)a.out + row * a.ldo + c) = o;
    }
  }
  // departure: the last slab of the image leaves the exchange state zero for the next launch (every slab has read the sums:
  // it departs after its reads)
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (unsigned)nsl - 1) {
      for (int i = 0; i < a.groups * 2; ++i) __hip_atomic_store(gs + i, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static int ilog2x(long v) {
  if (v <= 0 || (v & (v - 1)) != 0) return -1;
  int l = 0;
  while ((1L << l) < v) ++l;
  return l;
}

// Workgroups of `kern` (T threads, its static LDS) that are resident together on the CURRENT device, halved for margin:
// occupancy query x CU count / 2, additionally limited by the 32 waves a CU holds.  Cached per (kernel, device).
static int gc_resident_cap(const void* kern, int T, int slot) {
  static int cache[8][16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (cache[slot][dev] == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, T, 0) != hipSuccess) per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    const int by_waves = 2048 / T;
    if (per_cu > by_waves) per_cu = by_waves;
    cache[slot][dev] = per_cu > 0 && cus > 0 ? (per_cu * cus) / 2 : -1;
  }
  return cache[slot][dev] > 0 ? cache[slot][dev] : 0;
}
// bound of every inter-slab wait, in 100 MHz ticks (default 2 s; SDM_GN_COOP_TIMEOUT_MS overrides), and the test hook
// SDM_GN_COOP_TEST_EXTRA=n (a slab then expects n arrivals that never come: the give-up path, bounded by the timeout)
static unsigned long long gc_timeout_ticks() {
  const char* e = getenv("SDM_GN_COOP_TIMEOUT_MS");
  const double ms = e ? atof(e) : 2000.0;
  return (unsigned long long)((ms > 0.0 ? ms : 2000.0) * 1e5);
}
static unsigned gc_expect_extra() {
  const char* e = getenv("SDM_GN_COOP_TEST_EXTRA");
  return e ? (unsigned)atoi(e) : 0u;
}

#define GC_MAX_IMAGES 256
extern "C" size_t sdm_gn_apply_coop_state_bytes(void) {
  return (size_t)GC_MAX_IMAGES * GC_MAXG * 2 * sizeof(double) + (size_t)GC_MAX_IMAGES * 2 * sizeof(unsigned);
}

extern "C" int sdm_gn_apply_coop(const void* src, int64_t ld_src, int src_raw, float alpha, const float* bias, int act, int images,
                                 int HW, int C, int groups, float eps, const float* gamma, const float* beta, const float* mod,
                                 int mod_nstride, const void* x0, int64_t ldx0, const void* x1, int64_t ldx1, int csplit, void* out,
                                 int64_t ldo, int bf16, void* state, size_t state_bytes, sdm_stream_t s) {
  SDM_CHECK_ARG(src && gamma && beta && out && images > 0 && HW > 0 && groups > 0 && C % groups == 0, "gn_apply_coop: bad args");
  SDM_CHECK_ARG(state && state_bytes >= sdm_gn_apply_coop_state_bytes() && aligned16(state), "gn_apply_coop: state buffer");
  SDM_CHECK_ARG(x0 == nullptr || x1 != nullptr || csplit >= C, "gn_apply_coop: csplit < C without second source");
  SDM_CHECK_ARG(act >= 0 && act <= 2, "gn_apply_coop: bad act");
  const int cpg = C / groups;
  const int lc4 = ilog2x(C / 4), lq4 = cpg % 4 == 0 ? ilog2x(cpg / 4) : -1;
  // applicability (else the caller runs the one-workgroup-per-group kernel): power-of-two quads per row and per group, a
  // group's quads inside one wave (the shuffle reduction), a row no wider than the workgroup, at least two slabs per image
  // and no more workgroups than half of what this device keeps resident
  if (C % 4 != 0 || lc4 < 0 || lq4 < 0 || (cpg / 4) > 64 || (C / 4) > GC_T || groups > GC_MAXG || images > GC_MAX_IMAGES)
    return SDM_NOT_APPLICABLE;
  const int cap = gc_resident_cap(bf16 ? (const void*)gn_apply_coop_kernel<true> : (const void*)gn_apply_coop_kernel<false>, GC_T, bf16 ? 1 : 0);
  const long long quads = (long long)HW * (C / 4);
  long long nsl = 1;
  while (nsl * GC_T * GC_MAXV < quads) nsl <<= 1;
  // The exchange costs ~8 us whatever the size (fp64 atomics, the arrival poll and the read-back are each a round trip
  // to the memory side: measured 10-14 us per launch from 2048 to 131072 quads against 6-33 us for the per-group kernel),
  // so it is used from 32 slabs per image on -- the 64x64 maps -- and declined below (SDM_GN_COOP_MIN_SLABS overrides).
  static const char* ms = getenv("SDM_GN_COOP_MIN_SLABS");
  const long long min_slabs = ms ? atoll(ms) : 32;
  if (nsl < 2 || nsl < min_slabs || nsl * images > cap) return SDM_NOT_APPLICABLE;
  const int slab_rows = (int)((HW + nsl - 1) / nsl);
  if ((long long)slab_rows * (C / 4) > (long long)GC_T * GC_MAXV) return SDM_NOT_APPLICABLE;
  const size_t esz = bf16 ? 2 : 4;
  auto al = [](const void* q, size_t a_) { return (((uintptr_t)q) & (a_ - 1)) == 0; };
  SDM_CHECK_ARG(al(src, src_raw ? 16 : 4 * esz) && ld_src % 4 == 0 && al(out, 4 * esz) && ldo % 4 == 0 && aligned16(gamma) &&
                    aligned16(beta) && (mod == nullptr || (aligned16(mod) && mod_nstride % 4 == 0)) &&
                    (bias == nullptr || aligned16(bias)) && (x0 == nullptr || (al(x0, 4 * esz) && ldx0 % 4 == 0 && csplit % 4 == 0)) &&
                    (x1 == nullptr || (al(x1, 4 * esz) && ldx1 % 4 == 0)),
                "gn_apply_coop: rows and tables must be aligned to four elements");
  GnCoopArgs a;
  a.src = src; a.ld_src = ld_src; a.raw = src_raw; a.alpha = alpha; a.bias = bias; a.act = act;
  a.HW = HW; a.C = C; a.groups = groups; a.eps = eps;
  a.gamma = gamma; a.beta = beta; a.mod = mod; a.mod_ns = mod_nstride;
  a.x0 = x0; a.ldx0 = ldx0; a.x1 = x1; a.ldx1 = ldx1; a.csplit = x1 ? csplit : (1 << 30);
  a.out = out; a.ldo = ldo;
  a.slab_rows = slab_rows; a.lc4 = lc4; a.lq4 = lq4;
  a.sums = (double*)state;
  a.ctr = (unsigned*)((char*)state + (size_t)GC_MAX_IMAGES * GC_MAXG * 2 * sizeof(double));
  a.errw = g_sdm_errw_dev; a.timeout = gc_timeout_ticks(); a.expect_extra = gc_expect_extra();
  dim3 grid((unsigned)nsl, images);
  if (bf16) SDM_LAUNCH(gn_apply_coop_kernel<true>, grid, dim3(GC_T), 0, (hipStream_t)s, a);
  else SDM_LAUNCH(gn_apply_coop_kernel<false>, grid, dim3(GC_T), 0, (hipStream_t)s, a);
  SDM_CHECK_LAUNCH("gn_apply_coop");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// AdaGN backward in ONE launch (sdm_gn_adagn_bwd runs three: channel sums, coefficients, apply).  OPT-IN experiment
// (ops.GN_COOP_BWD / SDM_GN_COOP_BWD=1): correct, but inside the fp32 training step it loses to the three launches -- the
// slabs must all be resident before any can finish and the weight-gradient stream keeps the CUs full (64x64 maps: 118 us
// against 69 us; small maps tie at ~16 us), profiles/r03_gn_bwd_coop.txt.  Same slab ownership and the same single
// exchange as the forward kernel above:
//   a lane keeps its dout / y / z elements (all of one channel quad) in registers;
//   per-channel partials (sum dout, sum dout*y) meet in LDS; thread t < C sends channel t's pair to ws[image][t][2] with
//   fp32 atomics (nobody waits for those) and the lanes of a group reduce  P = sum_c s_c gamma_c S1_c,  Q = sum_c s_c gamma_c S2_c
//   into TWO fp64 atomics per group: that pair is all the apply needs (k1, k2 of sdm_gn_adagn_bwd's coefficient pass);
//   arrival / poll / read-back as above; apply from registers;
//   the LAST slab of an image to depart turns the finished channel sums into dmod / dgamma / dbeta (what the
//   coefficient pass emitted) and leaves ws, the group sums and the counters zero.
// An image that fits one slab skips the exchange altogether (no atomics, no polling).
// --------------------------------------------------------------------------------------------
struct GnBwdCoopArgs {
  const void* dout; long long lddo; const void* y; long long ldy; const void* z; long long ldz;
  int HW, C, groups;
  const float* gamma; const float* beta; const float* mod; int mod_ns;
  const float* mean; const float* rstd;
  void* dz; long long lddz;
  float* dmod; float* dgamma; float* dbeta;
  float* ws;                     // [images][C][2], zero between launches
  int slab_rows, lc4, lq4;
  double* sums; unsigned* ctr;   // the forward kernel's exchange state
  unsigned* errw; unsigned long long timeout; unsigned expect_extra;     // as in GnCoopArgs
};

template <bool BF>
__device__ __forceinline__ f32x4 gc_load4(const void* p, long long idx) {
  if (BF) {
    const u16x4 u = *(const u16x4*)((const bf16_t*)p + idx);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = bf2f(u[i]);
    return v;
  }
  return *(const f32x4*)((const float*)p + idx);
}

template <bool BF, int T>
__global__ __launch_bounds__(T) void gn_bwd_coop_kernel(const GnBwdCoopArgs a) {
  __shared__ float lsC[2][GC_T];         // this slab's per-channel partials (C <= GC_T = 1024)
  __shared__ double lg[2][GC_MAXG];      // single-slab images: the group pair without the trip through memory
  __shared__ float st[2][GC_MAXG];       // k1, k2 per group
  __shared__ int s_last;
  __shared__ int s_ok;
  const int tid = threadIdx.x, lane = tid & 63;
  const int n = blockIdx.y, nsl = gridDim.x;
  const int C4 = a.C >> 2;
  const int r0 = blockIdx.x * a.slab_rows;
  const int nrows = min(a.slab_rows, a.HW - r0);
  const long long row0 = (long long)n * a.HW + r0;
  const int quad = tid & (C4 - 1);
  const int c = quad * 4, g = quad >> a.lq4;
  const int rstep = T >> a.lc4, rfirst = tid >> a.lc4;
  const int cpg = a.C / a.groups;
  for (int i = tid; i < 2 * GC_T; i += T) (&lsC[0][0])[i] = 0.f;
  __syncthreads();

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 d[GC_MAXV], yv[GC_MAXV], zv[GC_MAXV];
  f32x4 s1 = zero4, s2 = zero4;
#pragma unroll
  for (int k = 0; k < GC_MAXV; ++k) {
    const int r = rfirst + k * rstep;
    d[k] = zero4; yv[k] = zero4; zv[k] = zero4;
    if (r < nrows) {
      d[k] = gc_load4<BF>(a.dout, (row0 + r) * a.lddo + c);
      yv[k] = gc_load4<BF>(a.y, (row0 + r) * a.ldy + c);
      if (a.z) zv[k] = gc_load4<BF>(a.z, (row0 + r) * a.ldz + c);
      s1 += d[k];
      s2 += d[k] * yv[k];
    }
  }
  // lanes of a wave that hold the same quad (C4 < 64): fold before touching LDS
  for (int o = C4; o < 64; o <<= 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s1[i] += __shfl_xor(s1[i], o, 64);
      s2[i] += __shfl_xor(s2[i], o, 64);
    }
  }
  if (C4 >= 64 || lane < C4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      atomicAdd(&lsC[0][c + i], s1[i]);
      atomicAdd(&lsC[1][c + i], s2[i]);
    }
  }
  __syncthreads();
  double* gs = a.sums + (long long)n * GC_MAXG * 2;
  unsigned* ctr = a.ctr + n * 2;
  float* wsn = a.ws + (long long)n * a.C * 2;
  for (int t = tid; t < a.C; t += T) {   // whole waves: C is a multiple of 128 (host check), so the shuffles below are uniform
    const float S1 = lsC[0][t], S2 = lsC[1][t];
    if (nsl > 1) {
      atomicAdd(wsn + t * 2, S1);
      atomicAdd(wsn + t * 2 + 1, S2);
    }
    const float sg = (a.mod ? a.mod[(long long)n * a.mod_ns + t] : 1.f) * a.gamma[t];
    double p = (double)sg * (double)S1, q = (double)sg * (double)S2;
    for (int o = 1; o < cpg; o <<= 1) {
      p += __shfl_xor(p, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    if ((t & (cpg - 1)) == 0) {
      const int gg = t / cpg;
      if (nsl > 1) {
        atomicAdd(gs + gg * 2, p);
        atomicAdd(gs + gg * 2 + 1, q);
      } else {
        lg[0][gg] = p;
        lg[1][gg] = q;
      }
    }
  }
  if (nsl > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) s_ok = 1;
  if (nsl > 1) {
    if (tid == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool okw = gc_wait_arrivals(ctr, (unsigned)nsl + a.expect_extra, a.timeout);
      if (!okw && a.errw) __hip_atomic_store(a.errw, SDM_DEVERR_GN_BWD_COOP_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      s_ok = okw;
    }
  }
  __syncthreads();
  const bool ok = s_ok != 0;
  if (tid < a.groups) {
    double P, Q;
    if (nsl > 1) {
      P = __hip_atomic_load(gs + tid * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      Q = __hip_atomic_load(gs + tid * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      P = lg[0][tid];
      Q = lg[1][tid];
    }
    const double mu = a.mean[n * a.groups + tid], rs = a.rstd[n * a.groups + tid];
    const double cnt = (double)a.HW * (double)cpg;
    const double A = P / cnt, B = rs * (Q - mu * P) / cnt;
    st[0][tid] = (float)(-rs * rs * B);
    st[1][tid] = (float)(-rs * A + mu * rs * rs * B);
  }
  __syncthreads();
  {
    const float rs = a.rstd[n * a.groups + g];
    const float k1 = st[0][g], k2 = st[1][g];
    const f32x4 ga = *(const f32x4*)(a.gamma + c);
    f32x4 k0;
    if (a.mod) {
      const f32x4 sm = *(const f32x4*)(a.mod + (long long)n * a.mod_ns + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) k0[i] = rs * sm[i] * ga[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) k0[i] = rs * ga[i];
    }
#pragma unroll
    for (int k = 0; k < GC_MAXV; ++k) {
      const int r = rfirst + k * rstep;
      if (r >= nrows) continue;
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = k0[i] * d[k][i] + k1 * yv[k][i] + k2;
        if (a.z) v[i] *= sdm_dsilu(zv[k][i]);
        if (!ok) v[i] = __builtin_nanf("");
      }
      const long long idx = (row0 + r) * a.lddz + c;
      if (BF) {
        u16x4 u;
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = f2bf(v[i]);
        *(u16x4*)((bf16_t*)a.dz + idx) = u;
      } else {
        *(f32x4*)((float*)a.dz + idx) = v;
      }
    }
  }
  // departure; the last slab of the image finishes the per-channel outputs and resets the state
  __syncthreads();
  if (nsl > 1) {
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == (unsigned)nsl - 1);
    }
    __syncthreads();
    if (!s_last) return;
  }
  for (int t = tid; t < a.C; t += T) {
    float S1, S2;
    if (nsl > 1) {
      S1 = __hip_atomic_load(wsn + t * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      S2 = __hip_atomic_load(wsn + t * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(wsn + t * 2, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(wsn + t * 2 + 1, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      S1 = lsC[0][t];
      S2 = lsC[1][t];
    }
    const int gg = t / cpg;
    const float mu = a.mean[n * a.groups + gg], rs = a.rstd[n * a.groups + gg];
    const float sm = a.mod ? a.mod[(long long)n * a.mod_ns + t] : 1.f;
    const float ga = a.gamma[t], be = a.beta[t];
    const float xh = rs * (S2 - mu * S1);
    if (a.dmod) a.dmod[(long long)n * a.C + t] = ga * xh + (a.mod ? (be + 1.f) : be) * S1;
    if (a.dgamma) atomicAdd(&a.dgamma[t], sm * xh);
    if (a.dbeta) atomicAdd(&a.dbeta[t], sm * S1);
  }
  if (nsl > 1 && tid < a.groups * 2) __hip_atomic_store(gs + tid, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (nsl > 1 && tid == 0) {
    __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

extern "C" int sdm_gn_bwd_coop(const void* dout, int64_t lddo, const void* y, int64_t ldy, const void* z, int64_t ldz, int images,
                               int HW, int C, int groups, const float* gamma, const float* beta, const float* mod,
                               int mod_nstride, const float* mean, const float* rstd, void* dz, int64_t lddz, float* dmod,
                               float* dgamma, float* dbeta, float* ws, int bf16, void* state, size_t state_bytes,
                               sdm_stream_t s) {
  SDM_CHECK_ARG(dout && y && gamma && beta && mean && rstd && dz && ws, "gn_bwd_coop: null pointer");
  SDM_CHECK_ARG(images > 0 && HW > 0 && groups > 0 && C % groups == 0, "gn_bwd_coop: bad dims");
  SDM_CHECK_ARG(state && state_bytes >= sdm_gn_apply_coop_state_bytes() && aligned16(state), "gn_bwd_coop: state buffer");
  const int cpg = C / groups;
  const int lc4 = ilog2x(C / 4), lq4 = cpg % 4 == 0 ? ilog2x(cpg / 4) : -1;
  // envelope (else the caller runs sdm_gn_adagn_bwd): power-of-two quads per row and per group, a channel per thread in the
  // per-channel steps (C <= 1024, whole 128-lane pairs of waves), a group inside one wave, at most 256 workgroups in all
  if (C % 128 != 0 || lc4 < 0 || lq4 < 0 || C > GC_T || cpg > 64 || groups > GC_MAXG || images > GC_MAX_IMAGES) return SDM_NOT_APPLICABLE;
  // workgroup size: 256 threads by default -- a 1024-thread workgroup (4 waves x 80 registers per SIMD) does not fit on a CU
  // beside a weight-gradient workgroup of the side stream and waits for whole CUs to drain (measured: the fp32 step got
  // 0.15 ms SLOWER with it); SDM_GN_BWD_T=1024 selects the large form
  static const char* ts = getenv("SDM_GN_BWD_T");
  const int T = (ts && atoi(ts) == 1024) ? 1024 : 256;
  if (C / 4 > T) return SDM_NOT_APPLICABLE;
  const long long quads = (long long)HW * (C / 4);
  long long nsl = 1;
  while (nsl * T * GC_MAXV < quads) nsl <<= 1;
  const void* kfn = T == 1024 ? (bf16 ? (const void*)gn_bwd_coop_kernel<true, 1024> : (const void*)gn_bwd_coop_kernel<false, 1024>)
                              : (bf16 ? (const void*)gn_bwd_coop_kernel<true, 256> : (const void*)gn_bwd_coop_kernel<false, 256>);
  if (nsl * images > gc_resident_cap(kfn, T, 2 + (T == 1024 ? 2 : 0) + (bf16 ? 1 : 0))) return SDM_NOT_APPLICABLE;     // all resident at once
  const int slab_rows = (int)((HW + nsl - 1) / nsl);
  if ((long long)slab_rows * (C / 4) > (long long)T * GC_MAXV) return SDM_NOT_APPLICABLE;
  const size_t esz = bf16 ? 2 : 4;
  auto al = [](const void* q, size_t a_) { return (((uintptr_t)q) & (a_ - 1)) == 0; };
  SDM_CHECK_ARG(al(dout, 4 * esz) && lddo % 4 == 0 && al(y, 4 * esz) && ldy % 4 == 0 && al(dz, 4 * esz) && lddz % 4 == 0 &&
                    (z == nullptr || (al(z, 4 * esz) && ldz % 4 == 0)) && aligned16(gamma) && aligned16(beta) &&
                    (mod == nullptr || (aligned16(mod) && mod_nstride % 4 == 0)),
                "gn_bwd_coop: rows and tables must be aligned to four elements");
  GnBwdCoopArgs a;
  a.dout = dout; a.lddo = lddo; a.y = y; a.ldy = ldy; a.z = z; a.ldz = ldz;
  a.HW = HW; a.C = C; a.groups = groups;
  a.gamma = gamma; a.beta = beta; a.mod = mod; a.mod_ns = mod_nstride;
  a.mean = mean; a.rstd = rstd;
  a.dz = dz; a.lddz = lddz; a.dmod = dmod; a.dgamma = dgamma; a.dbeta = dbeta; a.ws = ws;
  a.slab_rows = slab_rows; a.lc4 = lc4; a.lq4 = lq4;
  a.sums = (double*)state;
  a.ctr = (unsigned*)((char*)state + (size_t)GC_MAX_IMAGES * GC_MAXG * 2 * sizeof(double));
  a.errw = g_sdm_errw_dev; a.timeout = gc_timeout_ticks(); a.expect_extra = gc_expect_extra();
  dim3 grid((unsigned)nsl, images);
  if (T == 1024) {
    if (bf16) SDM_LAUNCH((gn_bwd_coop_kernel<true, 1024>), grid, dim3(1024), 0, (hipStream_t)s, a);
    else SDM_LAUNCH((gn_bwd_coop_kernel<false, 1024>), grid, dim3(1024), 0, (hipStream_t)s, a);
  } else {
    if (bf16) SDM_LAUNCH((gn_bwd_coop_kernel<true, 256>), grid, dim3(256), 0, (hipStream_t)s, a);
    else SDM_LAUNCH((gn_bwd_coop_kernel<false, 256>), grid, dim3(256), 0, (hipStream_t)s, a);
  }
  SDM_CHECK_LAUNCH("gn_bwd_coop");
  return SDM_OK;
}
