// Attention core for SHORT sequences (S <= 256 tokens: the 16x16 / 8x8 / 4x4 feature maps of a 64x64 or 128x128 U-Net),
// custom_layers.py:127-163 of the reference:
//     scores = (q k^T) * d_k^-0.5 ; attn = softmax over the QUERY axis (dim=1, :147) ; res = attn v
// At these sizes the three generic launches (scores GEMM [+ split-K finishing pass], column softmax, P.V GEMM [+ finishing
// pass]; bf16 rows: + a V transpose) are pure fixed cost: 4 MFLOP of work behind ~40 us of dependent launches, four times
// per denoise step.  Here the core is two launches, both on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32: 16-row
// tiles, so a 16-token sequence wastes nothing), operands straight from global / L2 into MFMA fragments (no LDS staging:
// every operand is a few hundred KB and each fragment is used by one wave), for fp32 or bf16 rows:
//   attn_scores_small : one workgroup per block of 16 KEYS holds the scores of ALL queries against them (S x 16) in
//                       accumulators, reduces max / sum over the queries on chip and writes the probabilities P once;
//   attn_pv_small     : res tile (32 queries x 64 channels) = P[32 x S] V[S x 64], V read k-major (no transpose copy).
// P has the layout of the generic path ([batch][S][ldp], fp32 or bf16 with zero-filled pad columns), so the training
// backward (which keeps P) is unchanged.
#include "sdm_common.h"

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ f32x4_t load4(const T* p);
template <>
__device__ __forceinline__ f32x4_t load4<float>(const float* p) {
  return *(const f32x4_t*)p;
}
template <>
__device__ __forceinline__ f32x4_t load4<bf16_t>(const bf16_t* p) {
  const u16x4 v = *(const u16x4*)p;
  f32x4_t r = {bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
  return r;
}
template <typename T>
__device__ __forceinline__ float load1(const T* p);
template <>
__device__ __forceinline__ float load1<float>(const float* p) {
  return *p;
}
template <>
__device__ __forceinline__ float load1<bf16_t>(const bf16_t* p) {
  return bf2f(*p);
}
template <typename T>
__device__ __forceinline__ void store1(T* p, float v);
template <>
__device__ __forceinline__ void store1<float>(float* p, float v) {
  *p = v;
}
template <>
__device__ __forceinline__ void store1<bf16_t>(bf16_t* p, float v) {
  *p = f2bf(v);
}

#define AS_KB 16          // keys per workgroup
#define AS_MAXS 256       // queries held on chip
#define AS_WAVES 16         // 1024 threads: one 16-query tile per wave at S = 256, four waves per SIMD to hide load latency

// v_mfma_f32_16x16x4_f32: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; D: col = l&15, row = 4*(l>>4) + reg.
// Each lane loads 4 consecutive reduction elements (16 bytes of fp32, 8 of bf16) at k0 + 4*(l>>4) and feeds element s to
// MFMA s: the same permutation of the reduction index on both operands, so the sum is unchanged.
template <typename T>
__global__ __launch_bounds__(64 * AS_WAVES) void attn_scores_small_kernel(const T* __restrict__ qkv, long long ld, int S, int heads,
                                                                           int dk, float scale, T* __restrict__ P, long long ldp) {
  __shared__ float xs[AS_MAXS][AS_KB + 1];            // scores of all queries against this block's keys
  __shared__ float s_m[AS_KB], s_inv[AS_KB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, img = b / heads, head = b % heads;
  const int j0 = blockIdx.x * AS_KB;
  const T* qb = qkv + (long long)img * S * ld + (long long)head * 3 * dk;      // q chunk of this (image, head)
  const T* kb = qb + dk;
  const int mt = (S + 15) >> 4;                       // 16-query tiles
  // work split over the 8 waves: tiles first, then slices of the reduction (head width)
  int mtp = 1;
  while (mtp < mt && mtp < AS_WAVES) mtp <<= 1;
  int ksplit = AS_WAVES / mtp;
  while (ksplit > 1 && (dk / ksplit) % 16 != 0) ksplit >>= 1;
  const int tile0 = wave % mtp, ks = wave / mtp;
  const bool active = ks < ksplit;
  const int kper = dk / ksplit, kbeg = ks * kper;
  const int li = lane & 15, kg = lane >> 4;

  for (int e = tid; e < AS_MAXS * (AS_KB + 1); e += 64 * AS_WAVES) (&xs[0][0])[e] = 0.f;
  __syncthreads();

  if (active) {
    const int key = min(j0 + li, S - 1);
    const T* kp = kb + (long long)key * ld + kbeg + 4 * kg;
    for (int t = tile0; t < mt; t += mtp) {
      const int q = min(t * 16 + li, S - 1);
      const T* qp = qb + (long long)q * ld + kbeg + 4 * kg;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      // 64 reduction elements per trip (eight 16-byte loads -> 16 MFMAs); the loads of trip k+1 are issued before the
      // MFMAs of trip k (two register sets), so a wave waits for memory once, not once per trip
      f32x4_t a[2][4], bb[2][4];
      const int trips = kper >> 6;
      if (trips > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a[0][u] = load4<T>(qp + 16 * u);
          bb[0][u] = load4<T>(kp + 16 * u);
        }
      }
      for (int tr = 0; tr < trips; tr += 2) {
        if (tr + 1 < trips) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            a[1][u] = load4<T>(qp + (tr + 1) * 64 + 16 * u);
            bb[1][u] = load4<T>(kp + (tr + 1) * 64 + 16 * u);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][u][s], bb[0][u][s], acc, 0, 0, 0);
        if (tr + 1 < trips) {
          if (tr + 2 < trips) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              a[0][u] = load4<T>(qp + (tr + 2) * 64 + 16 * u);
              bb[0][u] = load4<T>(kp + (tr + 2) * 64 + 16 * u);
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][u][s], bb[1][u][s], acc, 0, 0, 0);
        }
      }
      for (int k = trips << 6; k < kper; k += 16) {
        const f32x4_t a1 = load4<T>(qp + k), b1 = load4<T>(kp + k);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b1[s], acc, 0, 0, 0);
      }
      // D: column (key) = li, rows (queries) = 4*kg + r
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&xs[t * 16 + 4 * kg + r][li], acc[r]);      // LDS add: folds the k-slices
    }
  }
  __syncthreads();

  // column softmax over the queries: TPK threads per key.  The scaled score is ROUNDED (stored back) before the maximum is
  // subtracted: hipcc contracts x * scale - m into fma(x, scale, -m) (also through __fmul_rn), and then the maximum
  // element itself comes out as the rounding error of x*scale -- ~1e-7 of the score, i.e. exp(+1e9) = inf once logits
  // reach 1e16 (an untrained net under guidance does that) -- where softmax(scores * scale) of the reference gives exp(0)
  constexpr int TPK = 64 * AS_WAVES / AS_KB;          // threads per key (a power of two <= 64: one wave or a part of one)
  const int key = tid / TPK, sub = tid % TPK;
  float m = -INFINITY;
  for (int i = sub; i < S; i += TPK) {
    const float v = xs[i][key] * scale;
    xs[i][key] = v;
    m = fmaxf(m, v);
  }
#pragma unroll
  for (int o = TPK / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float l = 0.f;
  for (int i = sub; i < S; i += TPK) l += __expf(xs[i][key] - m);
#pragma unroll
  for (int o = TPK / 2; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
  if (sub == 0) {
    s_m[key] = m;
    s_inv[key] = 1.f / l;
  }
  __syncthreads();
  // P rows: 16 consecutive keys per query
  T* pb = P + (long long)b * S * ldp;
  for (int e = tid; e < S * AS_KB; e += 64 * AS_WAVES) {
    const int i = e >> 4, jj = e & 15;
    const int j = j0 + jj;
    if (j < S) store1<T>(pb + (long long)i * ldp + j, __expf(xs[i][jj] - s_m[jj]) * s_inv[jj]);
    else if (j < ldp) store1<T>(pb + (long long)i * ldp + j, 0.f);      // reduction padding of P.V stays zero
  }
}

// res[q][dd] = sum_j P[q][j] v[j][dd]; workgroup = 32 queries x 64 channels, wave = 16 queries x 32 channels
template <typename T>
__global__ __launch_bounds__(256) void attn_pv_small_kernel(const T* __restrict__ P, long long ldp, const T* __restrict__ qkv,
                                                            long long ld, int S, int heads, int dk, T* __restrict__ res,
                                                            long long ldr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, img = b / heads, head = b % heads;
  const int ndt = (dk + 63) >> 6;
  const int qt = blockIdx.x / ndt, dt = blockIdx.x % ndt;
  const int q0 = qt * 32 + (wave >> 1) * 16, d0 = dt * 64 + (wave & 1) * 32;
  const int li = lane & 15, kg = lane >> 4;
  if (q0 >= S || d0 >= dk) return;
  const T* pb = P + (long long)b * S * ldp + (long long)min(q0 + li, S - 1) * ldp;
  const T* vb = qkv + (long long)img * S * ld + (long long)head * 3 * dk + 2 * dk;
  const int c0 = min(d0 + li, dk - 1), c1 = min(d0 + 16 + li, dk - 1);
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int Sk = (S + 15) & ~15;
  for (int j = 0; j < Sk; j += 16) {
    const int jj = j + 4 * kg;
    f32x4_t a;
    if (jj + 3 < S) {
      a = load4<T>(pb + jj);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) a[s] = jj + s < S ? load1<T>(pb + jj + s) : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const long long row = (long long)min(jj + s, S - 1) * ld;
      const float v0 = load1<T>(vb + row + c0), v1 = load1<T>(vb + row + c1);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], v0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], v1, acc1, 0, 0, 0);
    }
  }
  T* rb = res + ((long long)img * S) * ldr + (long long)head * dk;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + 4 * kg + r;
    if (q >= S) continue;
    if (d0 + li < dk) store1<T>(rb + (long long)q * ldr + d0 + li, acc0[r]);
    if (d0 + 16 + li < dk) store1<T>(rb + (long long)q * ldr + d0 + 16 + li, acc1[r]);
  }
}

// ---- no-grad form: nothing but the result is kept, so the probabilities never exist in memory ---------------------------
// attn_scores_tile: ONE 16 x 16 score tile per workgroup (a query tile against a key block; the 16 waves split the head
// width), written as RAW scaled scores together with the tile's column statistics {max, sum exp} per key -- 256 workgroups
// at S = 256 instead of 16 that each pulled the whole Q matrix through one CU (34 us -> a few).
// attn_pv_raw: folds the per-tile statistics of every key once per workgroup (S <= 256 keys: 2 KB of LDS) and forms
// exp(s - m_j) / l_j while it loads its P operand.
template <typename T>
__global__ __launch_bounds__(1024) void attn_scores_tile_kernel(const T* __restrict__ qkv, long long ld, int S, int heads, int dk,
                                                                float scale, int ksplit, float* __restrict__ raw, long long ldraw,
                                                                float* __restrict__ pstats) {
  __shared__ float xs[16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, img = b / heads, head = b % heads;
  const int j0 = blockIdx.x * 16, q0 = blockIdx.y * 16, mt = gridDim.y;
  const T* qb = qkv + (long long)img * S * ld + (long long)head * 3 * dk;
  const T* kb = qb + dk;
  const int li = lane & 15, kg = lane >> 4;
  if (tid < 16 * 17) (&xs[0][0])[tid] = 0.f;
  __syncthreads();
  if (wave < ksplit) {
    const int kper = dk / ksplit, kbeg = wave * kper;
    const T* kp = kb + (long long)min(j0 + li, S - 1) * ld + kbeg + 4 * kg;
    const T* qp = qb + (long long)min(q0 + li, S - 1) * ld + kbeg + 4 * kg;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < kper; k += 64) {          // kper is a multiple of 16
      f32x4_t a[4], bb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k + 16 * u < kper) {
          a[u] = load4<T>(qp + k + 16 * u);
          bb[u] = load4<T>(kp + k + 16 * u);
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k + 16 * u < kper) {
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s], bb[u][s], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(&xs[4 * kg + r][li], acc[r]);
  }
  __syncthreads();
  if (tid < 256) {
    const int key = tid >> 4, i = tid & 15;            // 16 lanes per key: the tile's queries
    const int q = q0 + i, j = j0 + key;
    float v = xs[i][key] * scale;
    asm volatile("" : "+v"(v));                        // the rounded product (see attn_scores_small_kernel)
    if (q < S && j < S) raw[((long long)b * S + q) * ldraw + j] = v;
    v = q < S ? v : -INFINITY;
    float m = v;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float l = q < S ? __expf(v - m) : 0.f;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
    if (i == 0 && j < S) {
      float* o2 = pstats + (((long long)b * S + j) * mt + blockIdx.y) * 2;      // a key's tile statistics are contiguous
      o2[0] = m;
      o2[1] = l;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_pv_raw_kernel(const float* __restrict__ raw, long long ldraw, const float* __restrict__ pstats,
                                                          int mt, const T* __restrict__ qkv, long long ld, int S, int heads, int dk,
                                                          T* __restrict__ res, long long ldr) {
  __shared__ __attribute__((aligned(16))) float s_m[AS_MAXS], s_inv[AS_MAXS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, img = b / heads, head = b % heads;
  for (int j = S + tid; j < AS_MAXS; j += 256) { s_m[j] = 0.f; s_inv[j] = 0.f; }      // (masked keys: exp(-inf - 0) * 0)
  for (int j = tid; j < S; j += 256) {               // fold the per-tile statistics of key j (mt <= 16 contiguous pairs)
    const float2* p = (const float2*)(pstats + ((long long)b * S + j) * mt * 2);
    float2 v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = t < mt ? p[t] : make_float2(-INFINITY, 0.f);     // independent loads, one latency
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 16; ++t) mx = fmaxf(mx, v[t].x);
    float sm = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) sm += v[t].x > -INFINITY ? v[t].y * __expf(v[t].x - mx) : 0.f;
    s_m[j] = mx;
    s_inv[j] = 1.f / sm;
  }
  __syncthreads();
  const int ndt = (dk + 63) >> 6;
  const int qt = blockIdx.x / ndt, dt = blockIdx.x % ndt;
  const int q0 = qt * 32 + (wave >> 1) * 16, d0 = dt * 64 + (wave & 1) * 32;
  const int li = lane & 15, kg = lane >> 4;
  if (q0 >= S || d0 >= dk) return;
  const float* pb = raw + ((long long)b * S + min(q0 + li, S - 1)) * ldraw;
  const T* vb = qkv + (long long)img * S * ld + (long long)head * 3 * dk + 2 * dk;
  const int c0 = min(d0 + li, dk - 1), c1 = min(d0 + 16 + li, dk - 1);
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int Sk = (S + 15) & ~15;
  // one 16-key step = one 16-byte load of raw scores (+ the keys' statistics from LDS) and eight scalar V loads feeding eight
  // MFMAs; the loads of step j+16 are issued before the MFMAs of step j (two register sets), so the wave waits for memory
  // once per step pair, not once per step
  auto load_step = [&](int j, f32x4_t& rv, f32x4_t& v0, f32x4_t& v1) {      // issues loads only: nothing waits here
    const int jj = j + 4 * kg;
    if (jj + 3 < S) {
      rv = *(const f32x4_t*)(pb + jj);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) rv[s] = jj + s < S ? pb[jj + s] : -INFINITY;      // exp(-inf - m) = 0
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const long long row = (long long)min(jj + s, S - 1) * ld;
      v0[s] = load1<T>(vb + row + c0);
      v1[s] = load1<T>(vb + row + c1);
    }
  };
  auto mma_step = [&](int j, const f32x4_t& rv, const f32x4_t& v0, const f32x4_t& v1) {
    const int jj = min(j + 4 * kg, AS_MAXS - 4);
    const f32x4_t mv = *(const f32x4_t*)(s_m + jj), iv = *(const f32x4_t*)(s_inv + jj);
    f32x4_t a;
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = __expf(rv[s] - mv[s]) * iv[s];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], v0[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], v1[s], acc1, 0, 0, 0);
    }
  };
  f32x4_t aA, vA0, vA1, aB, vB0, vB1;
  load_step(0, aA, vA0, vA1);
  for (int j = 0; j < Sk; j += 32) {
    if (j + 16 < Sk) load_step(j + 16, aB, vB0, vB1);
    mma_step(j, aA, vA0, vA1);
    if (j + 16 < Sk) {
      if (j + 32 < Sk) load_step(j + 32, aA, vA0, vA1);
      mma_step(j + 16, aB, vB0, vB1);
    }
  }
  T* rb = res + ((long long)img * S) * ldr + (long long)head * dk;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + 4 * kg + r;
    if (q >= S) continue;
    if (d0 + li < dk) store1<T>(rb + (long long)q * ldr + d0 + li, acc0[r]);
    if (d0 + 16 + li < dk) store1<T>(rb + (long long)q * ldr + d0 + 16 + li, acc1[r]);
  }
}

static int attn_small_check(const void* qkv, int64_t ld, int batch, int S, int heads, int dk, const void* P, int64_t ldp,
                            int elt) {
  SDM_CHECK_ARG(qkv && P && batch > 0 && heads > 0 && batch % heads == 0, "attn_small: bad args");
  SDM_CHECK_ARG(S > 0 && S <= AS_MAXS, "attn_small: S = %d (this path holds at most %d queries on chip)", S, AS_MAXS);
  SDM_CHECK_ARG(dk > 0 && dk % 16 == 0, "attn_small: head width %d must be a multiple of 16", dk);
  SDM_CHECK_ARG(ld >= 3LL * heads * dk && ldp >= S, "attn_small: row pitches");
  SDM_CHECK_ARG((ld * elt) % 16 == 0 && (((uintptr_t)qkv) & 15) == 0 && (dk * elt) % 16 == 0 && (ldp * elt) % (4 * elt) == 0 &&
                    (ldp % 4) == 0 && (((uintptr_t)P) & 15) == 0,
                "attn_small: rows must be 16-byte aligned (pitch of qkv, head width, pitch of P a multiple of 4)");
  return SDM_OK;
}

extern "C" int sdm_attn_scores_small(const void* qkv, int64_t ld, int batch, int S, int heads, int dk, float scale, void* P,
                                     int64_t ldp, int bf16, sdm_stream_t s) {
  int rc = attn_small_check(qkv, ld, batch, S, heads, dk, P, ldp, bf16 ? 2 : 4);
  if (rc != SDM_OK) return rc;
  dim3 grid(cdiv(bf16 ? (ldp < ((S + 15) & ~15) ? ldp : ((S + 15) & ~15)) : S, AS_KB), batch);
  if (bf16)
    SDM_LAUNCH(attn_scores_small_kernel<bf16_t>, grid, dim3(64 * AS_WAVES), 0, (hipStream_t)s, (const bf16_t*)qkv,
                       (long long)ld, S, heads, dk, scale, (bf16_t*)P, (long long)ldp);
  else
    SDM_LAUNCH(attn_scores_small_kernel<float>, grid, dim3(64 * AS_WAVES), 0, (hipStream_t)s, (const float*)qkv,
                       (long long)ld, S, heads, dk, scale, (float*)P, (long long)ldp);
  SDM_CHECK_LAUNCH("attn_scores_small");
  return SDM_OK;
}

extern "C" int sdm_attn_pv_small(const void* P, int64_t ldp, const void* qkv, int64_t ld, int batch, int S, int heads, int dk,
                                 void* res, int64_t ldr, int bf16, sdm_stream_t s) {
  int rc = attn_small_check(qkv, ld, batch, S, heads, dk, P, ldp, bf16 ? 2 : 4);
  if (rc != SDM_OK) return rc;
  SDM_CHECK_ARG(res && ldr >= (int64_t)heads * dk, "attn_pv_small: bad output");
  dim3 grid(cdiv(S, 32) * cdiv(dk, 64), batch);
  if (bf16)
    SDM_LAUNCH(attn_pv_small_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)P, (long long)ldp,
                       (const bf16_t*)qkv, (long long)ld, S, heads, dk, (bf16_t*)res, (long long)ldr);
  else
    SDM_LAUNCH(attn_pv_small_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, (const float*)P, (long long)ldp,
                       (const float*)qkv, (long long)ld, S, heads, dk, (float*)res, (long long)ldr);
  SDM_CHECK_LAUNCH("attn_pv_small");
  return SDM_OK;
}

extern "C" size_t sdm_attn_small_nograd_workspace_bytes(int batch, int S) {
  const size_t sp = (size_t)((S + 3) & ~3), mt = (size_t)((S + 15) >> 4);
  return ((size_t)batch * S * sp + (size_t)batch * mt * S * 2) * sizeof(float);
}

extern "C" int sdm_attn_small_nograd(const void* qkv, int64_t ld, int batch, int S, int heads, int dk, float scale, void* res,
                                     int64_t ldr, int bf16, float* ws, size_t ws_bytes, sdm_stream_t s) {
  SDM_CHECK_ARG(qkv && res && ws && batch > 0 && heads > 0 && batch % heads == 0, "attn_small_nograd: bad args");
  SDM_CHECK_ARG(S > 0 && S <= AS_MAXS && dk > 0 && dk % 16 == 0, "attn_small_nograd: S <= %d, head width a multiple of 16", AS_MAXS);
  const int elt = bf16 ? 2 : 4;
  SDM_CHECK_ARG(ld >= 3LL * heads * dk && (ld * elt) % 16 == 0 && (((uintptr_t)qkv) & 15) == 0 && (dk * elt) % 16 == 0 &&
                    ldr >= (int64_t)heads * dk && aligned16(ws) && ws_bytes >= sdm_attn_small_nograd_workspace_bytes(batch, S),
                "attn_small_nograd: 16-byte aligned rows and a workspace of sdm_attn_small_nograd_workspace_bytes");
  const int sp = (S + 3) & ~3, mt = (S + 15) >> 4;
  float* raw = ws;
  float* pstats = ws + (size_t)batch * S * sp;
  int ksplit = 16;
  while (ksplit > 1 && dk % (16 * ksplit) != 0) ksplit >>= 1;
  dim3 g1(cdiv(S, 16), mt, batch), g2(cdiv(S, 32) * cdiv(dk, 64), batch);
  hipStream_t st = (hipStream_t)s;
  if (bf16) {
    SDM_LAUNCH(attn_scores_tile_kernel<bf16_t>, g1, dim3(1024), 0, st, (const bf16_t*)qkv, (long long)ld, S, heads, dk, scale,
                       ksplit, raw, (long long)sp, pstats);
    SDM_CHECK_LAUNCH("attn_scores_tile");
    SDM_LAUNCH(attn_pv_raw_kernel<bf16_t>, g2, dim3(256), 0, st, (const float*)raw, (long long)sp, (const float*)pstats, mt,
                       (const bf16_t*)qkv, (long long)ld, S, heads, dk, (bf16_t*)res, (long long)ldr);
  } else {
    SDM_LAUNCH(attn_scores_tile_kernel<float>, g1, dim3(1024), 0, st, (const float*)qkv, (long long)ld, S, heads, dk, scale,
                       ksplit, raw, (long long)sp, pstats);
    SDM_CHECK_LAUNCH("attn_scores_tile");
    SDM_LAUNCH(attn_pv_raw_kernel<float>, g2, dim3(256), 0, st, (const float*)raw, (long long)sp, (const float*)pstats, mt,
                       (const float*)qkv, (long long)ld, S, heads, dk, (float*)res, (long long)ldr);
  }
  SDM_CHECK_LAUNCH("attn_pv_raw");
  return SDM_OK;
}
