// AdaGN backward, coefficient pass folded into the apply pass (two launches instead of three per GroupNorm backward:
// models/custom_layers.py:26-45,241-245 of the reference is one autograd node; here it was sums -> coefficients -> apply).
// Included by sdm_pointwise.hip (fp32 rows, BF = false) and sdm_bf16_train.hip (bf16 rows, BF = true); each translation unit
// instantiates only its own row type (sdm_dsilu differs between them: SDM_FAST_SIGMOID).
//
// Grid (row chunks, images).  Every workgroup of image n rebuilds the image's three per-channel coefficients from the
// channel sums in `ws` (2 C floats, L2-resident; the group reductions walk LDS in channel order, so every workgroup -- and
// every run -- gets the same bits), keeps its lanes' coefficients in registers and streams its rows.  Workgroup 0 of the
// image also emits dmod / dgamma / dbeta.  The sums must be zero again when the call has run (zero-filled-scratch contract):
// each workgroup counts itself in once its reads of the sums have returned (they were consumed into LDS in front of a
// barrier), and the last one to arrive clears them.  The counter lives in the first `images` words of the `coef` scratch
// (no contract on entry): the sums kernel -- the previous launch on the stream -- zeroes it.
#pragma once
#include <type_traits>

#include "sdm_common.h"

template <bool BF>
struct GnbRow;
template <>
struct GnbRow<false> {
  typedef float elem;
  static constexpr int V = 4, MAXC = 1024;
};
template <>
struct GnbRow<true> {
  typedef bf16_t elem;
  static constexpr int V = 8, MAXC = 2048;
};

template <bool BF>
__global__ __launch_bounds__(256) void gn_bwd_apply_fused_kernel(
    const typename GnbRow<BF>::elem* __restrict__ dout, long long lddo, const typename GnbRow<BF>::elem* __restrict__ y,
    long long ldy, const typename GnbRow<BF>::elem* __restrict__ z, long long ldz, float* __restrict__ ws, int* __restrict__ ctr,
    int HW, int C, int groups, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mod,
    int mod_ns, const float* __restrict__ mean, const float* __restrict__ rstd, typename GnbRow<BF>::elem* __restrict__ dz,
    long long lddz, float* __restrict__ dmod, float* __restrict__ dgamma, float* __restrict__ dbeta, int rows_per_block) {
  typedef GnbRow<BF> R;
  constexpr int V = R::V;
  __shared__ float k0s[R::MAXC], k1s[R::MAXC], k2s[R::MAXC];
  __shared__ float gA[R::MAXC / 2], gB[R::MAXC / 2], gMu[R::MAXC / 2], gRs[R::MAXC / 2];      // groups <= MAXC / 2 (host check)
  __shared__ int s_last;
  const int tid = threadIdx.x, n = blockIdx.y, cpg = C / groups;
  const float* sums = ws + (long long)n * C * 2;
  // lanes: q = C / V per row, 256 / q rows in flight; a lane's V channels and their coefficients never change
  typedef typename std::conditional<BF, u16x8, f32x4>::type vec_t;
  const int q = C / V, rl = tid / q, cl = tid - rl * q, RL = 256 / q;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > HW) r1 = HW;
  int r = r0 + (rl < RL ? rl : HW);
  vec_t dv = {}, yv = {}, zv = {};
  if (r < r1) {        // the first row's operands travel while the coefficients are rebuilt
    const long long row = (long long)n * HW + r;
    dv = *(const vec_t*)(dout + row * lddo + cl * V);
    yv = *(const vec_t*)(y + row * ldy + cl * V);
    if (z) zv = *(const vec_t*)(z + row * ldz + cl * V);
  }

  for (int c = tid; c < C; c += 256) {
    const float S1 = sums[2 * c], S2 = sums[2 * c + 1];
    const float sm = mod ? mod[(long long)n * mod_ns + c] : 1.f;
    const float ga = gamma[c];
    const int g = c / cpg;
    const float mu = mean[n * groups + g], rs = rstd[n * groups + g];
    const float xh = rs * (S2 - mu * S1);   // sum dout * xhat
    k0s[c] = rs * sm * ga;
    k1s[c] = sm * ga * S1;
    k2s[c] = sm * ga * xh;
    if (c - g * cpg == 0) { gMu[g] = mu; gRs[g] = rs; }
    if (blockIdx.x == 0) {
      const float be = beta[c];
      if (dmod) dmod[(long long)n * C + c] = ga * xh + (mod ? (be + 1.f) : be) * S1;
      if (dgamma) atomicAdd(&dgamma[c], sm * xh);
      if (dbeta) atomicAdd(&dbeta[c], sm * S1);
    }
  }
  __syncthreads();                     // every read of the sums by this workgroup has returned
  const float cnt = (float)HW * (float)cpg;
  for (int g = tid; g < groups; g += 256) {
    float A = 0.f, B = 0.f;
    for (int i = 0; i < cpg; ++i) {
      A += k1s[g * cpg + i];
      B += k2s[g * cpg + i];
    }
    const float mu = gMu[g], rs = gRs[g];
    A /= cnt;
    B /= cnt;
    gA[g] = -rs * rs * B;
    gB[g] = -rs * A + mu * rs * rs * B;
  }
  __syncthreads();

  float k0[V], k1[V], k2[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = (rl < RL ? cl : 0) * V + i, g = c / cpg;
    k0[i] = k0s[c];
    k1[i] = gA[g];
    k2[i] = gB[g];
  }
  while (r < r1) {
    const long long row = (long long)n * HW + r;
    const int rn = r + RL;
    vec_t dn = {}, yn = {}, zn = {};
    if (rn < r1) {     // next row's operands in flight under this row's arithmetic and store
      const long long rown = (long long)n * HW + rn;
      dn = *(const vec_t*)(dout + rown * lddo + cl * V);
      yn = *(const vec_t*)(y + rown * ldy + cl * V);
      if (z) zn = *(const vec_t*)(z + rown * ldz + cl * V);
    }
    float v[V];
    if constexpr (BF) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = k0[i] * bf2f(dv[i]) + k1[i] * bf2f(yv[i]) + k2[i];
      if (z) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= sdm_dsilu(bf2f(zv[i]));
      }
      u16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
      *(u16x8*)(dz + row * lddz + cl * 8) = o;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = k0[i] * dv[i] + k1[i] * yv[i] + k2[i];
      if (z) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= sdm_dsilu(zv[i]);
      }
      *(f32x4*)(dz + row * lddz + cl * 4) = f32x4{v[0], v[1], v[2], v[3]};
    }
    dv = dn; yv = yn; zv = zn;
    r = rn;
  }
  // arrival: this workgroup's reads of the sums returned in front of the first barrier; the returning atomic is paid once, at
  // the end, by one lane -- the last workgroup of the image to arrive clears what the call consumed
  if (tid == 0) s_last = atomicAdd(&ctr[n], 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (s_last) {
    float* sw = ws + (long long)n * C * 2;
    for (int i = tid; i < 2 * C; i += 256) sw[i] = 0.f;
  }
}

// the fused form applies: vector rows (checked by the caller), C within the LDS tables, at most 256 lanes per row
static inline bool gn_bwd_fused_ok(int C, int groups, int V, int maxc) {
  static const char* sw = getenv("SDM_GN_BWD_FUSED");      // "0": the three-launch form (A/B, and the reference for the test)
  if (sw && sw[0] == '0') return false;
  return C % V == 0 && C <= maxc && C / V <= 256 && groups <= maxc / 2;
}
