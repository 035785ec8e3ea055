// Bandwidth-bound kernels of the U-Net step: NCHW<->NHWC edge convs, GroupNorm/AdaGN folding,
// column softmax, embedding MLP GEMVs, sampler updates, loss, fused Adam.  All wave64.
#include "sdm_common.h"

// --------------------------------------------------------------------------------------------
// Edge convolutions (3 or 6 image channels on one side).  The small channel count is a template parameter so that
// every per-thread array is indexed by compile-time constants and stays in registers (a runtime-indexed array
// would live in scratch memory).
// --------------------------------------------------------------------------------------------
#define STEM_PIX 32
#define EDGE_MAXC 8

// NHWC-side element type of the edge kernels: float (fp32 path) or bf16_t (bf16 storage / training path)
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { *p = f2bf(v); }

#define EDGE_DISPATCH(C, CALL)                                   \
  switch (C) {                                                   \
    case 1: { constexpr int EC = 1; CALL; } break;               \
    case 2: { constexpr int EC = 2; CALL; } break;               \
    case 3: { constexpr int EC = 3; CALL; } break;               \
    case 4: { constexpr int EC = 4; CALL; } break;               \
    case 5: { constexpr int EC = 5; CALL; } break;               \
    case 6: { constexpr int EC = 6; CALL; } break;               \
    case 7: { constexpr int EC = 7; CALL; } break;               \
    default: { constexpr int EC = 8; CALL; } break;              \
  }

// input patches of STEM_PIX consecutive pixels -> LDS  [pixel][ci*9 + tap]
template <int CIN>
__device__ __forceinline__ void stem_fill_patches(float (*patch)[EDGE_MAXC * 9], const float* __restrict__ x, long long p0,
                                                  long long P, int H, int W) {
  constexpr int K = CIN * 9;
  for (int e = threadIdx.x; e < STEM_PIX * K; e += 256) {
    const int pi = e / K, k = e - pi * K;
    const int ci = k / 9, tap = k - ci * 9;
    const long long p = p0 + pi;
    float v = 0.f;
    if (p < P) {
      const int n = (int)(p / (H * W)), rem = (int)(p - (long long)n * H * W);
      const int y = rem / W + tap / 3 - 1, xx = rem % W + tap % 3 - 1;
      if ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) v = x[(((long long)n * CIN + ci) * H + y) * W + xx];
    }
    patch[pi][k] = v;
  }
}

// stem: Conv2d(CIN -> C) k3 p1, NCHW image in, NHWC rows out (+SiLU); threads over output channels
template <int CIN>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        float* __restrict__ preact, int images, int H, int W, int Cout,
                                                        int act) {
  constexpr int K = CIN * 9;
  __shared__ float patch[STEM_PIX][EDGE_MAXC * 9];
  const long long P = (long long)images * H * W;
  const long long p0 = (long long)blockIdx.x * STEM_PIX;
  stem_fill_patches<CIN>(patch, x, p0, P, H, W);
  __syncthreads();
  for (int co = threadIdx.x; co < Cout; co += 256) {
    float wr[K];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) wr[ci * 9 + tap] = w[((long long)tap * Cout + co) * CIN + ci];
    const float b = bias ? bias[co] : 0.f;
    for (int pi = 0; pi < STEM_PIX; ++pi) {
      const long long p = p0 + pi;
      if (p >= P) break;
      float a = b;
#pragma unroll
      for (int k = 0; k < K; ++k) a = fmaf(patch[pi][k], wr[k], a);
      if (preact) preact[p * Cout + co] = a;
      out[p * Cout + co] = sdm_act(a, act);
    }
  }
}

extern "C" int sdm_stem_conv(const float* x, const float* w, const float* bias, float* out, float* preact, int images,
                             int Cin, int H, int W, int Cout, int act, sdm_stream_t s) {
  SDM_CHECK_ARG(x && w && out && images > 0 && H > 0 && W > 0 && Cout > 0, "stem_conv: bad args");
  SDM_CHECK_ARG(Cin >= 1 && Cin <= EDGE_MAXC, "stem_conv: Cin %d not in [1,%d]", Cin, EDGE_MAXC);
  const long long P = (long long)images * H * W;
  EDGE_DISPATCH(Cin, SDM_LAUNCH(stem_conv_kernel<EC>, dim3(cdiv(P, STEM_PIX)), dim3(256), 0, (hipStream_t)s, x, w,
                                        bias, out, preact, images, H, W, Cout, act));
  SDM_CHECK_LAUNCH("stem_conv");
  return SDM_OK;
}

// stem weight/bias gradient: dw[tap][co][ci] = sum_p g[p][co] * x[n][ci][p+d(tap)] ; db[co] = sum_p g[p][co]
// each workgroup walks pix_per_block pixels; thread = output channel, CIN*9+1 register accumulators; partials are
// written to a scratch slab [block][K+1][Cout] and folded by stem_wgrad_reduce_kernel (no atomics).
template <int CIN, typename TG>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const TG* __restrict__ g,
                                                         float* __restrict__ slab, int images, int H, int W, int Cout,
                                                         int pix_per_block) {
  constexpr int K = CIN * 9;
  __shared__ float patch[STEM_PIX][EDGE_MAXC * 9];
  const long long P = (long long)images * H * W;
  const long long pb = (long long)blockIdx.x * pix_per_block;
  for (int cb = 0; cb < Cout; cb += 256) {
    const int co = cb + threadIdx.x;
    float acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.f;
    float accb = 0.f;
    for (int q = 0; q < pix_per_block; q += STEM_PIX) {
      __syncthreads();
      stem_fill_patches<CIN>(patch, x, pb + q, P, H, W);
      __syncthreads();
      if (co < Cout) {
        for (int pi = 0; pi < STEM_PIX; ++pi) {
          const long long p = pb + q + pi;
          if (p >= P || q + pi >= pix_per_block) break;
          const float gv = ldf(g + p * Cout + co);
          accb += gv;
#pragma unroll
          for (int k = 0; k < K; ++k) acc[k] = fmaf(gv, patch[pi][k], acc[k]);
        }
      }
    }
    if (co < Cout) {
      float* sl = slab + (long long)blockIdx.x * (K + 1) * Cout;
#pragma unroll
      for (int k = 0; k < K; ++k) sl[(long long)k * Cout + co] = acc[k];
      sl[(long long)K * Cout + co] = accb;
    }
  }
}

// out layout: dw[tap][co][ci] and db[co] from slab [block][ci*9+tap | K][co]
__global__ void stem_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, float* __restrict__ db,
                                         int nblocks, int Cin, int Cout) {
  const int K = Cin * 9;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= (K + 1) * Cout) return;
  const int k = e / Cout, co = e - k * Cout;
  float a = 0.f;
  for (int b = 0; b < nblocks; ++b) a += slab[((long long)b * (K + 1) + k) * Cout + co];
  if (k == K) {
    if (db) db[co] += a;
  } else {
    const int ci = k / 9, tap = k - ci * 9;
    dw[((long long)tap * Cout + co) * Cin + ci] += a;
  }
}

#define EDGE_WGRAD_BLOCKS 256
extern "C" size_t sdm_edge_wgrad_workspace_bytes(int Cin_or_Cout_small, int C) {
  return (size_t)EDGE_WGRAD_BLOCKS * (9 * Cin_or_Cout_small + 1) * C * sizeof(float);
}

template <typename TG>
static int stem_conv_wgrad_impl(const float* x, const TG* g, float* dw, float* db, float* ws, int images, int Cin, int H, int W,
                                int Cout, sdm_stream_t s) {
  SDM_CHECK_ARG(x && g && dw && ws && Cin >= 1 && Cin <= EDGE_MAXC, "stem_conv_wgrad: bad args");
  hipStream_t st = (hipStream_t)s;
  const long long P = (long long)images * H * W;
  int ppb = (int)((P + EDGE_WGRAD_BLOCKS - 1) / EDGE_WGRAD_BLOCKS);
  ppb = ((ppb + STEM_PIX - 1) / STEM_PIX) * STEM_PIX;
  const int nb = cdiv(P, ppb);
  EDGE_DISPATCH(Cin, SDM_LAUNCH((stem_wgrad_kernel<EC, TG>), dim3(nb), dim3(256), 0, st, x, g, ws, images, H, W, Cout, ppb));
  SDM_CHECK_LAUNCH("stem_conv_wgrad");
  SDM_LAUNCH(stem_wgrad_reduce_kernel, dim3(cdiv((long)(9 * Cin + 1) * Cout, 256)), dim3(256), 0, st, ws, dw, db, nb,
                     Cin, Cout);
  SDM_CHECK_LAUNCH("stem_wgrad_reduce");
  return SDM_OK;
}

extern "C" int sdm_stem_conv_wgrad(const float* x, const float* g, float* dw, float* db, float* ws, int images, int Cin,
                                   int H, int W, int Cout, sdm_stream_t s) {
  return stem_conv_wgrad_impl<float>(x, g, dw, db, ws, images, Cin, H, W, Cout, s);
}
extern "C" int sdm_stem_conv_wgrad_bf16(const float* x, const uint16_t* g, float* dw, float* db, float* ws, int images, int Cin,
                                        int H, int W, int Cout, sdm_stream_t s) {
  return stem_conv_wgrad_impl<bf16_t>(x, g, dw, db, ws, images, Cin, H, W, Cout, s);
}

// --------------------------------------------------------------------------------------------
// head: Conv2d(C -> COUT<=8) k3 p1, NHWC rows in, NCHW out (+Tanh).  One wave per output pixel.
// --------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void head_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int images, int Cin, int H, int W, int act) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long P = (long long)images * H * W;
  for (long long p = (long long)blockIdx.x * 4 + wave; p < P; p += (long long)gridDim.x * 4) {
    const int n = (int)(p / (H * W)), rem = (int)(p - (long long)n * H * W);
    const int y = rem / W, xx = rem % W;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
      const int sy = y + tap / 3 - 1, sx = xx + tap % 3 - 1;
      if ((unsigned)sy >= (unsigned)H || (unsigned)sx >= (unsigned)W) continue;
      const float* xr = x + (((long long)n * H + sy) * W + sx) * Cin;
      for (int c = lane; c < Cin; c += 64) {
        const float xv = xr[c];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xv, w[((long long)tap * COUT + co) * Cin + c], acc[co]);
      }
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const float v = wave_sum(acc[co]);
      if (lane == co) out[(((long long)n * COUT + co) * H + y) * W + xx] = sdm_act(v + (bias ? bias[co] : 0.f), act);
    }
  }
}

extern "C" int sdm_head_conv(const float* x, const float* w, const float* bias, float* out, int images, int Cin, int H,
                             int W, int Cout, int act, sdm_stream_t s) {
  SDM_CHECK_ARG(x && w && out && images > 0 && H > 0 && W > 0 && Cin > 0, "head_conv: bad args");
  SDM_CHECK_ARG(Cout >= 1 && Cout <= EDGE_MAXC, "head_conv: Cout %d not in [1,%d]", Cout, EDGE_MAXC);
  const long long P = (long long)images * H * W;
  int grid = (int)((P + 3) / 4);
  if (grid > 8192) grid = 8192;
  EDGE_DISPATCH(Cout, SDM_LAUNCH(head_conv_kernel<EC>, dim3(grid), dim3(256), 0, (hipStream_t)s, x, w, bias, out,
                                         images, Cin, H, W, act));
  SDM_CHECK_LAUNCH("head_conv");
  return SDM_OK;
}

// dx[p][c] = silu'(z[p][c]) * sum_tap sum_co dy[n][co][p - d(tap)] * w[tap][co][c]
template <int COUT, typename T>
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                         T* __restrict__ dx, const T* __restrict__ z, int images,
                                                         int Cin, int H, int W) {
  __shared__ float g[4][9 * EDGE_MAXC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long P = (long long)images * H * W;
  for (long long pb = (long long)blockIdx.x * 4; pb < P; pb += (long long)gridDim.x * 4) {
    const long long p = pb + wave;
    __syncthreads();
    if (p < P) {
      const int n = (int)(p / (H * W)), rem = (int)(p - (long long)n * H * W);
      const int y = rem / W, xx = rem % W;
      for (int e = lane; e < 9 * COUT; e += 64) {
        const int tap = e / COUT, co = e - tap * COUT;
        const int sy = y - (tap / 3 - 1), sx = xx - (tap % 3 - 1);
        float v = 0.f;
        if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) v = dy[(((long long)n * COUT + co) * H + sy) * W + sx];
        g[wave][e] = v;
      }
    }
    __syncthreads();
    if (p < P) {
      for (int c = lane; c < Cin; c += 64) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 9 * COUT; ++e) a = fmaf(g[wave][e], w[(long long)e * Cin + c], a);
        if (z) a *= sdm_dsilu(ldf(z + p * Cin + c));
        stf(dx + p * Cin + c, a);
      }
    }
  }
}

template <typename T>
static int head_conv_dgrad_impl(const float* dy, const float* w, T* dx, const T* z, int images, int Cin, int H, int W, int Cout,
                                sdm_stream_t s) {
  SDM_CHECK_ARG(dy && w && dx && Cout >= 1 && Cout <= EDGE_MAXC, "head_conv_dgrad: bad args");
  const long long P = (long long)images * H * W;
  int grid = (int)((P + 3) / 4);
  if (grid > 8192) grid = 8192;
  EDGE_DISPATCH(Cout, SDM_LAUNCH((head_dgrad_kernel<EC, T>), dim3(grid), dim3(256), 0, (hipStream_t)s, dy, w, dx, z,
                                         images, Cin, H, W));
  SDM_CHECK_LAUNCH("head_conv_dgrad");
  return SDM_OK;
}

extern "C" int sdm_head_conv_dgrad(const float* dy, const float* w, float* dx, const float* z, int images, int Cin, int H,
                                   int W, int Cout, sdm_stream_t s) {
  return head_conv_dgrad_impl<float>(dy, w, dx, z, images, Cin, H, W, Cout, s);
}
// sdm_edge_bf16.hip: the transposed conv on v_mfma_f32_16x16x4_f32 (act 3 = data-gradient mode)
int sdm_launch_stem_mfma_bf16(const float* x, const float* w, const float* bias, bf16_t* out, bf16_t* preact, int images, int Cin,
                              int H, int W, int Cout, int act, hipStream_t st);
extern "C" int sdm_head_conv_dgrad_bf16(const float* dy, const float* w, uint16_t* dx, const uint16_t* z, int images, int Cin,
                                        int H, int W, int Cout, sdm_stream_t s) {
  SDM_CHECK_ARG(dy && w && dx && images > 0 && H > 0 && W > 0, "head_conv_dgrad_bf16: bad args");
  static const char* mf = getenv("SDM_STEM_MFMA");          // "0": the one-wave-per-pixel kernel
  if (!(mf && mf[0] == '0') && Cout >= 1 && Cout <= 8) {
    // (the matrix-core kernel's "input" is the gradient image with Cout planes, its "output" the Cin-channel rows)
    const int rc = sdm_launch_stem_mfma_bf16(dy, w, nullptr, dx, const_cast<uint16_t*>(z), images, Cout, H, W, Cin, 3, (hipStream_t)s);
    if (rc == SDM_OK) {
      SDM_CHECK_LAUNCH("head_conv_dgrad_mfma_bf16");
      return SDM_OK;
    }
  }
  return head_conv_dgrad_impl<bf16_t>(dy, w, dx, z, images, Cin, H, W, Cout, s);
}

// dw[tap][co][c] = sum_p g[n][co][p] * x[p + d(tap)][c]; threads over c, COUT*9 register accumulators each,
// partials to a slab [block][tap*COUT+co | 9*COUT (unused)][Cin]; bias sums by the first COUT threads.
template <int COUT, typename TX>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const TX* __restrict__ x, const float* __restrict__ g,
                                                         float* __restrict__ slab, float* __restrict__ db, int images,
                                                         int Cin, int H, int W, int pix_per_block) {
  const long long P = (long long)images * H * W;
  const long long pb = (long long)blockIdx.x * pix_per_block;
  for (int c = threadIdx.x; c < Cin; c += 256) {
    float acc[9 * COUT];
#pragma unroll
    for (int k = 0; k < 9 * COUT; ++k) acc[k] = 0.f;
    for (int q = 0; q < pix_per_block; ++q) {
      const long long p = pb + q;
      if (p >= P) break;
      const int n = (int)(p / (H * W)), rem = (int)(p - (long long)n * H * W);
      const int y = rem / W, xx = rem % W;
      float gv[COUT];
#pragma unroll
      for (int co = 0; co < COUT; ++co) gv[co] = g[(((long long)n * COUT + co) * H + y) * W + xx];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int sy = y + tap / 3 - 1, sx = xx + tap % 3 - 1;
        if ((unsigned)sy >= (unsigned)H || (unsigned)sx >= (unsigned)W) continue;
        const float xv = ldf(x + (((long long)n * H + sy) * W + sx) * Cin + c);
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[tap * COUT + co] = fmaf(gv[co], xv, acc[tap * COUT + co]);
      }
    }
    float* sl = slab + (long long)blockIdx.x * (9 * COUT) * Cin;
#pragma unroll
    for (int k = 0; k < 9 * COUT; ++k) sl[(long long)k * Cin + c] = acc[k];
  }
  if (db && threadIdx.x < COUT) {
    float a = 0.f;
    for (int q = 0; q < pix_per_block; ++q) {
      const long long p = pb + q;
      if (p >= P) break;
      const int n = (int)(p / (H * W)), rem = (int)(p - (long long)n * H * W);
      a += g[((long long)n * COUT + threadIdx.x) * H * W + rem];
    }
    atomicAdd(&db[threadIdx.x], a);
  }
}

__global__ void head_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nblocks, int Cin,
                                         int Cout) {
  const int e = blockIdx.x * 256 + threadIdx.x;      // e = (tap*Cout+co)*Cin + c == index into dw
  if (e >= 9 * Cout * Cin) return;
  float a = 0.f;
  for (int b = 0; b < nblocks; ++b) a += slab[(long long)b * 9 * Cout * Cin + e];
  dw[e] += a;
}

template <typename TX>
static int head_conv_wgrad_impl(const TX* x, const float* g, float* dw, float* db, float* ws, int images, int Cin, int H, int W,
                                int Cout, sdm_stream_t s) {
  SDM_CHECK_ARG(x && g && dw && ws && Cout >= 1 && Cout <= EDGE_MAXC, "head_conv_wgrad: bad args");
  hipStream_t st = (hipStream_t)s;
  const long long P = (long long)images * H * W;
  int ppb = (int)((P + EDGE_WGRAD_BLOCKS - 1) / EDGE_WGRAD_BLOCKS);
  if (ppb < 1) ppb = 1;
  const int nb = cdiv(P, ppb);
  EDGE_DISPATCH(Cout, SDM_LAUNCH((head_wgrad_kernel<EC, TX>), dim3(nb), dim3(256), 0, st, x, g, ws, db, images, Cin, H, W,
                                         ppb));
  SDM_CHECK_LAUNCH("head_conv_wgrad");
  SDM_LAUNCH(head_wgrad_reduce_kernel, dim3(cdiv((long)9 * Cout * Cin, 256)), dim3(256), 0, st, ws, dw, nb, Cin, Cout);
  SDM_CHECK_LAUNCH("head_wgrad_reduce");
  return SDM_OK;
}

extern "C" int sdm_head_conv_wgrad(const float* x, const float* g, float* dw, float* db, float* ws, int images, int Cin,
                                   int H, int W, int Cout, sdm_stream_t s) {
  return head_conv_wgrad_impl<float>(x, g, dw, db, ws, images, Cin, H, W, Cout, s);
}
extern "C" int sdm_head_conv_wgrad_bf16(const uint16_t* x, const float* g, float* dw, float* db, float* ws, int images, int Cin,
                                        int H, int W, int Cout, sdm_stream_t s) {
  return head_conv_wgrad_impl<bf16_t>((const bf16_t*)x, g, dw, db, ws, images, Cin, H, W, Cout, s);
}

// --------------------------------------------------------------------------------------------
// GroupNorm statistics folded with the AdaGN modulation -> per-(image,channel) affine
// one block per (image, group); fp64 accumulation
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gn_affine_kernel(const float* __restrict__ y, long long ldy, int HW, int C,
                                                         int groups, float eps, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ mod,
                                                         int mod_ns, float* __restrict__ scale, float* __restrict__ shift,
                                                         float* __restrict__ mean_o, float* __restrict__ rstd_o, int vec) {
  __shared__ double red[2][16];
  const int n = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cpg = C / groups;
  const float* base = y + (long long)n * HW * ldy + g * cpg;
  double s1 = 0.0, s2 = 0.0;
  const long long total = (long long)HW * cpg;
  if (vec) {  // cpg % 4 == 0 and 16-byte aligned rows: float4 loads, fp32 partials folded into fp64 every 8 loads
    const int q4 = cpg >> 2;
    const long long tot4 = (long long)HW * q4;
    for (long long e0 = threadIdx.x; e0 < tot4; e0 += 1024 * 8) {
      float p1 = 0.f, p2 = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long e = e0 + (long long)u * 1024;
        if (e < tot4) {
          const long long pix = sdm_idiv(e, q4);
          const int c = (int)(e - pix * q4) * 4;
          const f32x4 v = *(const f32x4*)(base + pix * ldy + c);
          p1 += (v.x + v.y) + (v.z + v.w);
          p2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
      }
      s1 += p1;
      s2 += p2;
    }
  } else {
    for (long long e = threadIdx.x; e < total; e += 1024) {
      const long long pix = sdm_idiv(e, cpg);
      const int c = (int)(e - pix * cpg);
      const double v = base[pix * ldy + c];
      s1 += v;
      s2 += v * v;
    }
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
  __syncthreads();
  s1 = 0.0;
  s2 = 0.0;
#pragma unroll
  for (int w = 0; w < 16; ++w) { s1 += red[0][w]; s2 += red[1][w]; }
  const double mean = s1 / (double)total;
  double var = s2 / (double)total - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float mu = (float)mean;
  if (threadIdx.x == 0) {
    mean_o[n * groups + g] = mu;
    rstd_o[n * groups + g] = rstd;
  }
  for (int c = threadIdx.x; c < cpg; c += 1024) {
    const int ch = g * cpg + c;
    const float ga = gamma[ch], be = beta[ch];
    const float sm = mod ? mod[(long long)n * mod_ns + ch] : 1.f;
    const float a = rstd * ga * sm;
    const float b = mod ? (be - mu * rstd * ga) * sm + sm : (be - mu * rstd * ga);
    scale[(long long)n * C + ch] = a;
    shift[(long long)n * C + ch] = b;
  }
}

extern "C" int sdm_gn_affine(const float* y, int64_t ldy, int images, int HW, int C, int groups, float eps,
                             const float* gamma, const float* beta, const float* mod, int mod_nstride, float* scale,
                             float* shift, float* mean, float* rstd, sdm_stream_t s) {
  SDM_CHECK_ARG(y && gamma && beta && scale && shift && mean && rstd, "gn_affine: null pointer");
  SDM_CHECK_ARG(images > 0 && HW > 0 && groups > 0 && C % groups == 0, "gn_affine: C %d not divisible by groups %d", C,
                groups);
  const int vec = ((C / groups) % 4 == 0) && ldy % 4 == 0 && aligned16(y);
  SDM_LAUNCH(gn_affine_kernel, dim3(images * groups), dim3(1024), 0, (hipStream_t)s, y, (long long)ldy, HW, C,
                     groups, eps, gamma, beta, mod, mod_nstride, scale, shift, mean, rstd, vec);
  SDM_CHECK_LAUNCH("gn_affine");
  return SDM_OK;
}

// out = scale*y + shift + cat(x0,x1)
__global__ __launch_bounds__(256) void affine_residual_kernel(const float* __restrict__ y, long long ldy,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int ans,
                                                              const float* __restrict__ x0, long long ldx0,
                                                              const float* __restrict__ x1, long long ldx1, int csplit,
                                                              float* __restrict__ out, long long ldo, long long rows,
                                                              int HW, int C, int vec) {
  if (vec) {
    const int C4 = C >> 2;
    const long long total = rows * C4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
      const long long r = sdm_idiv(e, C4);
      const int c = (int)(e - r * C4) * 4;
      const int n = (int)sdm_idiv(r, HW);
      f32x4 v = *(const f32x4*)(y + r * ldy + c);
      if (scale) {
        const f32x4 a = *(const f32x4*)(scale + (long long)n * ans + c);
        const f32x4 b = *(const f32x4*)(shift + (long long)n * ans + c);
        v = v * a + b;
      }
      if (x0) v += (c < csplit) ? *(const f32x4*)(x0 + r * ldx0 + c) : *(const f32x4*)(x1 + r * ldx1 + (c - csplit));
      *(f32x4*)(out + r * ldo + c) = v;
    }
  } else {
    const long long total = rows * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
      const long long r = sdm_idiv(e, C);
      const int c = (int)(e - r * C);
      const int n = (int)sdm_idiv(r, HW);
      float v = y[r * ldy + c];
      if (scale) v = v * scale[(long long)n * ans + c] + shift[(long long)n * ans + c];
      if (x0) v += (c < csplit) ? x0[r * ldx0 + c] : x1[r * ldx1 + (c - csplit)];
      out[r * ldo + c] = v;
    }
  }
}

extern "C" int sdm_affine_residual(const float* y, int64_t ldy, const float* scale, const float* shift, int aff_nstride,
                                   const float* x0, int64_t ldx0, const float* x1, int64_t ldx1, int csplit, float* out,
                                   int64_t ldo, int images, int HW, int C, sdm_stream_t s) {
  SDM_CHECK_ARG(y && out && images > 0 && HW > 0 && C > 0, "affine_residual: bad args");
  SDM_CHECK_ARG(x0 == nullptr || x1 != nullptr || csplit >= C, "affine_residual: csplit < C without second source");
  const long long rows = (long long)images * HW;
  const int vec = (C % 4 == 0) && (x0 == nullptr || csplit % 4 == 0) && ldy % 4 == 0 && ldx0 % 4 == 0 &&
                  (x1 == nullptr || ldx1 % 4 == 0) && ldo % 4 == 0 && aff_nstride % 4 == 0 && aligned16(y) &&
                  (x0 == nullptr || aligned16(x0)) && aligned16(out) &&
                  (x1 == nullptr || aligned16(x1)) && (scale == nullptr || (aligned16(scale) && aligned16(shift)));
  long long work = rows * (vec ? C / 4 : C);
  int grid = (int)((work + 255) / 256);
  if (grid > 4096) grid = 4096;
  SDM_LAUNCH(affine_residual_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, y, (long long)ldy, scale, shift,
                     aff_nstride, x0, (long long)ldx0, x1, (long long)ldx1, csplit, out, (long long)ldo, rows, HW, C, vec);
  SDM_CHECK_LAUNCH("affine_residual");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// AdaGN backward (see header).  ws: sums [images][C][2] = (sum dout, sum dout*y) -- zero on entry, zero again on
// return (the coefficient pass clears what it consumed); coef: [images][3][C], no contract.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_bwd_sums_kernel(const float* __restrict__ dout, long long lddo,
                                                          const float* __restrict__ y, long long ldy, int HW, int C,
                                                          int rows_per_block, float* __restrict__ ws) {
  const int n = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > HW) r1 = HW;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s1 = 0.f, s2 = 0.f;
    for (int r = r0; r < r1; ++r) {
      const long long row = (long long)n * HW + r;
      const float d = dout[row * lddo + c];
      s1 += d;
      s2 = fmaf(d, y[row * ldy + c], s2);
    }
    atomicAdd(&ws[((long long)n * C + c) * 2 + 0], s1);
    atomicAdd(&ws[((long long)n * C + c) * 2 + 1], s2);
  }
}

// float4 form (C % 4 == 0, C <= 1024, 16-byte aligned rows): a row of C channels takes C/4 lanes, the remaining
// 256/(C/4) lane groups walk different rows; partials meet in LDS, one atomic pair per channel per workgroup.
__global__ __launch_bounds__(256) void gn_bwd_sums_vec_kernel(const float* __restrict__ dout, long long lddo,
                                                              const float* __restrict__ y, long long ldy, int HW, int C,
                                                              int rows_per_block, float* __restrict__ ws) {
  __shared__ float part[2][256][4];
  const int n = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > HW) r1 = HW;
  const int q = C >> 2;                  // lanes per row
  const int rl = threadIdx.x / q, cl = threadIdx.x - rl * q;
  const int RL = 256 / q;                // rows in flight
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  if (rl < RL) {
    for (int r = r0 + rl; r < r1; r += RL) {
      const long long row = (long long)n * HW + r;
      const f32x4 d = *(const f32x4*)(dout + row * lddo + cl * 4);
      const f32x4 yv = *(const f32x4*)(y + row * ldy + cl * 4);
      s1 += d;
      s2 += d * yv;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    part[0][threadIdx.x][k] = s1[k];
    part[1][threadIdx.x][k] = s2[k];
  }
  __syncthreads();
  // thread t < C finishes channel t: lane group cl = t/4, element t%4, summed over the RL row groups
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c >> 2, k = c & 3;
    float a = 0.f, b = 0.f;
    for (int j = 0; j < RL; ++j) {
      a += part[0][j * q + g][k];
      b += part[1][j * q + g][k];
    }
    atomicAdd(&ws[((long long)n * C + c) * 2 + 0], a);
    atomicAdd(&ws[((long long)n * C + c) * 2 + 1], b);
  }
}

// one block per (image, group): turns the sums into per-channel coefficients (k1,k2,k3) stored planar in
// coef[n][3][C], emits dmod / dgamma / dbeta, and clears the sums it consumed.
__global__ __launch_bounds__(64) void gn_bwd_coef_kernel(float* __restrict__ ws, int HW, int C, int groups,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mod, int mod_ns,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ coef, float* __restrict__ dmod,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int n = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cpg = C / groups;
  const float mu = mean[n * groups + g], rs = rstd[n * groups + g];
  float A = 0.f, B = 0.f;
  for (int c = threadIdx.x; c < cpg; c += 64) {
    const int ch = g * cpg + c;
    const float S1 = ws[((long long)n * C + ch) * 2], S2 = ws[((long long)n * C + ch) * 2 + 1];
    const float sm = mod ? mod[(long long)n * mod_ns + ch] : 1.f;
    const float sg = sm * gamma[ch];
    A += sg * S1;
    B += sg * rs * (S2 - mu * S1);
  }
  A = wave_sum(A);
  B = wave_sum(B);
  const float cnt = (float)HW * (float)cpg;
  A /= cnt;
  B /= cnt;
  for (int c = threadIdx.x; c < cpg; c += 64) {
    const int ch = g * cpg + c;
    float* sums = ws + ((long long)n * C + ch) * 2;
    const float S1 = sums[0], S2 = sums[1];
    sums[0] = 0.f;      // same thread read these two words in the first loop as well: nobody else touches them
    sums[1] = 0.f;
    const float sm = mod ? mod[(long long)n * mod_ns + ch] : 1.f;
    const float ga = gamma[ch], be = beta[ch];
    const float xh = rs * (S2 - mu * S1);  // sum dout * xhat
    float* k = coef + (long long)n * 3 * C + ch;
    k[0] = rs * sm * ga;
    k[C] = -rs * rs * B;
    k[2 * C] = -rs * A + mu * rs * rs * B;
    if (dmod) dmod[(long long)n * C + ch] = ga * xh + (mod ? (be + 1.f) : be) * S1;
    if (dgamma) atomicAdd(&dgamma[ch], sm * xh);
    if (dbeta) atomicAdd(&dbeta[ch], sm * S1);
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ dout, long long lddo,
                                                           const float* __restrict__ y, long long ldy,
                                                           const float* __restrict__ z, long long ldz,
                                                           const float* __restrict__ coef, float* __restrict__ dz,
                                                           long long lddz, long long rows, int HW, int C, int vec) {
  if (vec) {
    const int C4 = C >> 2;
    const long long total = rows * C4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
      const long long r = sdm_idiv(e, C4);
      const int c = (int)(e - r * C4) * 4;
      const int n = (int)sdm_idiv(r, HW);
      const float* k = coef + (long long)n * 3 * C + c;
      const f32x4 k0 = *(const f32x4*)k, k1 = *(const f32x4*)(k + C), k2 = *(const f32x4*)(k + 2 * C);
      f32x4 v = k0 * *(const f32x4*)(dout + r * lddo + c) + k1 * *(const f32x4*)(y + r * ldy + c) + k2;
      if (z) {
        const f32x4 zv = *(const f32x4*)(z + r * ldz + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= sdm_dsilu(zv[i]);
      }
      *(f32x4*)(dz + r * lddz + c) = v;
    }
    return;
  }
  const long long total = rows * C;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = sdm_idiv(e, C);
    const int c = (int)(e - r * C);
    const int n = (int)sdm_idiv(r, HW);
    const float* k = coef + (long long)n * 3 * C + c;
    float v = k[0] * dout[r * lddo + c] + k[C] * y[r * ldy + c] + k[2 * C];
    if (z) v *= sdm_dsilu(z[r * ldz + c]);
    dz[r * lddz + c] = v;
  }
}

// host-side launcher of the coefficient pass, shared with the bf16 rows variant (sdm_bf16_train.hip)
int sdm_launch_gn_bwd_coef(float* ws, int images, int HW, int C, int groups, const float* gamma, const float* beta,
                           const float* mod, int mod_ns, const float* mean, const float* rstd, float* coef, float* dmod,
                           float* dgamma, float* dbeta, hipStream_t st) {
  SDM_LAUNCH(gn_bwd_coef_kernel, dim3(images * groups), dim3(64), 0, st, ws, HW, C, groups, gamma, beta, mod, mod_ns,
                     mean, rstd, coef, dmod, dgamma, dbeta);
  SDM_CHECK_LAUNCH("gn_bwd_coef");
  return SDM_OK;
}

extern "C" int sdm_gn_adagn_bwd(const float* dout, int64_t lddo, const float* y, int64_t ldy, const float* z, int64_t ldz,
                                int images, int HW, int C, int groups, const float* gamma, const float* beta,
                                const float* mod, int mod_nstride, const float* mean, const float* rstd, float* dz,
                                int64_t lddz, float* dmod, float* dgamma, float* dbeta, float* ws, float* coef,
                                sdm_stream_t s) {
  SDM_CHECK_ARG(dout && y && gamma && beta && mean && rstd && dz && ws && coef, "gn_adagn_bwd: null pointer");
  SDM_CHECK_ARG(images > 0 && HW > 0 && groups > 0 && C % groups == 0, "gn_adagn_bwd: bad dims");
  hipStream_t st = (hipStream_t)s;
  const bool vec = C % 4 == 0 && lddo % 4 == 0 && ldy % 4 == 0 && lddz % 4 == 0 && (z == nullptr || ldz % 4 == 0) &&
                   aligned16(dout) && aligned16(y) && aligned16(dz) && (z == nullptr || aligned16(z)) && aligned16(coef);
  int rpb = (HW + 63) / 64;
  if (rpb < 8) rpb = 8;
  if (vec && C <= 1024)
    SDM_LAUNCH(gn_bwd_sums_vec_kernel, dim3(cdiv(HW, rpb), images), dim3(256), 0, st, dout, (long long)lddo, y,
                       (long long)ldy, HW, C, rpb, ws);
  else
    SDM_LAUNCH(gn_bwd_sums_kernel, dim3(cdiv(HW, rpb), images), dim3(256), 0, st, dout, (long long)lddo, y,
                       (long long)ldy, HW, C, rpb, ws);
  SDM_CHECK_LAUNCH("gn_bwd_sums");
  SDM_LAUNCH(gn_bwd_coef_kernel, dim3(images * groups), dim3(64), 0, st, ws, HW, C, groups, gamma, beta, mod,
                     mod_nstride, mean, rstd, coef, dmod, dgamma, dbeta);
  SDM_CHECK_LAUNCH("gn_bwd_coef");
  const long long rows = (long long)images * HW;
  const long long work = rows * (vec ? C / 4 : C);
  int grid = (int)((work + 255) / 256);
  if (grid > 4096) grid = 4096;
  SDM_LAUNCH(gn_bwd_apply_kernel, dim3(grid), dim3(256), 0, st, dout, (long long)lddo, y, (long long)ldy, z,
                     (long long)ldz, coef, dz, (long long)lddz, rows, HW, C, vec ? 1 : 0);
  SDM_CHECK_LAUNCH("gn_bwd_apply");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// column softmax: normalise over rows i for every column j.  Block = 64 columns x 4 row-phases.
// --------------------------------------------------------------------------------------------
#define CS_W 16   // row phases (waves) per workgroup; lane = column.  16 phases keep enough loads in flight (the 4-phase form
                  // streamed the S=4096 score matrices at 1.3 TB/s)
__global__ __launch_bounds__(64 * CS_W) void col_softmax_kernel(float* __restrict__ sc, int S, long long lds) {
  __shared__ float sm[CS_W][64], ss[CS_W][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  float* base = sc + (long long)blockIdx.y * S * lds;
  float m = -INFINITY, l = 0.f;
  if (j < S) {
    for (int i = wave; i < S; i += CS_W) {
      const float v = base[(long long)i * lds + j];
      if (v > m) {
        l = l * __expf(m - v) + 1.f;
        m = v;
      } else {
        l += __expf(v - m);
      }
    }
  }
  sm[wave][lane] = m;
  ss[wave][lane] = l;
  __syncthreads();
  float M = sm[0][lane];
#pragma unroll
  for (int w = 1; w < CS_W; ++w) M = fmaxf(M, sm[w][lane]);
  float Lsum = 0.f;
#pragma unroll
  for (int w = 0; w < CS_W; ++w)
    if (ss[w][lane] > 0.f) Lsum += ss[w][lane] * __expf(sm[w][lane] - M);
  if (j < S) {
    const float inv = 1.f / Lsum;
#pragma unroll 4
    for (int i = wave; i < S; i += CS_W) {
      float* p = base + (long long)i * lds + j;
      *p = __expf(*p - M) * inv;
    }
  }
}

extern "C" int sdm_col_softmax(float* scores, int batch, int S, int64_t lds, sdm_stream_t s) {
  SDM_CHECK_ARG(scores && batch > 0 && S > 0 && lds >= S, "col_softmax: bad args");
  SDM_LAUNCH(col_softmax_kernel, dim3(cdiv(S, 64), batch), dim3(64 * CS_W), 0, (hipStream_t)s, scores, S, (long long)lds);
  SDM_CHECK_LAUNCH("col_softmax");
  return SDM_OK;
}

__global__ __launch_bounds__(64 * CS_W) void col_softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, int S,
                                                                    long long lds, float scale) {
  __shared__ float sd[CS_W][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const long long bo = (long long)blockIdx.y * S * lds;
  float dot = 0.f;
  if (j < S) {
#pragma unroll 4
    for (int i = wave; i < S; i += CS_W) dot = fmaf(p[bo + (long long)i * lds + j], dp[bo + (long long)i * lds + j], dot);
  }
  sd[wave][lane] = dot;
  __syncthreads();
  dot = 0.f;
#pragma unroll
  for (int w = 0; w < CS_W; ++w) dot += sd[w][lane];
  if (j < S)
#pragma unroll 4
    for (int i = wave; i < S; i += CS_W) {
      const long long o = bo + (long long)i * lds + j;
      dp[o] = scale * p[o] * (dp[o] - dot);
    }
}

extern "C" int sdm_col_softmax_bwd(const float* p, float* dp, int batch, int S, int64_t lds, float scale, sdm_stream_t s) {
  SDM_CHECK_ARG(p && dp && batch > 0 && S > 0 && lds >= S, "col_softmax_bwd: bad args");
  SDM_LAUNCH(col_softmax_bwd_kernel, dim3(cdiv(S, 64), batch), dim3(64 * CS_W), 0, (hipStream_t)s, p, dp, S,
                     (long long)lds, scale);
  SDM_CHECK_LAUNCH("col_softmax_bwd");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// embedding: sinusoid features and small-batch linears
// --------------------------------------------------------------------------------------------
__global__ void sinusoid_kernel(const int64_t* __restrict__ t, const float* __restrict__ freq, int n, int D,
                                float* __restrict__ out) {
  const int half = D / 2;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * half) return;
  const int r = e / half, i = e - r * half;
  const float a = (float)t[r] * freq[i];
  out[(long long)r * D + i] = sinf(a);
  out[(long long)r * D + half + i] = cosf(a);
}

// freq table is host-computed exactly as the reference does (torch.exp(arange(half) * -ln(1e4)/(half-1))).
extern "C" int sdm_sinusoid(const int64_t* t, const float* freq, int n, int D, float* out, sdm_stream_t s) {
  SDM_CHECK_ARG(t && freq && out && n > 0 && D >= 2 && D % 2 == 0, "sinusoid: bad args");
  const int total = n * (D / 2);
  SDM_LAUNCH(sinusoid_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)s, t, freq, n, D, out);
  SDM_CHECK_LAUNCH("sinusoid");
  return SDM_OK;
}

#define LS_MT 8
// one wave per output feature; x rows streamed LS_MT at a time
__device__ __forceinline__ void linear_small_body(const float* __restrict__ x, long long ldx, const float* __restrict__ w,
                                                  const float* __restrict__ b, const float* __restrict__ add,
                                                  long long ldadd, float* __restrict__ out, long long ldo,
                                                  float* __restrict__ preact, int M, int K, int O, int act, int o, int lane,
                                                  int mb, int nmb) {
  // batch rows in blocks of LS_MT; block mb of nmb walks blocks mb, mb + nmb, ... (the launchers give every block of eight rows
  // its own wave: one wave per feature walking a batch of 32 took 78 us per embedding layer at the head of the C4 step)
  if (o >= O) return;
  const float* wr = w + (long long)o * K;
  for (int m0 = mb * LS_MT; m0 < M; m0 += nmb * LS_MT) {
    float acc[LS_MT];
#pragma unroll
    for (int i = 0; i < LS_MT; ++i) acc[i] = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float wv = wr[k];
#pragma unroll
      for (int i = 0; i < LS_MT; ++i)
        if (m0 + i < M) acc[i] = fmaf(x[(long long)(m0 + i) * ldx + k], wv, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < LS_MT; ++i) {
      const float v = wave_sum(acc[i]);
      if (lane == 0 && m0 + i < M) {
        float r = v + (b ? b[o] : 0.f);
        if (preact) preact[(long long)(m0 + i) * ldo + o] = r;
        r = sdm_act(r, act);
        if (add) r += add[(long long)(m0 + i) * ldadd + o];   // ldadd 0: one row broadcast over the batch
        out[(long long)(m0 + i) * ldo + o] = r;
      }
    }
  }
}

__global__ __launch_bounds__(256) void linear_small_kernel(const float* x, long long ldx, const float* w, const float* b,
                                                           const float* add, long long ldadd, float* out, long long ldo,
                                                           float* preact, int M, int K, int O, int act) {
  linear_small_body(x, ldx, w, b, add, ldadd, out, ldo, preact, M, K, O, act, blockIdx.x * 4 + (threadIdx.x >> 6),
                    threadIdx.x & 63, blockIdx.y, gridDim.y);
}

extern "C" int sdm_linear_small(const float* x, int64_t ldx, const float* w, const float* b, const float* add, int64_t ldadd,
                                int add_rows, float* out, int64_t ldo, float* out_preact, int M, int K, int O, int act,
                                sdm_stream_t s) {
  SDM_CHECK_ARG(x && w && out && M > 0 && K > 0 && O > 0, "linear_small: bad args");
  SDM_CHECK_ARG(add == nullptr || (add_rows == M && ldadd >= O) || (add_rows == 1 && ldadd == 0),
                "linear_small: addend has %d rows (pitch %lld) for a batch of %d: needs M rows, or 1 row with pitch 0", add_rows,
                (long long)ldadd, M);
  SDM_LAUNCH(linear_small_kernel, dim3(cdiv(O, 4), cdiv(M, LS_MT) < 32 ? cdiv(M, LS_MT) : 32), dim3(256), 0, (hipStream_t)s, x, (long long)ldx, w, b, add,
                     (long long)ldadd, out, (long long)ldo, out_preact, M, K, O, act);
  SDM_CHECK_LAUNCH("linear_small");
  return SDM_OK;
}

__global__ __launch_bounds__(256) void linear_multi_kernel(const float* x, long long ldx, int M, int K,
                                                           const sdm_linear_job* jobs) {
  const sdm_linear_job j = jobs[blockIdx.y];
  linear_small_body(x, ldx, j.w, j.b, nullptr, 0, j.out, (long long)j.O, nullptr, M, K, j.O, SDM_ACT_NONE,
                    blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63, blockIdx.z, gridDim.z);
}

extern "C" int sdm_linear_multi(const float* x, int64_t ldx, int M, int K, const sdm_linear_job* jobs_dev, int njobs,
                                int maxO, sdm_stream_t s) {
  SDM_CHECK_ARG(x && jobs_dev && M > 0 && K > 0 && njobs > 0 && maxO > 0, "linear_multi: bad args");
  SDM_LAUNCH(linear_multi_kernel, dim3(cdiv(maxO, 4), njobs, cdiv(M, LS_MT) < 32 ? cdiv(M, LS_MT) : 32), dim3(256), 0, (hipStream_t)s, x, (long long)ldx, M, K,
                     jobs_dev);
  SDM_CHECK_LAUNCH("linear_multi");
  return SDM_OK;
}

// backward of a small-batch linear y = x W^T + b given dy (already multiplied by act'):
//   dx[m][k] (+)= sum_o dy[m][o] W[o][k] ;  dW[o][k] += sum_m dy[m][o] x[m][k] ; db[o] += sum_m dy[m][o]
#define LSB_OC 64
__global__ __launch_bounds__(256) void linear_small_bwd_dx_kernel(const float* __restrict__ dy, long long lddy,
                                                                  const float* __restrict__ w, float* __restrict__ dx,
                                                                  long long lddx, int M, int K, int O) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int m = blockIdx.y;
  const int o0 = blockIdx.z * LSB_OC;
  if (k >= K) return;
  int o1 = o0 + LSB_OC;
  if (o1 > O) o1 = O;
  float a = 0.f;
  for (int o = o0; o < o1; ++o) a = fmaf(dy[(long long)m * lddy + o], w[(long long)o * K + k], a);
  atomicAdd(&dx[(long long)m * lddx + k], a);
}

// dw[o][k] += sum_m dy[m][o] * x[m][k]; db[o] += sum_m dy[m][o]   (M = batch rows of the embedding: 4 .. 256)
// One output row per workgroup (O x K / 256 workgroups: the launch is a chain of memory round trips, so parallelism is what
// it needs), the batch rows taken EIGHT AT A TIME with all sixteen loads issued before the first use: a round trip per eight
// rows instead of one per row (the rolled loop took 12.6 us at M = 32 at best and 37-47 us inside a C4 / C5 step; a variant with
// 16 output rows per workgroup and the dy rows in LDS was SLOWER, 74 us: too few workgroups -- profiles/r04_c4_bf16_train_kernel_stats.csv).
__global__ __launch_bounds__(256) void linear_small_bwd_dw_kernel(const float* __restrict__ dy, long long lddy,
                                                                  const float* __restrict__ x, long long ldx,
                                                                  float* __restrict__ dw, float* __restrict__ db, int M,
                                                                  int K, int O) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int o = blockIdx.y;
  const int kk = k < K ? k : K - 1;
  float a = 0.f, bsum = 0.f;
  const float old = k < K ? dw[(long long)o * K + k] : 0.f;       // in flight under the reduction
  for (int m0 = 0; m0 < M; m0 += 8) {
    float d[8], xv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + i < M ? m0 + i : M - 1;
      d[i] = dy[(long long)m * lddy + o];
      xv[i] = x[(long long)m * ldx + kk];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float di = m0 + i < M ? d[i] : 0.f;
      bsum += di;
      a = fmaf(di, xv[i], a);
    }
  }
  if (k < K) dw[(long long)o * K + k] = old + a;
  if (db && k == 0) db[o] += bsum;
}

__global__ __launch_bounds__(256) void linear_multi_bwd_dx_kernel(const sdm_linear_job* __restrict__ jobs,
                                                                  const int* __restrict__ chunk_job,
                                                                  const int* __restrict__ chunk_o0, float* __restrict__ dx,
                                                                  long long lddx, int K) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int m = blockIdx.y;
  const sdm_linear_job j = jobs[chunk_job[blockIdx.z]];
  const int o0 = chunk_o0[blockIdx.z];
  if (k >= K) return;
  int o1 = o0 + LSB_OC;
  if (o1 > j.O) o1 = j.O;
  const float* dy = j.out + (long long)m * j.O;
  float a = 0.f;
  for (int o = o0; o < o1; ++o) a = fmaf(dy[o], j.w[(long long)o * K + k], a);
  atomicAdd(&dx[(long long)m * lddx + k], a);
}

extern "C" int sdm_linear_multi_bwd_dx(const sdm_linear_job* jobs_dev, const int* chunk_job, const int* chunk_o0, int nchunks,
                                       float* dx, int64_t lddx, int M, int K, sdm_stream_t s) {
  SDM_CHECK_ARG(jobs_dev && chunk_job && chunk_o0 && dx && nchunks > 0 && M > 0 && K > 0, "linear_multi_bwd_dx: bad args");
  SDM_LAUNCH(linear_multi_bwd_dx_kernel, dim3(cdiv(K, 256), M, nchunks), dim3(256), 0, (hipStream_t)s, jobs_dev,
                     chunk_job, chunk_o0, dx, (long long)lddx, K);
  SDM_CHECK_LAUNCH("linear_multi_bwd_dx");
  return SDM_OK;
}

extern "C" int sdm_linear_small_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* w, float* dx,
                                    int64_t lddx, int dx_accumulate, float* dw, float* db, int M, int K, int O,
                                    sdm_stream_t s) {
  SDM_CHECK_ARG(dy && w && M > 0 && K > 0 && O > 0, "linear_small_bwd: bad args");
  hipStream_t st = (hipStream_t)s;
  if (dx) {
    (void)dx_accumulate;  // always accumulates (see header)
    SDM_LAUNCH(linear_small_bwd_dx_kernel, dim3(cdiv(K, 256), M, cdiv(O, LSB_OC)), dim3(256), 0, st, dy,
                       (long long)lddy, w, dx, (long long)lddx, M, K, O);
    SDM_CHECK_LAUNCH("linear_small_bwd_dx");
  }
  if (dw) {
    SDM_CHECK_ARG(x != nullptr, "linear_small_bwd: dw needs x");
    SDM_LAUNCH(linear_small_bwd_dw_kernel, dim3(cdiv(K, 256), O), dim3(256), 0, st, dy, (long long)lddy, x,
                       (long long)ldx, dw, db, M, K, O);
    SDM_CHECK_LAUNCH("linear_small_bwd_dw");
  }
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// generic elementwise / reductions
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long long ldx, long long rows, int C,
                                                     int rows_per_block, float* __restrict__ out) {
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  const int C4 = C >> 2;
  if ((C & 3) == 0 && C4 <= 256 && (ldx & 3) == 0 && (((unsigned long long)x) & 15) == 0) {
    // 16 bytes per lane, the spare lanes of a short row walk further rows, four independent rows per trip; partials meet in LDS
    // (the scalar form below ran the 16384 x 128 gradient of the headline's up-sampling convs at 0.3 TB/s)
    __shared__ float red[256][5];
    int cl = 1;
    while (cl < C4) cl <<= 1;
    const int RL = 256 / cl;
    const int ch = threadIdx.x % cl, rl = threadIdx.x / cl;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (ch < C4) {
      long long r = r0 + rl;
      for (; r + 3 * RL < r1; r += 4 * RL) {
        const f32x4 v0 = *(const f32x4*)(x + r * ldx + 4 * ch), v1 = *(const f32x4*)(x + (r + RL) * ldx + 4 * ch);
        const f32x4 v2 = *(const f32x4*)(x + (r + 2 * RL) * ldx + 4 * ch), v3 = *(const f32x4*)(x + (r + 3 * RL) * ldx + 4 * ch);
        a += (v0 + v1) + (v2 + v3);
      }
      for (; r < r1; r += RL) a += *(const f32x4*)(x + r * ldx + 4 * ch);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[threadIdx.x][k] = a[k];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const int chunk = c >> 2, k = c & 3;
      float t = 0.f;
      for (int q = 0; q < RL; ++q) t += red[q * cl + chunk][k];
      atomicAdd(&out[c], t);
    }
    return;
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (long long r = r0; r < r1; ++r) a += x[r * ldx + c];
    atomicAdd(&out[c], a);
  }
}

extern "C" int sdm_colsum(const float* x, int64_t ldx, int rows, int C, float* out, int accumulate, sdm_stream_t s) {
  SDM_CHECK_ARG(x && out && rows > 0 && C > 0, "colsum: bad args");
  hipStream_t st = (hipStream_t)s;
  if (!accumulate) SDM_MEMSET0(out, (size_t)C * sizeof(float), st);
  int rpb = (rows + 1023) / 1024;
  if (rpb < 32) rpb = 32;
  SDM_LAUNCH(colsum_kernel, dim3(cdiv(rows, rpb)), dim3(256), 0, st, x, (long long)ldx, (long long)rows, C, rpb, out);
  SDM_CHECK_LAUNCH("colsum");
  return SDM_OK;
}

__global__ void silu_bwd_kernel(const float* dy, const float* z, float* dx, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    dx[i] = z ? dy[i] * sdm_dsilu(z[i]) : dy[i];
}
__global__ void add_kernel(const float* a, const float* b, float* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = a[i] + b[i];
}
__global__ void tanh_bwd_kernel(const float* dy, const float* y, float* dx, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    dx[i] = dy[i] * (1.f - y[i] * y[i]);
}
static inline int ew_grid(size_t n) {
  size_t g = (n + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}
__global__ void silu_kernel(const float* x, float* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = sdm_silu(x[i]);
}
extern "C" int sdm_silu(const float* x, float* y, size_t n, sdm_stream_t s) {
  SDM_CHECK_ARG(x && y && n > 0, "silu: bad args");
  SDM_LAUNCH(silu_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)s, x, y, n);
  SDM_CHECK_LAUNCH("silu");
  return SDM_OK;
}
extern "C" int sdm_silu_bwd(const float* dy, const float* z, float* dx, size_t n, sdm_stream_t s) {
  SDM_CHECK_ARG(dy && dx && n > 0, "silu_bwd: bad args");
  SDM_LAUNCH(silu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)s, dy, z, dx, n);
  SDM_CHECK_LAUNCH("silu_bwd");
  return SDM_OK;
}
extern "C" int sdm_add(const float* a, const float* b, float* y, size_t n, sdm_stream_t s) {
  SDM_CHECK_ARG(a && b && y && n > 0, "add: bad args");
  SDM_LAUNCH(add_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)s, a, b, y, n);
  SDM_CHECK_LAUNCH("add");
  return SDM_OK;
}
extern "C" int sdm_tanh_bwd(const float* dy, const float* y, float* dx, size_t n, sdm_stream_t s) {
  SDM_CHECK_ARG(dy && y && dx && n > 0, "tanh_bwd: bad args");
  SDM_LAUNCH(tanh_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)s, dy, y, dx, n);
  SDM_CHECK_LAUNCH("tanh_bwd");
  return SDM_OK;
}

__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ loss, float* __restrict__ grad, size_t n) {
  __shared__ double red[4];
  double acc = 0.0;
  const float gs = 2.0f / (float)n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float d = a[i] - b[i];
    acc += (double)d * d;
    if (grad) grad[i] = d * gs;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (float)((red[0] + red[1] + red[2] + red[3]) / (double)n));
}
extern "C" int sdm_mse(const float* a, const float* b, float* loss_out, float* grad, size_t n, sdm_stream_t s) {
  SDM_CHECK_ARG(a && b && loss_out && n > 0, "mse: bad args");
  hipStream_t st = (hipStream_t)s;
  SDM_MEMSET0(loss_out, sizeof(float), st);
  int g = ew_grid(n);
  if (g > 512) g = 512;
  SDM_LAUNCH(mse_kernel, dim3(g), dim3(256), 0, st, a, b, loss_out, grad, n);
  SDM_CHECK_LAUNCH("mse");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// diffusion glue
// --------------------------------------------------------------------------------------------
__global__ void q_sample_kernel(const float* x0, const float* eps, const float* ab, float* xt, size_t per, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float a = ab[i / per];
    xt[i] = sqrtf(a) * x0[i] + sqrtf(1.f - a) * eps[i];
  }
}
extern "C" int sdm_q_sample(const float* x0, const float* eps, const float* alpha_bar, float* xt, int images,
                            size_t per_image, sdm_stream_t s) {
  SDM_CHECK_ARG(x0 && eps && alpha_bar && xt && images > 0 && per_image > 0, "q_sample: bad args");
  const size_t n = (size_t)images * per_image;
  SDM_LAUNCH(q_sample_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)s, x0, eps, alpha_bar, xt, per_image, n);
  SDM_CHECK_LAUNCH("q_sample");
  return SDM_OK;
}

// ---- sampler step tail: the update kernels also ADVANCE the step cursor (one launch fewer per step than a separate
// advance kernel).  Every other read of *cursor in a step (sdm_step_begin, injected-noise lookups) is in a launch that
// precedes the update on the stream; inside the update kernel ONLY block 0 / thread 0 touches the cursor (read, then
// store), so no block can observe the new value -- anything the other blocks need per step (the Philox counter) comes
// from a word that sdm_step_begin wrote and nobody rewrites during the step (`rng_step` = the timestep).
__device__ __forceinline__ void cursor_advance(int* cursor, int nsteps) {
  if (cursor && blockIdx.x == 0 && threadIdx.x == 0) {
    const int raw = *cursor;
    *cursor = raw < nsteps ? raw + 1 : nsteps;
  }
}

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator behind
// torch.randn on GPUs): counter = {element block lo, hi, step, 0}, key = {seed lo, seed hi}.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// two uniform 32-bit words -> two standard normals (Box-Muller; u1 in (0, 1], u2 in [0, 1))
__device__ __forceinline__ void box_muller(unsigned a, unsigned b, float& z0, float& z1) {
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.28318530717958647692f * u2, &sn, &cs);
  z0 = r * cs;
  z1 = r * sn;
}

__global__ void ddpm_update_kernel(float* x, const float* eps, const float* z, const float* coef, size_t n,
                                   unsigned long long seed, const long long* rng_step, int* cursor, int nsteps) {
  const float s1 = coef[0], s2 = coef[1], sg = coef[2];
  if (!z && rng_step && sg != 0.f) {
    const unsigned step = (unsigned)*rng_step;
    // z ~ N(0, I) in-kernel (diffusion_sampling_algorithms.py:43-48: torch.randn_like, zero when noise_step <= 1, which
    // the coefficient table encodes as sigma = 0): 4 elements per Philox call
    const size_t n4 = (n + 3) / 4;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (size_t)gridDim.x * 256) {
      unsigned r[4];
      philox4x32_10((unsigned)q, (unsigned)(q >> 32), step, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
      float zz[4];
      box_muller(r[0], r[1], zz[0], zz[1]);
      box_muller(r[2], r[3], zz[2], zz[3]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t i = q * 4 + k;
        if (i < n) x[i] = s1 * (x[i] - s2 * eps[i]) + sg * zz[k];
      }
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
      float v = s1 * (x[i] - s2 * eps[i]);
      if (z) v += sg * z[i];
      x[i] = v;
    }
  }
  cursor_advance(cursor, nsteps);
}
extern "C" int sdm_ddpm_update(float* x, const float* eps, const float* z, const float* coef, size_t n,
                               unsigned long long seed, const int64_t* rng_step, int* cursor, int nsteps, sdm_stream_t s) {
  SDM_CHECK_ARG(x && eps && coef && n > 0, "ddpm_update: bad args");
  SDM_CHECK_ARG(cursor == nullptr || nsteps > 0, "ddpm_update: cursor without nsteps");
  SDM_LAUNCH(ddpm_update_kernel, dim3(ew_grid(rng_step && !z ? (n + 3) / 4 : n)), dim3(256), 0, (hipStream_t)s, x, eps, z,
                     coef, n, seed, (const long long*)rng_step, cursor, nsteps);
  SDM_CHECK_LAUNCH("ddpm_update");
  return SDM_OK;
}

__global__ void ddim_update_kernel(float* x, const float* eps, const float* coef, float* x0_out, size_t n, int* cursor,
                                   int nsteps) {
  const float inv = coef[0], sq1m = coef[1], sa1 = coef[2], sb1 = coef[3];
  const bool upd = coef[4] != 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float e = eps[i];
    const float x0 = inv * (x[i] - sq1m * e);
    if (x0_out) x0_out[i] = x0;
    if (upd) x[i] = sa1 * x0 + sb1 * e;
  }
  cursor_advance(cursor, nsteps);
}
extern "C" int sdm_ddim_update(float* x, const float* eps, const float* coef, float* x0_out, size_t n, int* cursor, int nsteps,
                               sdm_stream_t s) {
  SDM_CHECK_ARG(x && eps && coef && n > 0, "ddim_update: bad args");
  SDM_CHECK_ARG(cursor == nullptr || nsteps > 0, "ddim_update: cursor without nsteps");
  SDM_LAUNCH(ddim_update_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)s, x, eps, coef, x0_out, n, cursor, nsteps);
  SDM_CHECK_LAUNCH("ddim_update");
  return SDM_OK;
}

__global__ void cfg_combine_kernel(const float* e, float* out, float w, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float c = e[i], u = e[n + i];
    out[i] = u + w * (c - u);
  }
}
extern "C" int sdm_cfg_combine(const float* eps_2n, float* out, float w, size_t n_half, sdm_stream_t s) {
  SDM_CHECK_ARG(eps_2n && out && n_half > 0, "cfg_combine: bad args");
  SDM_LAUNCH(cfg_combine_kernel, dim3(ew_grid(n_half)), dim3(256), 0, (hipStream_t)s, eps_2n, out, w, n_half);
  SDM_CHECK_LAUNCH("cfg_combine");
  return SDM_OK;
}

__global__ void cold_update_kernel(float* x, const float* x0, const float* nz, const float* coef, size_t n, float* x0_keep,
                                   int* cursor, int nsteps) {
  const float a = coef[0], b = coef[1], c = coef[2], d = coef[3];
  const bool upd = coef[4] != 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v0 = x0[i];
    if (x0_keep) x0_keep[i] = v0;
    if (upd) {
      const float e = nz[i];
      x[i] = x[i] - (a * v0 + b * e) + (c * v0 + d * e);
    }
  }
  cursor_advance(cursor, nsteps);
}
extern "C" int sdm_cold_update(float* x, const float* x0_hat, const float* noise, const float* coef, size_t n, float* x0_keep,
                               int* cursor, int nsteps, sdm_stream_t s) {
  SDM_CHECK_ARG(x && x0_hat && noise && coef && n > 0, "cold_update: bad args");
  SDM_CHECK_ARG(cursor == nullptr || nsteps > 0, "cold_update: cursor without nsteps");
  SDM_LAUNCH(cold_update_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)s, x, x0_hat, noise, coef, n, x0_keep,
                     cursor, nsteps);
  SDM_CHECK_LAUNCH("cold_update");
  return SDM_OK;
}

__global__ void area_resize_kernel(const float* __restrict__ x, float* __restrict__ out, long long planes, int Hin, int Win,
                                   int Hout, int Wout) {
  const long long total = planes * Hout * Wout;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int j = (int)(e % Wout);
    const int i = (int)((e / Wout) % Hout);
    const long long pl = e / ((long long)Wout * Hout);
    const int y0 = (int)(((long long)i * Hin) / Hout), y1 = (int)((((long long)(i + 1)) * Hin + Hout - 1) / Hout);
    const int x0 = (int)(((long long)j * Win) / Wout), x1 = (int)((((long long)(j + 1)) * Win + Wout - 1) / Wout);
    const float* p = x + pl * Hin * Win;
    float acc = 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) acc += p[(long long)yy * Win + xx];
    out[e] = acc / (float)((y1 - y0) * (x1 - x0));
  }
}
extern "C" int sdm_area_resize(const float* x, float* out, int planes, int Hin, int Win, int Hout, int Wout, sdm_stream_t s) {
  SDM_CHECK_ARG(x && out && planes > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "area_resize: bad args");
  const size_t n = (size_t)planes * Hout * Wout;
  SDM_LAUNCH(area_resize_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)s, x, out, (long long)planes, Hin, Win,
                     Hout, Wout);
  SDM_CHECK_LAUNCH("area_resize");
  return SDM_OK;
}

__global__ __launch_bounds__(256) void step_begin_kernel(const int* cursor, const int64_t* t_table, const float* coef_table,
                                                         int ncoef, int nsteps, int64_t* t_cur, float* coef_cur,
                                                         const float* mod_time, const float* mod_cond, float* mods_cur, int rows,
                                                         int width) {
  const int raw = *cursor;
  const int c = raw < nsteps ? raw : nsteps - 1;      // reads stay inside the tables however often the step is replayed
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) t_cur[0] = t_table[c];
    if (threadIdx.x < ncoef) coef_cur[threadIdx.x] = coef_table[(long long)c * ncoef + threadIdx.x];
  }
  if (!mod_time) return;
  // the step's AdaGN modulation rows out of the per-run tables: mods_cur[i][:] = mod_time[c][:] (+ mod_cond[i][:])
  const int w4 = width >> 2;
  const f32x4* tr = (const f32x4*)(mod_time + (long long)c * width);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)rows * w4; e += (long long)gridDim.x * 256) {
    const int i = (int)(e / w4), k = (int)(e - (long long)i * w4);
    f32x4 v = tr[k];
    if (mod_cond) v += ((const f32x4*)(mod_cond + (long long)i * width))[k];
    ((f32x4*)(mods_cur + (long long)i * width))[k] = v;
  }
}
extern "C" int sdm_step_begin(const int* cursor, const int64_t* t_table, const float* coef_table, int ncoef, int nsteps,
                              int64_t* t_cur, float* coef_cur, const float* mod_time, const float* mod_cond, float* mods_cur,
                              int rows, int width, sdm_stream_t s) {
  SDM_CHECK_ARG(cursor && t_table && coef_table && t_cur && coef_cur && ncoef > 0 && ncoef <= 64 && nsteps > 0,
                "step_begin: bad args");
  SDM_CHECK_ARG(mod_time == nullptr || (mods_cur && rows > 0 && width > 0 && width % 4 == 0 && aligned16(mod_time) &&
                                        aligned16(mods_cur) && (mod_cond == nullptr || aligned16(mod_cond))),
                "step_begin: modulation tables need rows > 0, width %% 4 == 0 and 16-byte aligned buffers");
  long long g = mod_time ? ((long long)rows * (width / 4) + 255) / 256 : 1;
  if (g > 1024) g = 1024;
  SDM_LAUNCH(step_begin_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)s, cursor, t_table, coef_table, ncoef, nsteps,
                     t_cur, coef_cur, mod_time, mod_cond, mods_cur, rows, width);
  SDM_CHECK_LAUNCH("step_begin");
  return SDM_OK;
}

/* test / eager helper: ++*cursor (saturating at nsteps) as its own launch */
__global__ void step_advance_kernel(int* cursor, int nsteps) { cursor_advance(cursor, nsteps); }
extern "C" int sdm_step_advance(int* cursor, int nsteps, sdm_stream_t s) {
  SDM_CHECK_ARG(cursor && nsteps > 0, "step_advance: bad args");
  SDM_LAUNCH(step_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, cursor, nsteps);
  SDM_CHECK_LAUNCH("step_advance");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// fused multi-tensor Adam (no weight decay, bias-corrected; torch.optim.Adam semantics)
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ float adam_element(const sdm_adam_job& j, long long i, float gscale, float b1, float b2, float step,
                                              float rs, float eps) {
  const float g = j.g[i] * gscale;
  const float m = j.m[i] + (1.f - b1) * (g - j.m[i]);  // lerp, as torch does
  const float v = b2 * j.v[i] + (1.f - b2) * g * g;
  j.m[i] = m;
  j.v[i] = v;
  const float p = j.p[i] - step * (m / (sqrtf(v) * rs + eps));
  j.p[i] = p;
  return p;
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const sdm_adam_job* __restrict__ jobs, float lr, float b1,
                                                         float b2, float eps, float bc1, float bc2, float gscale,
                                                         const float* __restrict__ hyper) {
  __shared__ bf16_t tile[64][72];      // [c][r] of a 64 x 64 block of the updated weight (transposed copy); 144-byte rows
  const sdm_adam_job j = jobs[blockIdx.y];
  if (hyper) {
    lr = hyper[0];
    const float t = hyper[1];
    bc1 = 1.0f - powf(b1, t);
    bc2 = 1.0f - powf(b2, t);
  }
  const float step = lr / bc1;
  const float rs = 1.0f / sqrtf(bc2);
  if (!j.w16t) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < j.n; i += (long long)gridDim.x * 256) {
      const float p = adam_element(j, i, gscale, b1, b2, step, rs, eps);
      if (j.w16) j.w16[i] = f2bf(p);
    }
    return;
  }
  // The parameter is [T][R][Cc] (tap-major conv weight / Linear with T = 1) and the bf16 training path reads a
  // transposed copy [T][Cc][ldd] of it: walk it in 64 x 64 blocks, the update itself coalesced along Cc, the block's
  // new values through LDS so that the transposed copy is written in 128-byte runs along R.
  const int R = j.R, Cc = j.Cc, ldd = j.ldd;
  const int nr = (R + 63) >> 6, nc = (Cc + 63) >> 6;
  const long long T = j.n / ((long long)R * Cc);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (long long b = blockIdx.x; b < T * nr * nc; b += gridDim.x) {
    const long long t = b / (nr * nc);
    const int rem = (int)(b - t * nr * nc), r0 = (rem / nc) << 6, c0 = (rem % nc) << 6;
    const long long base = t * R * Cc;
    const int c = c0 + tx;
    for (int i = ty; i < 64; i += 4) {
      const int r = r0 + i;
      bf16_t h = 0;
      if (r < R && c < Cc) {
        const long long e = base + (long long)r * Cc + c;
        h = f2bf(adam_element(j, e, gscale, b1, b2, step, rs, eps));
        if (j.w16) j.w16[e] = h;
      }
      tile[tx][i] = h;
    }
    __syncthreads();
    bf16_t* o = j.w16t + t * Cc * ldd;
    if ((ldd & 7) == 0) {
      for (int e = threadIdx.x; e < 64 * 8; e += 256) {   // 8 consecutive r per lane: one 16-byte store
        const int ci = e >> 3, r8 = (e & 7) * 8;
        if (c0 + ci < Cc && r0 + r8 < R)                   // columns [R, ldd) of the copy stay zero (written once by the cast)
          *(u16x8*)(o + (long long)(c0 + ci) * ldd + r0 + r8) = *(const u16x8*)&tile[ci][r8];
      }
    } else {
      for (int i = ty; i < 64; i += 4)
        if (c0 + i < Cc && r0 + tx < R) o[(long long)(c0 + i) * ldd + r0 + tx] = tile[i][tx];
    }
    __syncthreads();
  }
}
extern "C" int sdm_adam_multi(const sdm_adam_job* jobs_dev, int njobs, int64_t max_n, float lr, float beta1, float beta2,
                              float eps, float bias_c1, float bias_c2, float grad_scale, const float* hyper_dev,
                              sdm_stream_t s) {
  SDM_CHECK_ARG(jobs_dev && njobs > 0 && max_n > 0, "adam_multi: bad args");
  long long g = (max_n + 255) / 256;
  if (g > 1024) g = 1024;
  SDM_LAUNCH(adam_multi_kernel, dim3((int)g, njobs), dim3(256), 0, (hipStream_t)s, jobs_dev, lr, beta1, beta2, eps,
                     bias_c1, bias_c2, grad_scale, hyper_dev);
  SDM_CHECK_LAUNCH("adam_multi");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// Input pipeline on the device (replaces cv2.imread + 4 DataLoader workers + the per-image Python
// hflip of train_diffusion.py:148-152,312-314): the whole dataset sits in HBM as decoded uint8 HWC (BGR) images; a batch
// is one gather that normalises through a 256-entry table ((v - 127.5) / 127.5, computed on the host exactly as the
// reference does: float64, then rounded to float32), flips horizontally where asked and writes NCHW fp32.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void u8_gather_nchw_kernel(const uint8_t* __restrict__ cache, const int64_t* __restrict__ idx,
                                                             const uint8_t* __restrict__ flip, const float* __restrict__ lut,
                                                             float* __restrict__ out, int C, int H, int W, long long total) {
  __shared__ float slut[256];
  slut[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const long long HW = (long long)H * W;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long b = e / (HW * C);
    long long r = e - b * HW * C;
    const int c = (int)sdm_idiv(r, HW);
    r -= (long long)c * HW;
    const int y = (int)(r / W), x = (int)(r - (long long)y * W);
    const int xs = (flip && flip[b]) ? W - 1 - x : x;
    out[e] = slut[cache[((idx[b] * H + y) * W + xs) * C + c]];
  }
}

extern "C" int sdm_u8_gather_nchw(const uint8_t* cache_nhwc, const int64_t* idx, const uint8_t* flip, const float* lut256,
                                  float* out_nchw, int batch, int C, int H, int W, sdm_stream_t s) {
  SDM_CHECK_ARG(cache_nhwc && idx && lut256 && out_nchw && batch > 0 && C > 0 && H > 0 && W > 0, "u8_gather_nchw: bad args");
  const long long total = (long long)batch * C * H * W;
  int grid = (int)((total + 255) / 256);
  if (grid > 8192) grid = 8192;
  SDM_LAUNCH(u8_gather_nchw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, cache_nhwc, idx, flip, lut256, out_nchw, C,
                     H, W, total);
  SDM_CHECK_LAUNCH("u8_gather_nchw");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// Fused tail of a conv block for no-grad forwards: [split-K finishing pass] + GroupNorm statistics + AdaGN modulation +
// affine apply + residual add, one workgroup per (image, group), two sweeps over that group's slice (the second one
// served by L2).  At N=1 every kernel of the sampling step costs a fixed ~4.5 us of dependent-launch latency whatever it
// does (profiles/r02_sample_n1_kernel_stats.csv): this replaces three such launches per conv block by one.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 gnf_value(const float* __restrict__ p, int raw, float alpha, const f32x4 b, int act) {
  f32x4 v = *(const f32x4*)p;
  if (raw) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = sdm_act(v[i] * alpha + b[i], act);
  }
  return v;
}

__global__ __launch_bounds__(1024) void gn_apply_fused_kernel(float* __restrict__ src, long long lds_, int raw, float alpha,
                                                              const float* __restrict__ bias, int act, int HW, int C,
                                                              int groups, float eps, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ mod,
                                                              int mod_ns, const float* __restrict__ x0, long long ldx0,
                                                              const float* __restrict__ x1, long long ldx1, int csplit,
                                                              float* __restrict__ out, long long ldo,
                                                              float* __restrict__ z_out, float* __restrict__ y_out, long long ldy,
                                                              float* __restrict__ mean_out, float* __restrict__ rstd_out, int hold) {
  __shared__ double red[2][16];
  __shared__ float s_stat[2];
  const int n = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cpg = C / groups, q4 = cpg >> 2;
  float* base = src + (long long)n * HW * lds_ + g * cpg;
  const long long tot4 = (long long)HW * q4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  double s1 = 0.0, s2 = 0.0;
  // Small maps (<= 8 channel quads per thread: everything below 64x64 at the U-Net's widths): the source values stay in
  // registers between the statistics sweep and the apply sweep -- one dependent trip to memory less in a launch that is
  // nothing but such trips (the N=1 sampling step runs twelve of these per denoise step)
  const bool held = hold && tot4 <= 1024 * 8;
  f32x4 sv[8];
  if (held) {
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long e = threadIdx.x + (long long)u * 1024;
      sv[u] = zero4;
      if (e < tot4) {
        const long long pix = sdm_idiv(e, q4);
        const int c = (int)(e - pix * q4) * 4;
        const f32x4 b = (raw && bias) ? *(const f32x4*)(bias + g * cpg + c) : zero4;
        sv[u] = *(const f32x4*)(base + pix * lds_ + c);
        f32x4 v = sv[u];
        if (raw) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = sdm_act(v[i] * alpha + b[i], act);
        }
        p1 += (v.x + v.y) + (v.z + v.w);
        p2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    }
    s1 = p1;
    s2 = p2;
  } else {
  for (long long e0 = threadIdx.x; e0 < tot4; e0 += 1024 * 8) {
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long e = e0 + (long long)u * 1024;
      if (e < tot4) {
        const long long pix = sdm_idiv(e, q4);
        const int c = (int)(e - pix * q4) * 4;
        const f32x4 b = (raw && bias) ? *(const f32x4*)(bias + g * cpg + c) : zero4;
        const f32x4 v = gnf_value(base + pix * lds_ + c, raw, alpha, b, act);
        p1 += (v.x + v.y) + (v.z + v.w);
        p2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    }
    s1 += p1;
    s2 += p2;
  }
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { a += red[0][w]; b += red[1][w]; }
    const double total = (double)HW * cpg;
    const double mean = a / total;
    double var = b / total - mean * mean;
    if (var < 0.0) var = 0.0;
    s_stat[0] = (float)mean;
    s_stat[1] = (float)(1.0 / sqrt(var + (double)eps));
    if (mean_out) {          // training: the backward needs the statistics
      mean_out[blockIdx.x] = s_stat[0];
      rstd_out[blockIdx.x] = s_stat[1];
    }
  }
  __syncthreads();
  const float mu = s_stat[0], rstd = s_stat[1];
  auto emit = [&](long long e, const f32x4 srcv) {
    const long long pix = sdm_idiv(e, q4);
    const int c = (int)(e - pix * q4) * 4;
    const int ch = g * cpg + c;
    float* sp = base + pix * lds_ + c;
    const f32x4 b = (raw && bias) ? *(const f32x4*)(bias + ch) : zero4;
    f32x4 v = srcv;
    if (raw) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = sdm_act(v[i] * alpha + b[i], act);
    }
    if (raw && y_out) {      // training: the conv's own outputs (pre-activation and activated rows) for its backward
      const long long yo = ((long long)n * HW + pix) * ldy + ch;
      if (z_out) {
        f32x4 zv = srcv;
#pragma unroll
        for (int i = 0; i < 4; ++i) zv[i] = zv[i] * alpha + b[i];
        *(f32x4*)(z_out + yo) = zv;
      }
      *(f32x4*)(y_out + yo) = v;
    }
    if (raw) *(f32x4*)sp = zero4;                       // the split-K scratch goes back zero-filled
    const f32x4 ga = *(const f32x4*)(gamma + ch), be = *(const f32x4*)(beta + ch);
    f32x4 o;
    if (mod) {
      const f32x4 sm = *(const f32x4*)(mod + (long long)n * mod_ns + ch);
#pragma unroll
      for (int i = 0; i < 4; ++i) {       // the same two-step form as sdm_gn_affine + sdm_affine_residual: scale*v + shift
        const float sc = rstd * ga[i] * sm[i];
        const float sh = (be[i] - mu * rstd * ga[i]) * sm[i] + sm[i];
        o[i] = v[i] * sc + sh;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = v[i] * (rstd * ga[i]) + (be[i] - mu * rstd * ga[i]);
    }
    const long long row = (long long)n * HW + pix;
    if (x0) o += (ch < csplit) ? *(const f32x4*)(x0 + row * ldx0 + ch) : *(const f32x4*)(x1 + row * ldx1 + (ch - csplit));
    *(f32x4*)(out + row * ldo + ch) = o;
  };
  if (held) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long e = threadIdx.x + (long long)u * 1024;
      if (e < tot4) emit(e, sv[u]);
    }
  } else {
    for (long long e = threadIdx.x; e < tot4; e += 1024) {
      const long long pix = sdm_idiv(e, q4);
      emit(e, *(const f32x4*)(base + pix * lds_ + (int)(e - pix * q4) * 4));
    }
  }
}

extern "C" int sdm_gn_apply_fused_train(const float* src, int64_t ld_src, int src_raw, float alpha, const float* bias, int act,
                                        int images, int HW, int C, int groups, float eps, const float* gamma, const float* beta,
                                        const float* mod, int mod_nstride, const float* x0, int64_t ldx0, const float* x1,
                                        int64_t ldx1, int csplit, float* out, int64_t ldo, float* z_out, float* y_out,
                                        int64_t ldy, float* mean, float* rstd, sdm_stream_t s);

extern "C" int sdm_gn_apply_fused(const float* src, int64_t ld_src, int src_raw, float alpha, const float* bias, int act,
                                  int images, int HW, int C, int groups, float eps, const float* gamma, const float* beta,
                                  const float* mod, int mod_nstride, const float* x0, int64_t ldx0, const float* x1,
                                  int64_t ldx1, int csplit, float* out, int64_t ldo, sdm_stream_t s) {
  return sdm_gn_apply_fused_train(src, ld_src, src_raw, alpha, bias, act, images, HW, C, groups, eps, gamma, beta, mod,
                                  mod_nstride, x0, ldx0, x1, ldx1, csplit, out, ldo, nullptr, nullptr, 0, nullptr, nullptr, s);
}

extern "C" int sdm_gn_apply_fused_train(const float* src, int64_t ld_src, int src_raw, float alpha, const float* bias, int act,
                                        int images, int HW, int C, int groups, float eps, const float* gamma, const float* beta,
                                        const float* mod, int mod_nstride, const float* x0, int64_t ldx0, const float* x1,
                                        int64_t ldx1, int csplit, float* out, int64_t ldo, float* z_out, float* y_out,
                                        int64_t ldy, float* mean, float* rstd, sdm_stream_t s) {
  SDM_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "gn_apply_fused: mean and rstd come together");
  SDM_CHECK_ARG(y_out == nullptr || (src_raw && aligned16(y_out) && ldy % 4 == 0 && (z_out == nullptr || aligned16(z_out))),
                "gn_apply_fused: y_out / z_out are the finished rows of RAW split-K sums (16-byte aligned, pitch %% 4 == 0)");
  SDM_CHECK_ARG(z_out == nullptr || y_out != nullptr, "gn_apply_fused: z_out needs y_out");
  SDM_CHECK_ARG(src && gamma && beta && out && images > 0 && HW > 0 && groups > 0 && C % groups == 0, "gn_apply_fused: bad args");
  SDM_CHECK_ARG(x0 == nullptr || x1 != nullptr || csplit >= C, "gn_apply_fused: csplit < C without second source");
  const int cpg = C / groups;
  SDM_CHECK_ARG(cpg % 4 == 0 && ld_src % 4 == 0 && ldo % 4 == 0 && aligned16(src) && aligned16(out) && aligned16(gamma) &&
                    aligned16(beta) && (mod == nullptr || (aligned16(mod) && mod_nstride % 4 == 0)) &&
                    (bias == nullptr || aligned16(bias)) &&
                    (x0 == nullptr || (aligned16(x0) && ldx0 % 4 == 0 && csplit % 4 == 0)) &&
                    (x1 == nullptr || (aligned16(x1) && ldx1 % 4 == 0)),
                "gn_apply_fused: needs channels-per-group %% 4 == 0 and 16-byte aligned rows / tables");
  SDM_CHECK_ARG(act >= 0 && act <= 2, "gn_apply_fused: bad act");
  SDM_LAUNCH(gn_apply_fused_kernel, dim3(images * groups), dim3(1024), 0, (hipStream_t)s, (float*)src, (long long)ld_src,
                     src_raw, alpha, bias, act, HW, C, groups, eps, gamma, beta, mod, mod_nstride, x0, (long long)ldx0, x1,
                     (long long)ldx1, csplit, out, (long long)ldo, z_out, y_out, (long long)ldy, mean, rstd, sdm_gn_hold());
  SDM_CHECK_LAUNCH("gn_apply_fused");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// NCHW fp32 [images][C][H][W] -> NHWC fp32 rows [images*H*W][Cp] (channels [C, Cp) zero): the 3- / 6-channel side of the
// edge convs as pixel rows, so that their WEIGHT gradients run on the matrix-core kernel (sdm_wgrad) like every other
// conv's (U_Net.py:55-60,121-128 backward).  Measured at 64x64 batch 4 (tools/scratch/edge_wgrad_ab.py): head conv
// 155.6 -> 49.5 us, stem conv 77.7 -> 52.4 us against the one-wave-per-pixel kernels + slab reduction.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int C, long long HW,
                                                           long long P, int Cp) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
    const long long n = sdm_idiv(p, HW), rem = p - n * HW;
    for (int c = 0; c < Cp; ++c) out[p * Cp + c] = c < C ? x[(n * C + c) * HW + rem] : 0.f;
  }
}
extern "C" int sdm_nchw_to_rows(const float* x_nchw, float* out_rows, int images, int C, int H, int W, int Cp, sdm_stream_t s) {
  SDM_CHECK_ARG(x_nchw && out_rows && images > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, "nchw_to_rows: bad args");
  const long long HW = (long long)H * W, P = HW * images;
  int grid = (int)((P + 255) / 256);
  if (grid > 8192) grid = 8192;
  SDM_LAUNCH(nchw_to_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, x_nchw, out_rows, C, HW, P, Cp);
  SDM_CHECK_LAUNCH("nchw_to_rows");
  return SDM_OK;
}
