// Stand-alone microbenchmark (MI355X): the phase-interleaved 256x256 bf16 contraction that igemm8_bf16_kernel in
// csrc/sdm_bf16.hip is built from, on a plain problem  C[m][n] = sum_t sum_k A[m + sh_t][k] * B[t][n][k]  (t = "taps": row
// shifts of the activation rows, as a 3x3 conv walks them; taps = 1 is a plain GEMM).  bf16 operands, reduction-contiguous,
// fp32 accumulation, bf16 result.
//
// Structure (after the 8-phase template of cdna_hip_programming.md section 5; written from its description):
//   * 8 waves = 2 (M) x 4 (N); a wave owns 128 x 64 of the tile as four 64 x 32 quadrants (qa, qb); quadrant (qa, qb)
//     lives in rows qa*128 + wr*64.., columns qb*128 + wc*32.. -- so a quadrant's operands come from ONE 128-row half
//     of the A tile and ONE 128-row half of the B tile, and a half-tile (128 rows x 64 k = 16 KB) is what one phase stages.
//   * LDS: 8 half-tile slots (2 K-tiles x {B0, A0, B1, A1}); staging s = 4T + j goes to slot s & 7 by LDS-DMA.
//   * a phase = [L: fragment reads of one half-tile's share (8 A or 4 B ds_read_b128), ONE half-tile staged (2 DMA per
//     lane), counted vmcnt] barrier [M: 16 x v_mfma_f32_16x16x32_bf16 = one quadrant x 64 k] barrier.
//     Phase f reads staging f, issues staging f + 6, waits for staging f + 1 (5 stagings stay in flight).
//   * waves 4-7 run one barrier behind waves 0-3: in every barrier interval one wave of each SIMD issues MFMAs while its
//     partner reads fragments and issues DMA.
//   * B fragments alternate between two register sets (B0 of tile T+1 is read during the last quadrant of tile T), so
//     the period is 8 phases = 2 K-tiles.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm8ph gemm8ph.hip ; run: ./gemm8ph [M N Kc taps W]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

#define HT_BYTES 16384   // one half-tile: 128 rows x 128 bytes
#define MAXT 16

struct Args {
  const bf16_t* A;      // [rows][lda], already offset so that row 0 + min shift is valid
  const bf16_t* B;      // [taps][N][Kc]
  bf16_t* C;            // [M][N]
  int M, N, Kc, taps, lda, ntn, mfast;
  long long shift[MAXT];   // activation-row shift of tap t
};

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int OFF>
__device__ __forceinline__ void dsr(f32x4& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ bf16_t f2bf(float f) {
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ int xcd_contig(int bid, int total) {
  const int q = total >> 3, r = total & 7, x = bid & 7, i = bid >> 3;
  return x < r ? x * (q + 1) + i : r * (q + 1) + (x - r) * q + i;
}

// STAG: waves 4-7 one barrier behind; PRIO: s_setprio around the MFMA cluster; DEPTH: stagings left in flight at a wait
template <bool STAG, bool PRIO, int DEPTH>
__global__ __launch_bounds__(512, 2) void gemm8ph(const Args a) {
  extern __shared__ __align__(16) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntiles = gridDim.x, ntm = ntiles / a.ntn;
  const int tile = xcd_contig(blockIdx.x, ntiles);
  const int tile_n = a.mfast ? tile / ntm : tile % a.ntn, tile_m = a.mfast ? tile % ntm : tile / a.ntn;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int kpt = a.Kc >> 6;              // K-tiles per tap
  const int nkt = kpt * a.taps;
  const int S = 4 * nkt;                  // stagings

  // ---- staging: lane -> (row of a 64-row pass, physical 16-byte slot); logical chunk = slot ^ ((row >> 1) & 7)
  const int srow = tid >> 3;
  const int cc8 = 8 * ((tid & 7) ^ ((srow >> 1) & 7));
  const bf16_t* pB[2][2];
  const bf16_t* pA[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = h * 128 + i * 64 + srow;
      pB[h][i] = a.B + (long long)(n0 + r) * a.Kc + cc8;
      pA[h][i] = a.A + ((long long)(m0 + r) + a.shift[0]) * a.lda + cc8;
    }
  int kcA[2] = {0, 0}, kcB[2] = {0, 0}, tapAh[2] = {0, 0};
  char* lds_w = smem + wave * 1024;
  // staging s = 4T + j : j = 0 B-half 0, 1 A-half 0, 2 B-half 1, 3 A-half 1 ; slot = s & 7 (compile-time in the loop)
  auto stage = [&](auto jc, auto slotc) {
    constexpr int j = decltype(jc)::value, slot = decltype(slotc)::value, h = j >> 1;
    char* dst = lds_w + slot * HT_BYTES;
    if constexpr ((j & 1) == 0) {
      __builtin_amdgcn_global_load_lds((glb_ptr)pB[h][0], (lds_ptr)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr)pB[h][1], (lds_ptr)(dst + 8192), 16, 0, 0);
      // next K-tile of this half: 64 further along the row, or the next tap's matrix
      long long step = 64;
      if (++kcB[h] == kpt) { kcB[h] = 0; step = (long long)a.N * a.Kc - a.Kc + 64; }
      pB[h][0] += step;
      pB[h][1] += step;
    } else {
      __builtin_amdgcn_global_load_lds((glb_ptr)pA[h][0], (lds_ptr)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr)pA[h][1], (lds_ptr)(dst + 8192), 16, 0, 0);
      long long step = 64;
      if (++kcA[h] == kpt) {
        kcA[h] = 0;
        const int t = tapAh[h]++;
        step = (t + 1 < a.taps ? (a.shift[t + 1] - a.shift[t]) * a.lda : 0) - a.Kc + 64;
      }
      pA[h][0] += step;
      pA[h][1] += step;
    }
  };
  auto wait_younger = [&](int n) {   // at most n stagings (2 DMA each) still in flight
    if (n >= DEPTH) wait_vm<2 * DEPTH>();
    else if (DEPTH > 4 && n == 4) wait_vm<8>();
    else if (DEPTH > 3 && n == 3) wait_vm<6>();
    else if (DEPTH > 2 && n == 2) wait_vm<4>();
    else if (DEPTH > 1 && n == 1) wait_vm<2>();
    else wait_vm<0>();
  };

  // ---- fragment reads: lane l reads row (l & 15) of a 16-row block, 16-byte chunk (l >> 4) of a 32-wide k-step
  const int l15 = lane & 15, lq = lane >> 4, sw = (l15 >> 1) & 7;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned oa0 = (unsigned)((wr * 64 + l15) * 128 + ((lq ^ sw) << 4));   // k-step 0 ; k-step 1 = offset ^ 64
  const unsigned ob0 = (unsigned)((wc * 32 + l15) * 128 + ((lq ^ sw) << 4));
  const unsigned ra[2][2] = {{lds0 + oa0, lds0 + (oa0 ^ 64u)}, {lds0 + oa0 + 65536u, lds0 + (oa0 ^ 64u) + 65536u}};   // [upper 64 KB][k-step]
  const unsigned rb[2][2] = {{lds0 + ob0, lds0 + (ob0 ^ 64u)}, {lds0 + ob0 + 65536u, lds0 + (ob0 ^ 64u) + 65536u}};

  f32x4 fa[2][4];       // [k-step][row block]
  f32x4 fx[2][2], fy[2][2];   // two B sets: [k-step][col block]
  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[i][j][r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto read_a = [&](auto slotc) {
    constexpr int slot = decltype(slotc)::value, up = slot >> 2, off = (slot & 3) * HT_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      dsr<off>(fa[ks][0], ra[up][ks]);
      dsr<off + 2048>(fa[ks][1], ra[up][ks]);
      dsr<off + 4096>(fa[ks][2], ra[up][ks]);
      dsr<off + 6144>(fa[ks][3], ra[up][ks]);
    }
  };
  auto read_b = [&](auto slotc, f32x4 (&fb)[2][2]) {
    constexpr int slot = decltype(slotc)::value, up = slot >> 2, off = (slot & 3) * HT_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      dsr<off>(fb[ks][0], rb[up][ks]);
      dsr<off + 2048>(fb[ks][1], rb[up][ks]);
    }
  };
  auto mfma_q = [&](f32x4 (&q)[4][2], f32x4 (&fb)[2][2]) {
    // the fragment registers were written by asm reads: make them opaque here so that no MFMA moves above the wait
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(fa[ks][r]));
#pragma unroll
      for (int c = 0; c < 2; ++c) asm volatile("" : "+v"(fb[ks][c]));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          q[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ks][c]),
                                                            __builtin_bit_cast(bf16x8, fa[ks][r]), q[r][c], 0, 0, 0);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: stagings 0 .. 6, wait for 0 and 1, read B0 of tile 0
  int issued = 0;
#define IC(v) std::integral_constant<int, (v)>()
  if (0 < S) { stage(IC(0), IC(0)); ++issued; }
  if (1 < S) { stage(IC(1), IC(1)); ++issued; }
  if (2 < S) { stage(IC(2), IC(2)); ++issued; }
  if (3 < S) { stage(IC(3), IC(3)); ++issued; }
  if (4 < S) { stage(IC(0), IC(4)); ++issued; }
  if (5 < S) { stage(IC(1), IC(5)); ++issued; }
  if (6 < S) { stage(IC(2), IC(6)); ++issued; }
  wait_younger(issued - 2);
  __builtin_amdgcn_s_barrier();
  read_b(IC(0), fx);
  if (STAG && wr == 1) __builtin_amdgcn_s_barrier();

  // phase P of iteration I: f = 8 I + P + 1
  auto phase = [&](auto Pc, int f) {
    constexpr int P = decltype(Pc)::value;
    constexpr int j = (P + 1) & 3, rslot = (P + 1) & 7, sslot = (P + 7) & 7, sj = (P + 3) & 3;
    constexpr bool odd = P >= 4;            // tile parity: B0 of the current tile in fy (odd) or fx (even)
    // L segment
    if constexpr (j == 1 || j == 3) read_a(IC(rslot));
    else if constexpr (j == 2) { if constexpr (odd) read_b(IC(rslot), fx); else read_b(IC(rslot), fy); }
    else { if (f < S) { if constexpr (odd) read_b(IC(rslot), fx); else read_b(IC(rslot), fy); } }
    if (f + 6 < S) stage(IC(sj), IC(sslot));
    wait_younger(S - f - 2);
    __builtin_amdgcn_s_barrier();
    // M segment
    wait_lgkm0();
    if constexpr (j == 1) { if constexpr (odd) mfma_q(acc[0][0], fy); else mfma_q(acc[0][0], fx); }
    else if constexpr (j == 2) { if constexpr (odd) mfma_q(acc[0][1], fx); else mfma_q(acc[0][1], fy); }
    else if constexpr (j == 3) { if constexpr (odd) mfma_q(acc[1][1], fx); else mfma_q(acc[1][1], fy); }
    else { if constexpr (odd) mfma_q(acc[1][0], fy); else mfma_q(acc[1][0], fx); }
    __builtin_amdgcn_s_barrier();
  };
  const int niter = (nkt + 1) >> 1;
  for (int I = 0; I < niter; ++I) {
    const int f = 8 * I + 1;
    phase(IC(0), f);
    phase(IC(1), f + 1);
    phase(IC(2), f + 2);
    phase(IC(3), f + 3);
    if (2 * I + 1 < nkt) {
      phase(IC(4), f + 4);
      phase(IC(5), f + 5);
      phase(IC(6), f + 6);
      phase(IC(7), f + 7);
    }
  }
  if (STAG && wr == 0) __builtin_amdgcn_s_barrier();

  // ---- epilogue (benchmark form): lane holds 4 consecutive n of one m per accumulator
#pragma unroll
  for (int qa = 0; qa < 2; ++qa)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int m = m0 + qa * 128 + wr * 64 + r * 16 + l15;
          const int n = n0 + qb * 128 + wc * 32 + c * 16 + lq * 4;
          if (m < a.M && n < a.N) {
            u16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[qa][qb][r][c][e]);
            *(u16x4*)(a.C + (long long)m * a.N + n) = o;
          }
        }
}

// reference for sampled outputs
__global__ void ref_kernel(const Args a, const int* sm, const int* sn, float* out, int cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  const int m = sm[i], n = sn[i];
  double s = 0;
  for (int t = 0; t < a.taps; ++t) {
    const bf16_t* ar = a.A + ((long long)m + a.shift[t]) * a.lda;
    const bf16_t* br = a.B + ((long long)t * a.N + n) * a.Kc;
    for (int k = 0; k < a.Kc; ++k)
      s += (double)__uint_as_float((unsigned)ar[k] << 16) * (double)__uint_as_float((unsigned)br[k] << 16);
  }
  out[i] = (float)s;
}

static bf16_t h_f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16;
  return (bf16_t)u;
}
static float h_bf2f(bf16_t b) {
  unsigned u = (unsigned)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

template <bool STAG, bool PRIO, int DEPTH>
static double run(const char* name, const Args& a, int iters, const int* dsm, const int* dsn, float* dref, int cnt,
                  const std::vector<float>& href) {
  auto kern = gemm8ph<STAG, PRIO, DEPTH>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * HT_BYTES);
  const int ntm = (a.M + 255) / 256;
  const dim3 grid(ntm * a.ntn);
  (void)hipMemset(a.C, 0xFF, (size_t)a.M * a.N * 2);
  hipLaunchKernelGGL(kern, grid, dim3(512), 8 * HT_BYTES, 0, a);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(e)); return 0; }
  // check the sampled outputs
  std::vector<bf16_t> hc((size_t)a.M * a.N);
  (void)hipMemcpy(hc.data(), a.C, hc.size() * 2, hipMemcpyDeviceToHost);
  std::vector<int> hm(cnt), hn(cnt);
  (void)hipMemcpy(hm.data(), dsm, cnt * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hn.data(), dsn, cnt * 4, hipMemcpyDeviceToHost);
  double worst = 0, scale = 0;
  for (int i = 0; i < cnt; ++i) scale = fmax(scale, fabs(href[i]));
  int bad = 0;
  for (int i = 0; i < cnt; ++i) {
    const double d = fabs(h_bf2f(hc[(size_t)hm[i] * a.N + hn[i]]) - href[i]);
    worst = fmax(worst, d);
    if (!(d <= 0.006 * scale + 0.01 * fabs(href[i]))) ++bad;
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(512), 8 * HT_BYTES, 0, a);
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(512), 8 * HT_BYTES, 0, a);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / iters;
  const double tf = 2.0 * a.M * a.N * (double)a.Kc * a.taps / (us * 1e-6) / 1e12;
  printf("%-28s %8.1f us %8.1f TFLOP/s   check: worst |d| %.4g of scale %.4g, %d / %d outside tolerance\n", name, us, tf,
         worst, scale, bad, cnt);
  fflush(stdout);
  return tf;
}

int main(int argc, char** argv) {
  int M = 32768, N = 512, Kc = 512, taps = 9, W = 64;
  if (argc > 5) { M = atoi(argv[1]); N = atoi(argv[2]); Kc = atoi(argv[3]); taps = atoi(argv[4]); W = atoi(argv[5]); }
  const int iters = argc > 6 ? atoi(argv[6]) : 20;
  if (M % 256 || N % 256 || Kc % 64 || taps > MAXT) { printf("bad shape\n"); return 1; }
  const int pad = W + 1;
  const size_t arows = (size_t)M + 2 * pad;
  std::vector<bf16_t> hA(arows * Kc), hB((size_t)taps * N * Kc);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
  for (auto& v : hA) v = h_f2bf(rnd());
  for (auto& v : hB) v = h_f2bf(rnd());
  bf16_t *dA, *dB, *dC;
  (void)hipMalloc(&dA, hA.size() * 2);
  (void)hipMalloc(&dB, hB.size() * 2);
  (void)hipMalloc(&dC, (size_t)M * N * 2);
  (void)hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  Args a;
  memset(&a, 0, sizeof(a));
  a.A = dA + (size_t)pad * Kc;
  a.B = dB;
  a.C = dC;
  a.M = M; a.N = N; a.Kc = Kc; a.taps = taps; a.lda = Kc; a.ntn = N / 256; a.mfast = 0;
  if (taps == 9) {
    int t = 0;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) a.shift[t++] = (long long)dy * W + dx;
  } else {
    for (int t = 0; t < taps; ++t) a.shift[t] = t - taps / 2;
  }
  const int cnt = 4096;
  std::vector<int> hm(cnt), hn(cnt);
  for (int i = 0; i < cnt; ++i) {
    s = s * 1664525u + 1013904223u; hm[i] = (s >> 4) % M;
    s = s * 1664525u + 1013904223u; hn[i] = (s >> 4) % N;
  }
  for (int i = 0; i < 256; ++i) { hm[i] = (i * 37) % 256 + (i & 1 ? M - 256 : 0); hn[i] = (i * 53) % 256 + (i & 2 ? N - 256 : 0); }
  int *dsm, *dsn;
  float* dref;
  (void)hipMalloc(&dsm, cnt * 4);
  (void)hipMalloc(&dsn, cnt * 4);
  (void)hipMalloc(&dref, cnt * 4);
  (void)hipMemcpy(dsm, hm.data(), cnt * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dsn, hn.data(), cnt * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(ref_kernel, dim3((cnt + 255) / 256), dim3(256), 0, 0, a, dsm, dsn, dref, cnt);
  std::vector<float> href(cnt);
  (void)hipMemcpy(href.data(), dref, cnt * 4, hipMemcpyDeviceToHost);
  printf("M=%d N=%d Kc=%d taps=%d (K=%d) W=%d : %d tiles of 256x256, %d K-tiles\n", M, N, Kc, taps, Kc * taps, W,
         (M / 256) * (N / 256), Kc / 64 * taps);
  for (int rep = 0; rep < 2; ++rep) {
    for (int mf = 0; mf < 2; ++mf) {
      a.mfast = mf;
      char nm[64];
      snprintf(nm, sizeof nm, "stag prio d5 mfast=%d", mf);
      run<true, true, 5>(nm, a, iters, dsm, dsn, dref, cnt, href);
    }
    a.mfast = 0;
    run<true, false, 5>("stag noprio d5", a, iters, dsm, dsn, dref, cnt, href);
    run<false, true, 5>("lockstep prio d5", a, iters, dsm, dsn, dref, cnt, href);
    run<true, true, 3>("stag prio d3", a, iters, dsm, dsn, dref, cnt, href);
    run<true, true, 2>("stag prio d2", a, iters, dsm, dsn, dref, cnt, href);
  }
  return 0;
}
