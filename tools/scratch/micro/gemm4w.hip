// Stand-alone microbenchmark (MI355X): the vendor GEMM's SHAPE of kernel (profiles/r04_vendor_gemm_pmc.json: macro tile 256x256x64,
// FOUR waves of 128x128, one wave per SIMD) written with this repository's LDS image and LDS-DMA feed, against the 8-wave
// phase-interleaved loop of gemm8ph.hip on the same problem  C[m][n] = sum_t sum_k A[m + sh_t][k] * B[t][n][k].
//   * 4 waves = 2 (M) x 2 (N); a wave owns 128 x 128 outputs = 8 x 8 blocks of v_mfma_f32_16x16x32_bf16 (256 accumulator registers).
//   * LDS = 2 slots x {A tile 256 rows x 128 B, B tile 256 rows x 128 B} = 128 KB; a tile is staged by 16 LDS-DMA instructions per lane.
//   * ONE barrier per K-tile, in the MIDDLE of it: [reads of k-step 1] [64 MFMAs of k-step 0] wait(frags, next tile's DMA) barrier
//     [DMA of tile T+2 into the slot just read] [reads of tile T+1's k-step 0] [64 MFMAs of k-step 1].  Fragment reads and DMA issue
//     ride under the MFMAs of the same wave (in-order issue, different pipes); 128 MFMAs = 2048 matrix cycles between barriers.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm4w gemm4w.hip ; run: ./gemm4w [M N Kc taps W iters]
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

#define TILE_B 32768     // one operand tile: 256 rows x 128 bytes
#define SLOT_B 65536     // A tile | B tile
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


template <int I, int N, class F>
__device__ __forceinline__ void cfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    cfor<I + 1, N>(f);
  }
}

__global__ __launch_bounds__(256, 1) void gemm4w(const Args a) {
  extern __shared__ __align__(16) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ntiles = gridDim.x, ntm = ntiles / a.ntn;
  const int tile = xcd_contig(blockIdx.x, ntiles);
  const int tile_n = a.mfast ? tile / ntm : tile % a.ntn, tile_m = a.mfast ? tile % ntm : tile / a.ntn;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int kpt = a.Kc >> 6;              // K-tiles per tap
  const int nkt = kpt * a.taps;

  // ---- staging: one instruction of the 4 waves = 32 rows x 128 B; lane -> (row tid >> 3 of the pass, physical chunk tid & 7)
  const int srow = tid >> 3;
  const int cc8 = 8 * ((tid & 7) ^ ((srow >> 1) & 7));
  const bf16_t* pA = a.A + ((long long)(m0 + srow) + a.shift[0]) * a.lda + cc8;
  const bf16_t* pB = a.B + (long long)(n0 + srow) * a.Kc + cc8;
  const long long sA = 32ll * a.lda, sB = 32ll * a.Kc;      // pass stride (elements)
  int kcA = 0, tapA = 0, kcB = 0;
  char* lds_w = smem + wave * 1024;
  auto stage = [&](int slot) {
    char* dst = lds_w + slot * SLOT_B;
#pragma unroll
    for (int p = 0; p < 8; ++p) __builtin_amdgcn_global_load_lds((glb_ptr)(pA + p * sA), (lds_ptr)(dst + p * 4096), 16, 0, 0);
#pragma unroll
    for (int p = 0; p < 8; ++p) __builtin_amdgcn_global_load_lds((glb_ptr)(pB + p * sB), (lds_ptr)(dst + TILE_B + p * 4096), 16, 0, 0);
    long long stepA = 64, stepB = 64;
    if (++kcA == kpt) {
      kcA = 0;
      const int t = tapA++;
      stepA = (t + 1 < a.taps ? (a.shift[t + 1] - a.shift[t]) * a.lda : 0) - a.Kc + 64;
    }
    if (++kcB == kpt) { kcB = 0; stepB = (long long)a.N * a.Kc - a.Kc + 64; }
    pA += stepA;
    pB += stepB;
  };

  // ---- fragment reads: lane l reads row (l & 15) of a 16-row block, 16-byte chunk (l >> 4) of a 32-wide k-step
  const int l15 = lane & 15, lq = lane >> 4, sw = (l15 >> 1) & 7;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned oa = (unsigned)((wr * 128 + l15) * 128 + ((lq ^ sw) << 4));             // k-step 0; k-step 1: offset ^ 64
  const unsigned ob = (unsigned)(TILE_B + (wc * 128 + l15) * 128 + ((lq ^ sw) << 4));
  f32x4 fa[2][8], fb[2][8];        // [k-step][block]
  f32x4 acc[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto read_a = [&](auto ksc, int slot) {
    constexpr int ks = decltype(ksc)::value;
    const unsigned ba = lds0 + slot * SLOT_B + (ks ? (oa ^ 64u) : oa);
    cfor<0, 8>([&](auto bc) { constexpr int b = decltype(bc)::value; dsr<b * 2048>(fa[ks][b], ba); });
  };
  auto read_b = [&](auto ksc, int slot) {
    constexpr int ks = decltype(ksc)::value;
    const unsigned bb = lds0 + slot * SLOT_B + (ks ? (ob ^ 64u) : ob);
    cfor<0, 8>([&](auto bc) { constexpr int b = decltype(bc)::value; dsr<b * 2048>(fb[ks][b], bb); });
  };
  auto touch = [&](f32x4 (&A)[8], f32x4 (&B)[8]) {      // fragments written by asm reads: opaque from here on
#pragma unroll
    for (int b = 0; b < 8; ++b) asm volatile("" : "+v"(A[b]), "+v"(B[b]));
  };
  auto mfma_half = [&](auto ksc, auto hc) {      // rows 4 hc .. 4 hc + 3 of the wave tile: 32 MFMAs
    constexpr int ks = decltype(ksc)::value, r0 = 4 * decltype(hc)::value;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = r0; r < r0 + 4; ++r)
#pragma unroll
      for (int c = 0; c < 8; ++c)
        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ks][c]),
                                                            __builtin_bit_cast(bf16x8, fa[ks][r]), acc[r][c], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
#define K0 std::integral_constant<int, 0>()
#define K1 std::integral_constant<int, 1>()

  // ---- prologue: tiles 0 and 1 staged, tile 0 landed, its k-step 0 on the way into set 0
  stage(0);
  if (nkt > 1) { stage(1); wait_vm<16>(); }
  else wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  read_a(K0, 0);
  read_b(K0, 0);

  for (int T = 0; T < nkt; ++T) {
    const int s = T & 1;
    wait_lgkm0();                                                // set 0 = (T, k-step 0) has landed (issued half a tile ago)
    touch(fa[0], fb[0]);
    read_a(K1, s);                                               // k-step 1 of this tile rides under the MFMAs of k-step 0
    mfma_half(K0, K0);
    read_b(K1, s);
    mfma_half(K0, K1);
    wait_lgkm0();                                                // set 1 landed
    touch(fa[1], fb[1]);
    wait_vm<0>();                                                // the next tile has landed (issued one tile ago)
    __builtin_amdgcn_s_barrier();                                // everyone has read this tile's fragments; everyone's DMA is visible
    if (T + 2 < nkt) stage(s);                                   // tile T + 2 into the slot just read
    if (T + 1 < nkt) read_a(K0, s ^ 1);
    mfma_half(K1, K0);
    if (T + 1 < nkt) read_b(K0, s ^ 1);
    mfma_half(K1, K1);
  }

  // ---- epilogue (benchmark form): lane holds 4 consecutive n of one m per accumulator
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int m = m0 + wr * 128 + r * 16 + l15;
      const int n = n0 + wc * 128 + c * 16 + lq * 4;
      if (m < a.M && n < a.N) {
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[r][c][e]);
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


static double run(const char* name, const Args& a, int iters, const int* dsm, const int* dsn, int cnt, const std::vector<float>& href) {
  auto kern = gemm4w;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT_B);
  const int ntm = (a.M + 255) / 256;
  const dim3 grid(ntm * a.ntn);
  (void)hipMemset(a.C, 0xFF, (size_t)a.M * a.N * 2);
  hipLaunchKernelGGL(kern, grid, dim3(256), 2 * SLOT_B, 0, a);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(e)); return 0; }
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
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), 2 * SLOT_B, 0, a);
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), 2 * SLOT_B, 0, a);
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
  for (int rep = 0; rep < 2; ++rep)
    for (int mf = 0; mf < 2; ++mf) {
      a.mfast = mf;
      char nm[64];
      snprintf(nm, sizeof nm, "gemm4w mfast=%d", mf);
      run(nm, a, iters, dsm, dsn, cnt, href);
    }
  return 0;
}
