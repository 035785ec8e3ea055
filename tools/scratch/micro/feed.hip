// Stand-alone microbenchmark (MI355X): how fast can a CU's workgroup move operand tiles L2 -> LDS, by which path?
// The contraction kernels (sdm_igemm8.hip, sdm_wgrad8.hip) are bound by exactly this rate (profiles/r04_wgrad8_dbg.txt:
// with MFMAs and fragment reads removed the K loop runs at 6.8 TB/s chip-wide, ~13 B/clk/CU; the vendor GEMM sustains ~17.5).
//   MODE 0: LDS-DMA   (global_load_lds_dwordx4, the kernels' path)
//   MODE 1: register-staged (global_load_dwordx4 -> VGPRs, ds_write_b128 DEPTH stagings later)
//   MODE 2: alternating stagings on the two paths (both in flight at once)
// One workgroup per CU (512 threads or 256), a ring of 8 x 16 KB half-tile slots, DEPTH stagings in flight, one barrier per
// staging (as a phase of the kernels has), every slot read back once with ds_read_b128 so that the data is consumed.  The
// source is a per-workgroup window of `win` KB walked repeatedly (L2-resident), rows of 128 B at a 1 KB pitch like a 512-channel
// bf16 activation tile.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o feed feed.hip ; run: ./feed [stagings per WG] [window KB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
#define SLOT 16384

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int NT, int DEPTH>
struct Feed {
  static constexpr int PER = SLOT / (NT * 16);           // 16-byte pieces per lane and staging (2 at 512 threads, 4 at 256)
  static constexpr int ROWS_PER_PASS = NT / 8;
  const char* base;
  long long win, lane_off;
  int pitch, piece, rpp, rot;
  char* smem;
  unsigned lds0;
  int tid, wave, nst;
  f32x4 regs[DEPTH + 1][PER];                            // register-staged data in flight
  f32x4 acc;
  __device__ __forceinline__ const char* src_of(int s, int p) const {
    // staging s covers 128 rows x 128 B: K-tile (s & 7) of row block (s >> 3), wrapped into the window
    // a staging covers (16384 / piece) rows x `piece` contiguous bytes; (1024 / piece) stagings walk a row block's 1 KB of k
    const int per_blk = 1024 / piece, rows = SLOT / piece;
    const long long blk = ((long long)(s / per_blk) * rows * pitch) % win;
    return base + blk + (long long)((s + rot) % per_blk) * piece + (long long)p * rpp * pitch + lane_off;
  }
  template <int Q>
  __device__ __forceinline__ void issue(int s) {
    const bool dma = MODE == 0 || (MODE == 2 && (s & 1) == 0);
    char* dst = smem + (s & 7) * SLOT + wave * 1024;
    if (dma) {
#pragma unroll
      for (int p = 0; p < PER; ++p)
        __builtin_amdgcn_global_load_lds((glb_ptr)src_of(s, p), (lds_ptr)(dst + p * (NT / 64) * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        const f32x4* g = (const f32x4*)src_of(s, p);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(regs[Q][p]) : "v"(g) : "memory");
      }
    }
  }
  // wait for staging s (DEPTH - 1 younger ones may stay in flight), land it, barrier, consume it, issue s + DEPTH
  template <int Q, int QN>
  __device__ __forceinline__ void step(int s) {
    wait_vm<(DEPTH - 1) * PER>();
    const bool dma = MODE == 0 || (MODE == 2 && (s & 1) == 0);
    if (!dma) {
      const unsigned dst = lds0 + (s & 7) * SLOT + tid * 16;
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        asm volatile("" : "+v"(regs[Q][p]));
        asm volatile("ds_write_b128 %0, %1" ::"v"(dst + p * NT * 16), "v"(regs[Q][p]) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds0 + (s & 7) * SLOT + ((tid * 16) & (SLOT - 1))) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(v));
    acc += v;
    __builtin_amdgcn_s_barrier();            // the slot of staging s + DEPTH (= s - 3 mod 8) is free
    if (s + DEPTH < nst) issue<QN>(s + DEPTH);
  }
};

template <int MODE, int NT, int DEPTH>
__global__ __launch_bounds__(NT) void feed(const char* __restrict__ src, long long win, int nst, float* __restrict__ sink, int pitch, int piece) {
  extern __shared__ __align__(16) char smem[];
  static_assert(DEPTH == 5, "unrolled for 5 stagings in flight");
  Feed<MODE, NT, DEPTH> f;
  f.tid = threadIdx.x;
  f.wave = __builtin_amdgcn_readfirstlane(f.tid >> 6);
  f.base = src + (long long)blockIdx.x * win;
  f.win = win;
  f.pitch = pitch; f.piece = piece % 100000; f.rpp = NT / (f.piece / 16); f.rot = piece >= 100000 ? (int)blockIdx.x : 0; piece = f.piece;
  f.lane_off = (long long)(f.tid / (piece / 16)) * pitch + (f.tid % (piece / 16)) * 16;   // (row of a pass, 16-byte chunk)
  f.smem = smem;
  f.lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  f.nst = nst;
  f.acc = f32x4{0.f, 0.f, 0.f, 0.f};
  f.template issue<0>(0); f.template issue<1>(1); f.template issue<2>(2); f.template issue<3>(3); f.template issue<4>(4);
  // steady state, unrolled by 6 so that register slots are compile-time: staging s uses register slot s % 6
  for (int s = 0; s < nst; s += 6) {
    f.template step<0, 5>(s);
    if (s + 1 < nst) f.template step<1, 0>(s + 1);
    if (s + 2 < nst) f.template step<2, 1>(s + 2);
    if (s + 3 < nst) f.template step<3, 2>(s + 3);
    if (s + 4 < nst) f.template step<4, 3>(s + 4);
    if (s + 5 < nst) f.template step<5, 4>(s + 5);
  }
  wait_vm<0>();
  if (f.acc[0] + f.acc[1] + f.acc[2] + f.acc[3] == 12345.678f) sink[blockIdx.x] = f.acc[0];
}

template <int MODE, int NT>
static void run(const char* name, const char* src, long long win, int nst, float* sink, int ncu, int pitch, int piece) {
  auto k = feed<MODE, NT, 5>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * SLOT);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k, dim3(ncu), dim3(NT), 8 * SLOT, 0, src, win, nst, sink, pitch, piece);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)ncu * nst * SLOT;
    if (rep == 2)
      printf("%-44s %8.1f us  %6.2f TB/s chip-wide  %5.1f B/clk/CU at 2.4 GHz  (%s)\n", name, ms * 1e3, bytes / ms / 1e9,
             bytes / ncu / (ms * 1e-3 * 2.4e9), hipGetErrorString(hipGetLastError()));
  }
}

// ---- register-staged with a DEEP prefetch: data in flight lives in VGPRs, not in reserved LDS slots, so the number of
// stagings in flight is not tied to the ring (the LDS slot is needed only from the landing ds_write to the consumer's read).
template <int I, int N, class F>
__device__ __forceinline__ void cfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    cfor<I + 1, N>(f);
  }
}
template <int NT, int DEPTH>
struct Deep {
  static constexpr int PER = SLOT / (NT * 16), ROWS_PER_PASS = NT / 8;
  const char* base;
  long long win, lane_off;
  int pitch, piece, rpp, rot;
  unsigned lds0;
  int tid, nst;
  f32x4 regs[DEPTH][PER];
  f32x4 acc;
  __device__ __forceinline__ const char* src_of(int s, int p) const {
    // a staging covers (16384 / piece) rows x `piece` contiguous bytes; (1024 / piece) stagings walk a row block's 1 KB of k
    const int per_blk = 1024 / piece, rows = SLOT / piece;
    const long long blk = ((long long)(s / per_blk) * rows * pitch) % win;
    return base + blk + (long long)((s + rot) % per_blk) * piece + (long long)p * rpp * pitch + lane_off;
  }
  template <int Q>
  __device__ __forceinline__ void issue(int s) {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const f32x4* g = (const f32x4*)src_of(s, p);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(regs[Q][p]) : "v"(g) : "memory");
    }
  }
  template <int Q>
  __device__ __forceinline__ void step(int s) {       // staging s sits in register slot Q = s % DEPTH
    wait_vm<(DEPTH - 1) * PER>();
    const unsigned dst = lds0 + (s & 3) * SLOT + tid * 16;
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      asm volatile("" : "+v"(regs[Q][p]));
      asm volatile("ds_write_b128 %0, %1" ::"v"(dst + p * NT * 16), "v"(regs[Q][p]) : "memory");
    }
    if (s + DEPTH < nst) issue<Q>(s + DEPTH);          // the registers are free again (the write has read them in order)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds0 + (s & 3) * SLOT + ((tid * 16) & (SLOT - 1))) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(v));
    acc += v;
    __builtin_amdgcn_s_barrier();
  }
};
template <int NT, int DEPTH>
__global__ __launch_bounds__(NT) void feed_deep(const char* __restrict__ src, long long win, int nst, float* __restrict__ sink, int pitch, int piece) {
  extern __shared__ __align__(16) char smem[];
  Deep<NT, DEPTH> f;
  f.tid = threadIdx.x;
  f.base = src + (long long)blockIdx.x * win;
  f.win = win;
  f.pitch = pitch; f.piece = piece % 100000; f.rpp = NT / (f.piece / 16); f.rot = piece >= 100000 ? (int)blockIdx.x : 0; piece = f.piece;
  f.lane_off = (long long)(f.tid / (piece / 16)) * pitch + (f.tid % (piece / 16)) * 16;
  f.lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  f.nst = nst;
  f.acc = f32x4{0.f, 0.f, 0.f, 0.f};
  cfor<0, DEPTH>([&](auto qc) { constexpr int q = decltype(qc)::value; if (q < nst) f.template issue<q>(q); });
  for (int s = 0; s < nst; s += DEPTH)
    cfor<0, DEPTH>([&](auto qc) { constexpr int q = decltype(qc)::value; if (s + q < nst) f.template step<q>(s + q); });
  wait_vm<0>();
  if (f.acc[0] + f.acc[1] + f.acc[2] + f.acc[3] == 12345.678f) sink[blockIdx.x] = f.acc[0];
}
template <int NT, int DEPTH>
static void run_deep(const char* name, const char* src, long long win, int nst, float* sink, int ncu, int pitch, int piece) {
  auto k = feed_deep<NT, DEPTH>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SLOT);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k, dim3(ncu), dim3(NT), 4 * SLOT, 0, src, win, nst, sink, pitch, piece);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)ncu * nst * SLOT;
    if (rep == 2)
      printf("%-44s %8.1f us  %6.2f TB/s chip-wide  %5.1f B/clk/CU at 2.4 GHz  (%s)\n", name, ms * 1e3, bytes / ms / 1e9,
             bytes / ncu / (ms * 1e-3 * 2.4e9), hipGetErrorString(hipGetLastError()));
  }
}

int main(int argc, char** argv) {
  const int nst = argc > 1 ? atoi(argv[1]) : 576;          // 576 stagings of 16 KB = 72 K-tiles of a 256x256 tile (9 MB per CU)
  const long long win = (long long)(argc > 2 ? atoi(argv[2]) : 512) * 1024;
  const int pitch = argc > 3 ? atoi(argv[3]) : 1024;
  const int rotate = argc > 5 ? atoi(argv[5]) : 0;       // 1: every workgroup starts its k walk at a different 128-B column
  const int piece = (argc > 4 ? atoi(argv[4]) : 128) + (rotate ? 100000 : 0);
  int ncu = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) == hipSuccess) ncu = prop.multiProcessorCount;
  char* src;
  float* sink;
  const size_t total = (size_t)ncu * win + (2 << 20);
  if (hipMalloc(&src, total) != hipSuccess || hipMalloc(&sink, ncu * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(src, 1, total);
  printf("%d CUs, %d stagings of 16 KB per workgroup, window %lld KB per workgroup, row pitch %d B, %d contiguous bytes per row, k walk rotated per workgroup: %d\n",
         ncu, nst, win / 1024, pitch, piece % 100000, rotate);
  run<0, 512>("LDS-DMA, 8 waves", src, win, nst, sink, ncu, pitch, piece);
  run<1, 512>("register-staged + ds_write_b128, 8 waves", src, win, nst, sink, ncu, pitch, piece);
  run<2, 512>("alternating DMA / register-staged, 8 waves", src, win, nst, sink, ncu, pitch, piece);
  run<0, 256>("LDS-DMA, 4 waves", src, win, nst, sink, ncu, pitch, piece);
  run<1, 256>("register-staged + ds_write_b128, 4 waves", src, win, nst, sink, ncu, pitch, piece);
  run<2, 256>("alternating DMA / register-staged, 4 waves", src, win, nst, sink, ncu, pitch, piece);
  run_deep<512, 3>("register-staged, 8 waves, 3 in flight", src, win, nst, sink, ncu, pitch, piece);
  run_deep<512, 5>("register-staged, 8 waves, 5 in flight", src, win, nst, sink, ncu, pitch, piece);
  run_deep<512, 8>("register-staged, 8 waves, 8 in flight", src, win, nst, sink, ncu, pitch, piece);
  run_deep<512, 12>("register-staged, 8 waves, 12 in flight", src, win, nst, sink, ncu, pitch, piece);
  run_deep<512, 16>("register-staged, 8 waves, 16 in flight", src, win, nst, sink, ncu, pitch, piece);
  run_deep<256, 4>("register-staged, 4 waves, 4 in flight", src, win, nst, sink, ncu, pitch, piece);
  run_deep<256, 8>("register-staged, 4 waves, 8 in flight", src, win, nst, sink, ncu, pitch, piece);
  run_deep<256, 12>("register-staged, 4 waves, 12 in flight", src, win, nst, sink, ncu, pitch, piece);
  return 0;
}
