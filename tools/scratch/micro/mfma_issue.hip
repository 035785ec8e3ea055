// Microbenchmark (MI355X): how much non-MFMA instruction issue can hide behind v_mfma_f32_32x32x2_f32 on one SIMD?
// Grid: one workgroup of W waves per CU-slot (W = 4: one wave per SIMD; W = 8: two per SIMD); each wave runs ITER iterations of
// [4 MFMAs on CH independent accumulators][NV dependent-free VALU adds][NS SALU adds].  Prints MFMA TFLOP/s for a sweep.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_issue mfma_issue.hip && ./mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NS, int CH>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x16 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  int s = blockIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q % CH], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(b));
#pragma unroll
    for (int i = 0; i < NS; ++i) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s));
  }
  float r = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int q = 0; q < 16; ++q) r += acc[c][q];
#pragma unroll
  for (int i = 0; i < 8; ++i) r += v[i];
  if (r == 12345.678f) out[0] = r + s;
}

template <int NV, int NS, int CH>
static void run(int waves_per_wg, int wgs, float* d) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, NS, CH>), dim3(wgs), dim3(64 * waves_per_wg), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, NS, CH>), dim3(wgs), dim3(64 * waves_per_wg), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)wgs * waves_per_wg * iters * 4 * 32 * 32 * 2 * 2;
  printf("waves/SIMD %d  chains %d  VALU/4MFMA %2d  SALU/4MFMA %2d : %7.1f TFLOP/s  (%.0f cycles per 4 MFMAs per SIMD at 2.4 GHz)\n",
         waves_per_wg / 4, CH, NV, NS, flops / ms * 1e-9, ms * 1e-3 * 2.4e9 / iters / (waves_per_wg / 4));
}

int main() {
  float* d;
  hipMalloc(&d, 4);
  const int wgs = 256;
  for (int w : {4, 8}) {
    run<0, 0, 1>(w, wgs, d);
    run<0, 0, 2>(w, wgs, d);
    run<0, 0, 4>(w, wgs, d);
    run<8, 0, 1>(w, wgs, d);
    run<16, 0, 1>(w, wgs, d);
    run<32, 0, 1>(w, wgs, d);
    run<16, 0, 4>(w, wgs, d);
    run<32, 0, 4>(w, wgs, d);
    run<0, 16, 1>(w, wgs, d);
    run<16, 10, 1>(w, wgs, d);
    run<16, 10, 4>(w, wgs, d);
  }
  return 0;
}
