// Microbenchmark (MI355X): what one LDS-DMA instruction costs a SIMD that is otherwise issuing MFMAs.
// Two waves per SIMD; each wave: [4 x v_mfma_f32_32x32x2_f32][ND DMA instructions of 16 bytes per lane into LDS, source = a small
// L2-resident buffer][optionally a counted wait].  Forms: global_load_lds_dwordx4 (64-bit per-lane address) and
// raw_buffer_load_lds (buffer descriptor + 32-bit per-lane offset).  Prints cycles added per DMA instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int ND, int FORM, int BF>
__global__ __launch_bounds__(512) void k(const float* src, float* out, int iters) {
  extern __shared__ __align__(16) float smem[];
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  const float* p = src + ((blockIdx.x * 512 + threadIdx.x) & 4095) * 4;      // 64 KB window: L2 / TCP resident
  float* dst = smem + __builtin_amdgcn_readfirstlane(wave) * 256 * 8;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
  const int voff = ((blockIdx.x * 512 + threadIdx.x) & 4095) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (BF) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      if (FORM == 0) __builtin_amdgcn_global_load_lds((glb_ptr)p, (lds_ptr)(dst + (i & 7) * 256), 16, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(dst + (i & 7) * 256), 16, voff, 0, 0, 0);
    }
    if (ND) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND < 8 ? ND : 8) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float r = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) r += acc[q];
  if (r == 12345.678f) out[0] = r + smem[lane];
}

template <int ND, int FORM>
static void run(const float* src, float* d) {
  const int iters = 20000, wgs = 256, w = 8;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<ND, FORM, 0>), dim3(wgs), dim3(64 * w), 8 * 8 * 1024, 0, src, d, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<ND, FORM, 0>), dim3(wgs), dim3(64 * w), 8 * 8 * 1024, 0, src, d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.4e9 / iters / 2;     // per wave-iteration share of a SIMD (two waves per SIMD)
  printf("%s  DMA/4MFMA %d : %7.1f cycles per wave-iteration (MFMA alone: 256)  -> %6.1f cycles per DMA instruction\n",
         FORM == 0 ? "global_load_lds_dwordx4 " : "raw_buffer_load_lds x4  ", ND, cyc, ND ? (cyc - 259.0) / ND : 0.0);
}

int main() {
  float *src, *d;
  (void)hipMalloc(&src, 1 << 20);
  (void)hipMemset(src, 0, 1 << 20);
  (void)hipMalloc(&d, 4);
  run<0, 0>(src, d);
  run<1, 0>(src, d);
  run<2, 0>(src, d);
  run<4, 0>(src, d);
  run<1, 1>(src, d);
  run<2, 1>(src, d);
  run<4, 1>(src, d);
  return 0;
}
