// Correctness probe for raw buffer LDS-DMA on gfx950: data lands lane-linearly in LDS, out-of-range lanes read zero, and how the
// scalar offset takes part in the range check.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr;

__global__ void k(const float* src, float* out, unsigned nbytes, unsigned soff, unsigned flags) {
  __shared__ __align__(16) float sm[256];
  for (int i = threadIdx.x; i < 256; i += 64) sm[i] = -1.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, flags);
  // lane l reads 16 bytes at byte offset l * 32 (every other chunk); lanes >= 48 point far outside
  unsigned voff = threadIdx.x * 32;
  if (threadIdx.x >= 48) voff = 0xFFFFFF00u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)sm, 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = sm[i];
}

int main() {
  float *src, *out, h[4096], o[256];
  for (int i = 0; i < 4096; ++i) h[i] = (float)i;
  (void)hipMalloc(&src, sizeof(h));
  (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  (void)hipMalloc(&out, sizeof(o));
  for (unsigned flags : {0x00020000u, 0x00027000u}) {
    for (unsigned cfg = 0; cfg < 3; ++cfg) {
      const unsigned nbytes = cfg == 2 ? 1024u : 16384u, soff = cfg == 1 ? 256u : 0u;
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, nbytes, soff, flags);
      (void)hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
      printf("flags %08x num_records %5u soffset %3u: lane0 %.0f %.0f | lane1 %.0f | lane31 %.0f | lane32 %.0f (byte 1024+soff) | lane47 %.0f | lane48 %.0f lane63 %.0f\n",
             flags, nbytes, soff, o[0], o[3], o[4], o[31 * 4], o[32 * 4], o[47 * 4], o[48 * 4], o[63 * 4 + 3]);
    }
  }
  return 0;
}
