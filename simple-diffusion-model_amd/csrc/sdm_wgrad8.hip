// bf16 weight gradient of a plain 3x3 stride-1 conv, three column taps per workgroup, phase-interleaved schedule: the
// kernel behind the large launches of sdm_wgrad_bf16 (the weight half of every nn.Conv2d backward of the reference,
// models/custom_layers.py:222-227, on feature maps of at least 16 x 16 with at least 64 pixels per K-step).
//
//   dw[(dy, dx)][i][j] (+)= alpha * sum_p G[p][i] * X[p + (dy, dx)][j]          G, X: bf16 pixel rows; dw fp32
//
// Why it exists: wgrad3_bf16_kernel (sdm_bf16_train.hip: 4 waves, 128 x 128, one barrier and a drained vmcnt per K-step)
// ran at ~700 TFLOP/s and was 20 % of a C5 bf16 training step (profiles/r03_c5_bf16_train_kernel_stats.csv); the forward
// contraction gained 1.45x from the phase-interleaved schedule on an unchanged feed (sdm_igemm8.hip).  This kernel keeps
// wgrad3's data reuse -- per 64-pixel K-step the gradient rows are staged once, the 66 activation pixels x0-1 .. x0+64 of
// image row y+dy once, and the three column taps dx = -1, 0, +1 are contracted from the same LDS image, a third of the
// L2 -> LDS bytes per flop of a per-tap tile -- and runs it the way sdm_igemm8.hip runs its tile:
//   * 8 waves = 2 (c_out) x 4 (c_in); a wave owns 64 x 32 outputs of each of the three taps (2 x 1 MFMA blocks of 32 x 32,
//     96 accumulator registers).
//   * LDS = a ring of 4 K-steps x {gradient stage 64 rows x 256 B, activation stage 68 rows x 256 B} = 132 KB, filled by
//     LDS-DMA exactly as the rows lie in HBM (K-major), 16-byte chunks XOR-swizzled by (row & 3) << 2 on the per-lane source
//     address and on the transposing fragment read (ds_read_b64_tr_b16), as in wgrad_bf16_kernel.  Three K-steps (99 KB)
//     are in flight; vmcnt is counted, never drained, until the tail.
//   * a PHASE = [L: transposing fragment reads, one stage issued, a counted s_waitcnt vmcnt] s_barrier [M: 8 x
//     v_mfma_f32_32x32x16_bf16 = one tap x 64 pixels] s_barrier.  K-step T has three phases (dx = -1, 0, +1):
//       L(T,0): the last 4 of the 16 gradient reads of step T + the 8 activation reads of shift 0;        issues G(T+3)
//       L(T,1): 8 activation reads of shift 1 + 6 gradient reads of step T+1;                              issues X(T+3)
//       L(T,2): 8 activation reads of shift 2 + 6 gradient reads of step T+1
//     (12-14 reads per L against 8 MFMAs per M: balanced); gradient fragments alternate between two register sets.
//   * waves 4-7 run ONE BARRIER BEHIND waves 0-3 (sdm_igemm8.hip): in every barrier interval one wave of each SIMD issues
//     its MFMAs while its partner on that SIMD reads fragments and issues DMA.
// Hazards.  RAW: G(T+1) is read from L(T,1) on and X(T+1) from L(T+1,0) on; the waits for them sit in L(T,0) / L(T,2), in
// front of those phases' first barrier -- with the one-barrier offset between the wave halves that is one barrier more than
// a reader needs.  WAR: G(T+3) overwrites the slot of G(T-1), last read in L(T-1,0); X(T+3) that of X(T-1), last read in
// L(T-1,2); a reader's lgkmcnt(0) sits behind the first barrier of its phase, and at least two further barriers separate
// that from the re-staging wave's issue.
// Narrow maps (W = 32 / 16: a K-step is 2 / 4 whole image rows, the activation "row" the same 66 consecutive pixels of the
// dy-shifted linear pixel index): products whose gradient pixel has no left / right neighbour are dropped by zeroing fixed
// fragment elements, as in wgrad3_bf16_kernel.  Bias gradient (column sums of G): the c_in-column-0 waves of the dy = -1
// workgroups of column tile 0 sum their gradient fragments with v_dot2c_f32_bf16 against (1, 1).
#include <stdlib.h>
#include <type_traits>

#include "sdm_bf16_int.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

#define W8_ROWB 256                      // bytes per LDS row: 128 channels
#define W8_GT (64 * W8_ROWB)             // gradient stage
#define W8_XT (68 * W8_ROWB)             // activation stage: 66 pixels, padded to whole 4-row DMA instructions
#define W8_ST (W8_GT + W8_XT)            // one K-step
#define W8_NST 4

__device__ float g_zero_chunk_w8[4];

__device__ __forceinline__ int xcd_contiguous_w8(int bid, int total) {   // see sdm_igemm.hip
  const int q = total >> 3, r = total & 7, x = bid & 7, i = bid >> 3;
  return x < r ? x * (q + 1) + i : r * (q + 1) + (x - r) * q + i;
}
template <int N>
__device__ __forceinline__ void wait_vm_w8() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define ICW(v) std::integral_constant<int, (v)>()
// compile-time loop: f(integral_constant<I>) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void cfor_w8(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    cfor_w8<I + 1, N>(f);
  }
}
// The transposing LDS read as inline asm.  Through the builtin the compiler sees an LDS load next to LDS-DMA stores it
// cannot disambiguate and puts s_waitcnt vmcnt(0) in front of the first read after every barrier: every stage in flight is
// drained once per phase and the counted waits above mean nothing (that is what the 4-wave wgrad kernels of
// sdm_bf16_train.hip were doing until round 4).  The price: the compiler does not know when the value arrives, so every use
// sits behind an explicit s_waitcnt lgkmcnt(0) and an opaque touch of the registers.
template <int OFF>
__device__ __forceinline__ void tr8(s16x4& v, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}

// DBG (timing experiments only, results are wrong when set; tools/wgrad_ab.py with SDM_WGRAD8_DBG): bit 0 = no DMA after the
// prologue (and no vmcnt waits), bit 1 = no fragment reads in the loop, bit 2 = no second barrier of a phase, bit 3 = no MFMAs
template <int WSH, int DBG = 0>
__global__ __launch_bounds__(512, 2) void wgrad8x3_bf16_kernel(const Wgrad3Launch L) {
  const sdm_wgrad_desc& d = L.d;
  extern __shared__ __align__(16) char smem_w8[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, h = lane >> 5;
  // 1-D grid of (pixel slice x tile) x 3 kernel rows.  The three kernel rows of one (slice, tile) read the SAME gradient rows and
  // activation rows shifted by one image row: they are made neighbours of one XCD's contiguous run (c, c + 1, c + 2), so that
  // they run at the same time on one L2 and two of the three reads hit it (as separate grid rows they landed on different XCDs
  // whenever the slice count was not a multiple of 8, and every kernel row streamed its operands from HBM: 3.0x algorithmic).
  const int c_ = xcd_contiguous_w8(blockIdx.x, gridDim.x);
  const int dyi = c_ % 3, wg_ = c_ / 3;
  const int ntiles_ = (gridDim.x / 3) / L.splitk;
  const int ks = wg_ / ntiles_, tile_ = wg_ - ks * ntiles_;
  const int tile_n = tile_ % L.ntn, tile_m = tile_ / L.ntn;
  const int dy = L.dy[dyi];
  const int m0 = tile_m * 128, n0 = tile_n * 128;
  const bf16_t* g0 = (const bf16_t*)d.g.p0;
  const bf16_t* x0 = (const bf16_t*)d.x.p0;
  const bf16_t* x1 = (const bf16_t*)d.x.p1;
  const bf16_t* zero = (const bf16_t*)g_zero_chunk_w8;
  const int H = d.Hk, W = d.Wk, HW = H * W;
  const long long P = (long long)d.images * HW;
  const int T = (int)(P >> 6);                                  // whole 64-pixel K-steps (host: P % 64 == 0)
  const int it0 = (int)((long long)T * ks / L.splitk), it1 = (int)((long long)T * (ks + 1) / L.splitk);
  const int nT = it1 - it0;
  if (nT <= 0) return;                                          // (block-uniform; the host never makes empty slices)

  // ---- DMA roles: 16 chunks per 256-byte row; one wave-instruction = 4 rows; pass i covers rows 32 i + 4 wave + (lane >> 4)
  const int drow = 4 * wave + (lane >> 4);
  const int lch = (lane & 15) ^ (((lane >> 4) & 3) << 2);       // logical chunk behind this lane's physical slot (row & 3)
  const int cm = m0 + 8 * lch, cn = n0 + 8 * lch;
  const bool gch = cm < ((d.M + 7) & ~7), xch = cn < ((d.N + 7) & ~7);
  const bool xsrc0 = cn < d.x.csplit;
  const bf16_t* xb = xsrc0 ? x0 : x1;
  const long long xld = xsrc0 ? d.x.ld0 : d.x.ld1;
  const int xcol = xsrc0 ? cn : cn - d.x.csplit;
  const long long goff0 = (long long)drow * d.g.ld0 + cm, goff1 = goff0 + 32 * d.g.ld0;
  const long long xoff0 = (long long)drow * xld + xcol, xoff1 = xoff0 + 32 * xld;
  const long long xoff2 = (long long)(64 + (lane >> 4)) * xld + xcol;          // wave 0: patch rows 64..67

  int un, uy, ux;                                               // position of the next K-step to issue (uniform)
  {
    const long long p0 = (long long)it0 * 64;
    un = (int)(p0 / HW);
    const int rem = (int)(p0 - (long long)un * HW);
    uy = rem / W;
    ux = rem - uy * W;
  }
  int issued = 0;                                               // K-steps whose G stage has been issued
  char* lds_w = smem_w8 + wave * 1024;
  auto issue_g = [&]() {       // gradient stage of K-step `issued`
    char* dst = lds_w + (issued & (W8_NST - 1)) * W8_ST;
    const long long grow0 = (((long long)un * H + uy) * W + ux) * d.g.ld0;
    const bf16_t* q0 = gch ? g0 + (grow0 + goff0) : zero;
    const bf16_t* q1 = gch ? g0 + (grow0 + goff1) : zero;
    __builtin_amdgcn_global_load_lds((glb_ptr)q0, (lds_ptr)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)q1, (lds_ptr)(dst + 32 * W8_ROWB), 16, 0, 0);
  };
  auto issue_x = [&]() {       // activation stage of the same K-step; advances the position
    char* dst = lds_w + (issued & (W8_NST - 1)) * W8_ST + W8_GT;
    const int xy = uy + dy;
    const bool yok = (unsigned)xy < (unsigned)H;
    const long long xrow0 = (((long long)un * H + xy) * W + ux - 1) * xld;        // patch row 0 = pixel x0 - 1
    const int lin0 = xy * W + ux - 1;                                             // ... as a pixel index inside its image
    auto ok = [&](int r) {
      return WSH >= 6 ? (xch && yok && (unsigned)(ux - 1 + r) < (unsigned)W) : (xch && (unsigned)(lin0 + r) < (unsigned)HW);
    };
    const bf16_t* q0 = ok(drow) ? xb + (xrow0 + xoff0) : zero;
    const bf16_t* q1 = ok(drow + 32) ? xb + (xrow0 + xoff1) : zero;
    __builtin_amdgcn_global_load_lds((glb_ptr)q0, (lds_ptr)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)q1, (lds_ptr)(dst + 32 * W8_ROWB), 16, 0, 0);
    if (wave == 0) {                           // patch rows 64..67 (pixels x0+63 .. x0+66; 66, 67 are never read: zeros)
      const int r = 64 + (lane >> 4);
      const bf16_t* q2 = (r < 66 && ok(r)) ? xb + (xrow0 + xoff2) : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr)q2, (lds_ptr)(smem_w8 + (issued & (W8_NST - 1)) * W8_ST + W8_GT + 64 * W8_ROWB), 16, 0, 0);
    }
    if (WSH >= 6) {
      ux += 64;
      if (ux >= W) { ux = 0; uy += 1; }
    } else {
      uy += 64 >> WSH;
    }
    if (uy >= H) { uy = 0; un += 1; }
    ++issued;
  };
  // counted waits.  Per K-step a wave issues 2 (G) + 2 (X) DMA instructions, wave 0 one more for the X stage.
  //   after G(T+3) in L(T,0): G(T+1) must have landed; younger: X(T+1) G(T+2) X(T+2) G(T+3)
  //   in L(T,2) (last issue X(T+3)): X(T+1) must have landed; younger: G(T+2) X(T+2) G(T+3) X(T+3)
  // -> 8 instructions may stay in flight (wave 0: 10); once the issue has stopped (tail) the counter is drained.
  auto wait_steady = [&]() {
    if (wave == 0) wait_vm_w8<10>();
    else wait_vm_w8<8>();
  };

  // ---- transposing-read roles (see wgrad_bf16_kernel): 16-lane group row q, channel quad p4
  const int i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3, chalf = (lane >> 4) & 1;
  int aoff[2], boff[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int col = wr * 64 + i * 32 + 16 * chalf + 4 * p4;
    aoff[i] = (8 * h + q) * W8_ROWB + (((col >> 3) ^ (q << 2)) << 4) + ((col & 4) << 1);
  }
#pragma unroll
  for (int sft = 0; sft < 3; ++sft) {            // patch rows shifted by 1 + dx = 0, 1, 2
    const int col = wc * 32 + 16 * chalf + 4 * p4;
    boff[sft] = W8_GT + (8 * h + q + sft) * W8_ROWB + (((col >> 3) ^ (((q + sft) & 3) << 2)) << 4) + ((col & 4) << 1);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem_w8;

  s16x4 fal[2][4][2], fah[2][4][2];   // gradient fragments [set][kk][block], rows +0..3 / +4..7 of the lane's 8-row group
  s16x4 fbl[4], fbh[4];               // activation fragments of the current shift [kk]
  f32x16 acc[3][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;
  float bsum[2] = {0.f, 0.f};
  const bool do_bias = d.dbias != nullptr && tile_n == 0 && dyi == 0 && wc == 0;     // wave-uniform

  // gradient fragment f (0..7) of a K-step: kk = f >> 1, block = f & 1; a fragment is two transposing reads
  auto read_a = [&](auto setc, auto f0c, auto f1c, unsigned stage) {
    constexpr int set = decltype(setc)::value, f0 = decltype(f0c)::value, f1 = decltype(f1c)::value;
    const unsigned b0 = lds0 + stage + aoff[0], b1 = lds0 + stage + aoff[1];
    cfor_w8<f0, f1>([&](auto fc) {
      constexpr int f = decltype(fc)::value, kk = f >> 1, blk = f & 1;
      tr8<kk * 16 * W8_ROWB>(fal[set][kk][blk], blk ? b1 : b0);
      tr8<(kk * 16 + 4) * W8_ROWB>(fah[set][kk][blk], blk ? b1 : b0);
    });
  };
  auto read_b = [&](auto sftc, unsigned stage) {
    constexpr int sft = decltype(sftc)::value;
    const unsigned b = lds0 + stage + boff[sft];
    cfor_w8<0, 4>([&](auto kc) {
      constexpr int kk = decltype(kc)::value;
      tr8<kk * 16 * W8_ROWB>(fbl[kk], b);
      tr8<(kk * 16 + 4) * W8_ROWB>(fbh[kk], b);
    });
  };
  auto mfma_tap = [&](auto setc, auto sftc) {
    constexpr int set = decltype(setc)::value, sft = decltype(sftc)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the fragments were written by asm reads: opaque here, so that nothing that uses them is scheduled above the wait
    s16x8 fa[4][2], fb[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      asm volatile("" : "+v"(fbl[kk]), "+v"(fbh[kk]));
      fb[kk] = __builtin_shufflevector(fbl[kk], fbh[kk], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        asm volatile("" : "+v"(fal[set][kk][i]), "+v"(fah[set][kk][i]));
        fa[kk][i] = __builtin_shufflevector(fal[set][kk][i], fah[set][kk][i], 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
    if (WSH < 6 && sft != 1) {          // narrow maps: gradient pixels at x == 0 (dx = -1) / x == W - 1 (dx = +1) have no neighbour
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bool grp = WSH == 4 || (sft == 0 ? (kk & 1) == 0 : (kk & 1) == 1);
        if (grp && h == (sft == 0 ? 0 : 1)) fb[kk][sft == 0 ? 0 : 7] = 0;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (!(DBG & 8)) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[sft][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[kk][i]),
                                                                __builtin_bit_cast(bf16x8, fb[kk]), acc[sft][i], 0, 0, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) asm volatile("" ::"v"(fa[kk][0]), "v"(fa[kk][1]), "v"(fb[kk]));
    }
    __builtin_amdgcn_s_setprio(0);
    if (sft == 0 && do_bias) {          // column sums of this K-step's gradient rows (a lane: channel l & 31, 8 pixels per fragment)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // (the pair is built from the 16-bit elements: a dword extracted through a bit cast of the whole fragment is
            // miscompiled by this toolchain -- every pair read the fragment's FIRST dword)
            const s16x2 pr = {fa[kk][i][2 * e], fa[kk][i][2 * e + 1]};
            bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pr), __builtin_bit_cast(bf16x2_t, 0x3f803f80u),
                                                      bsum[i], false);
          }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: K-steps 0..2 in flight, K-step 0 landed, 12 of its 16 gradient reads done, then the wave-half offset
  for (int p = 0; p < 3; ++p)
    if (p < nT) { issue_g(); issue_x(); }
  if (nT >= 3) wait_steady();      // G(0), X(0) landed (younger: G(1) X(1) G(2) X(2))
  else wait_vm_w8<0>();
  __builtin_amdgcn_s_barrier();
  read_a(ICW(0), ICW(0), ICW(6), 0u);
  if (wr == 1) __builtin_amdgcn_s_barrier();

  // K-step t (0-based inside the slice), register set t & 1; LDS slot t & 3
  auto kstep = [&](auto setc, int t) {
    constexpr int set = decltype(setc)::value;
    const unsigned so = (unsigned)((t & (W8_NST - 1)) * W8_ST), sn = (unsigned)(((t + 1) & (W8_NST - 1)) * W8_ST);
    const bool more = t + 1 < nT, steady = t + 3 < nT;
    // phase 0
    if (!(DBG & 2)) { read_a(ICW(set), ICW(6), ICW(8), so); read_b(ICW(0), so); }
    if (!(DBG & 1)) {
      if (steady) { issue_g(); wait_steady(); }
      else wait_vm_w8<0>();
    }
    __builtin_amdgcn_s_barrier();
    mfma_tap(ICW(set), ICW(0));
    if (!(DBG & 4)) __builtin_amdgcn_s_barrier();
    // phase 1
    if (!(DBG & 2)) { read_b(ICW(1), so); if (more) read_a(ICW(set ^ 1), ICW(0), ICW(3), sn); }
    if (!(DBG & 1) && steady) issue_x();
    __builtin_amdgcn_s_barrier();
    mfma_tap(ICW(set), ICW(1));
    if (!(DBG & 4)) __builtin_amdgcn_s_barrier();
    // phase 2
    if (!(DBG & 2)) { read_b(ICW(2), so); if (more) read_a(ICW(set ^ 1), ICW(3), ICW(6), sn); }
    if (!(DBG & 1)) {
      if (steady) wait_steady();
      else wait_vm_w8<0>();
    }
    __builtin_amdgcn_s_barrier();
    mfma_tap(ICW(set), ICW(2));
    if (!(DBG & 4)) __builtin_amdgcn_s_barrier();
  };
  for (int t = 0; t < nT; t += 2) {
    kstep(ICW(0), t);
    if (t + 1 < nT) kstep(ICW(1), t + 1);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();     // every wave has passed the same number of barriers

  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);
      const int ch = m0 + wr * 64 + i * 32 + (lane & 31);
      if (h == 0 && ch < d.M) atomicAdd(&d.dbias[ch], v);
    }
  }

  const int l31 = lane & 31;
#pragma unroll
  for (int sft = 0; sft < 3; ++sft) {
    float* dw = d.dw + (long long)L.tapw[dyi][sft] * d.w_tap_stride;
    const int col = n0 + wc * 32 + l31;
    if (col >= d.N) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= d.M) continue;
        float* o = dw + (long long)row * d.ldw + col;
        const float v = acc[sft][i][r] * d.alpha;
        if (L.atomic_out) atomicAdd(o, v);
        else *o = v;
      }
    }
  }
}

int sdm_launch_wgrad8x3_bf16(const Wgrad3Launch& L3, hipStream_t st) {
  const sdm_wgrad_desc* d = &L3.d;
  constexpr size_t lds = (size_t)W8_NST * W8_ST;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)wgrad8x3_bf16_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)wgrad8x3_bf16_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)wgrad8x3_bf16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const long P = (long)d->images * d->Hk * d->Wk;
  {
    const double px = (double)d->images;
    const double g_b = px * d->g.H * d->g.W * d->M, x_b = px * d->x.H * d->x.W * d->N, w_b = (double)d->M * d->N * 9;
    sdm_prof_shape(d->M, d->N, (int)P, 9, L3.splitk, 2.0 * (g_b + x_b) + 4.0 * w_b);
  }
  sdm_prof_pre(st);
  const dim3 grid(cdiv(d->M, 128) * L3.ntn * L3.splitk * 3, 1, 1);
  static const char* dbg_s = getenv("SDM_WGRAD8_DBG");
  const int dbg = dbg_s ? atoi(dbg_s) : 0;
  if (dbg && d->Wk >= 64) {
#define W8_DBG_CASE(v)                                                                                                        \
  case v:                                                                                                                     \
    (void)hipFuncSetAttribute((const void*)wgrad8x3_bf16_kernel<6, v>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    SDM_LAUNCH((wgrad8x3_bf16_kernel<6, v>), grid, dim3(512), lds, st, L3);                                                   \
    break;
    switch (dbg) {
      W8_DBG_CASE(1) W8_DBG_CASE(2) W8_DBG_CASE(3) W8_DBG_CASE(4) W8_DBG_CASE(8) W8_DBG_CASE(9) W8_DBG_CASE(10) W8_DBG_CASE(11)
      default: return sdm_set_error(SDM_ERR_ARG, "SDM_WGRAD8_DBG: unknown experiment %d", dbg);
    }
#undef W8_DBG_CASE
  } else
  if (d->Wk >= 64) SDM_LAUNCH(wgrad8x3_bf16_kernel<6>, grid, dim3(512), lds, st, L3);
  else if (d->Wk == 32) SDM_LAUNCH(wgrad8x3_bf16_kernel<5>, grid, dim3(512), lds, st, L3);
  else SDM_LAUNCH(wgrad8x3_bf16_kernel<4>, grid, dim3(512), lds, st, L3);
  sdm_prof_post(st, 3, 2.0 * d->M * d->N * (double)P * 9);
  SDM_CHECK_LAUNCH("wgrad8x3_bf16");
  return SDM_OK;
}
