// bf16 contraction, 256x256 tile, phase-interleaved ("8-phase") schedule: the kernel behind the large launches of
// sdm_igemm_bf16 (every nn.Conv2d / ConvTranspose2d / token Linear / attention GEMM of the reference,
// models/custom_layers.py:138-156,173-227, at >= 256 tiles of 256x256).
//
// Why it exists (profiles/r04_vendor_gemm_pmc.json, profiles/r04_gemm8ph.txt): the one-barrier-per-K-step 256x256 tile of
// sdm_bf16.hip ran the 512-channel 3x3 conv on 8 x 64x64 at 870 TFLOP/s where the vendor GEMM reaches 1225 on the same box.
// The vendor kernel is MT256x256x64 on 16x16 MFMAs; what it does differently is the SCHEDULE, not the feed path.  This
// kernel, on the same LDS-DMA feed and the same LDS image as before, reaches 1240-1260 on that problem:
//   * 8 waves = 2 (M) x 4 (N).  A wave owns 128 x 64 outputs as four 64 x 32 QUADRANTS (qa, qb) in tile rows
//     qa*128 + wr*64 .. and tile columns qb*128 + wc*32 ..: a quadrant's A rows lie in ONE 128-row half of the A tile, its
//     B rows in ONE 128-row half of the B tile.  A half-tile (128 rows x 64 k bf16 = 16 KB) is the unit of staging.
//   * LDS = 8 half-tile slots (two K-tiles x {B0, A0, B1, A1}).  Staging s = 4 T + j goes to slot s & 7 by LDS-DMA (two
//     global_load_lds_dwordx4 per lane), rows of 128 bytes, 16-byte chunks XOR-swizzled by (row >> 1) & 7 on the per-lane
//     SOURCE address and on the fragment read (conflict-free for the 16x16x32 operand read too).
//   * a PHASE = [L: the fragment reads of ONE half-tile (8 x ds_read_b128 for an A half, 4 for a B half), ONE staging
//     issued, a counted s_waitcnt vmcnt] s_barrier [M: 16 x v_mfma_f32_16x16x32_bf16 = one quadrant x 64 k] s_barrier.
//     Phase f reads staging f, issues staging f + 6 (its slot was last read in phase f - 2) and waits for staging f + 1:
//     five stagings (80 KB) stay in flight, vmcnt never drains inside the loop.
//   * waves 4-7 run ONE BARRIER BEHIND waves 0-3 (they pass an extra barrier before the loop, waves 0-3 one after it): in
//     every barrier interval one wave of each SIMD issues its 16 MFMAs while its partner on that SIMD reads fragments and
//     issues DMA.  Without the offset: 990 instead of 1250 TFLOP/s.
//   * B fragments alternate between two register sets (B0 of tile T+1 is read during the last quadrant of tile T), hence
//     a period of 8 phases = 2 K-tiles; A fragments use one set (a quadrant's A half is dead when the next is read).
//   * v_mfma_f32_16x16x32_bf16 with the roles swapped (weights as the A operand): a lane ends up with FOUR CONSECUTIVE
//     output columns of one row per accumulator: one ds_write_b128 per accumulator stages the tile for the (rolled, compact)
//     epilogue loop; the column statistics of attention's pass 1 are taken from the registers directly.
// Hazards of the schedule (a read of staging s in phase s; the wait for it in phase s - 1, in front of that phase's first
// barrier; with the one-barrier offset between the wave halves that is exactly one barrier more than the reader needs), WAR
// (a slot is restaged two phases after its last read; the reader's lgkmcnt(0) sits behind the first barrier of its phase,
// i.e. in front of both barriers the restaging waves pass first).
//
// Addressing is the FAST form of igemm_bf16_kernel: K and the concat split are multiples of 64, a lane's DMA source is a
// pointer advanced by one add per K-tile (rows outside the image point at a zero chunk with step 0) and rebuilt when the
// (tap, concat source) segment changes.  Same descriptor semantics (taps, sub-pixel phases, batch strides, XCD order).
#include <stdlib.h>
#include <type_traits>

#define SDM_FAST_SIGMOID
#include "sdm_bf16_int.h"

#define HT8 16384            // bytes of a half-tile slot
#define RING8 (8 * HT8)

__device__ float g_zero_chunk_8[4];

// -DSDM_IGEMM8_TIMING (tools/igemm8_timing.sh builds a second library with it; never in the product build): wave 0 of every
// workgroup stamps s_memrealtime (100 MHz) at entry, after the prologue tables, when the first operands have arrived, after the
// K loop and after the epilogue; sdm_debug_igemm8_times copies the table out.
#ifdef SDM_IGEMM8_TIMING
#define T8_MAXWG 16384
__device__ unsigned long long g_t8[T8_MAXWG][6];
#define T8_STAMP(i)                                                                                              \
  do {                                                                                                           \
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < T8_MAXWG) g_t8[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
extern "C" int sdm_debug_igemm8_times(unsigned long long* host, int nwg) {
  if (nwg > T8_MAXWG) nwg = T8_MAXWG;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_t8), (size_t)nwg * 6 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#else
#define T8_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ int xcd_contiguous_8(int bid, int total) {   // see sdm_igemm.hip
  const int q = total >> 3, r = total & 7, x = bid & 7, i = bid >> 3;
  return x < r ? x * (q + 1) + i : r * (q + 1) + (x - r) * q + i;
}
template <int N>
__device__ __forceinline__ void wait_vm8() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm8() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int OFF>
__device__ __forceinline__ void dsr8(f32x4& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
#define IC8(v) std::integral_constant<int, (v)>()

// TAPIN: K walk with the TAPS INNERMOST (64-channel chunk outer): the nine taps of a chunk re-read activation rows that are
// one pixel / one image row apart, so an XCD's tiles keep the chunk's rows (~0.8 MB) in their 4 MB L2 across the taps instead of
// re-streaming the whole 4 MB of tile rows once per tap (profiles/r04_igemm8_fwd512_pmc.json: L2-miss reads 4.8x algorithmic).
// Per-row validity of every tap is a bit mask made in the prologue, the taps' row offsets and weight-slice indices sit in the
// lanes of two registers (v_readlane by the uniform tap index): no table reads in the loop.
template <bool TAPIN>
__global__ __launch_bounds__(512, 2) void igemm8_bf16_kernel(const BfLaunch L) {
  const sdm_igemm_desc& d = L.d;
  extern __shared__ __align__(16) char smem8[];
  T8_STAMP(0);
  long long* s_orow = (long long*)(smem8 + RING8);
  int* s_tap = (int*)(s_orow + 256);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tile_id = xcd_contiguous_8(blockIdx.x, gridDim.x), ntm_ = gridDim.x / L.ntn;
  const int tile_n = L.mfast ? tile_id / ntm_ : tile_id % L.ntn, tile_m = L.mfast ? tile_id % ntm_ : tile_id / L.ntn;
  const int phase = blockIdx.y;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int b0 = blockIdx.z / d.bdiv, b1 = blockIdx.z % d.bdiv;
  const bf16_t* a0 = (const bf16_t*)d.a.p0 + b0 * d.a_bs0 + b1 * d.a_bs1;
  const bf16_t* a1 = (const bf16_t*)d.a.p1;
  const bf16_t* bw = (const bf16_t*)d.b + b0 * d.b_bs0 + b1 * d.b_bs1;
  const int HtWt = d.Ht * d.Wt;
  const int K = d.K, N = d.N, csplit = d.a.csplit;
  const bf16_t* zero = (const bf16_t*)g_zero_chunk_8;

  // ---- tables: output row of every tile row, the phase's taps
  if (tid < 256) {
    const int m = m0 + tid;
    long long o = -1;
    if (m < d.M) {
      if (L.plain_out) {
        o = m;
      } else {
        int n, y, x;
        if (L.pow2) { n = m >> L.lhw; const int rem = m & (HtWt - 1); y = rem >> L.lw; x = rem & (d.Wt - 1); }
        else { n = m / HtWt; const int rem = m - n * HtWt; y = rem / d.Wt; x = rem - y * d.Wt; }
        const int oy = y * d.ostride + d.ph_oy[phase], ox = x * d.ostride + d.ph_ox[phase];
        if (oy < d.Ho && ox < d.Wo) o = ((long long)n * d.Ho + oy) * d.Wo + ox;
      }
    }
    s_orow[tid] = o;
  } else if (tid < 256 + SDM_MAX_TAPS) {
    const int t = tid - 256;
    s_tap[t] = d.tap_dy[phase][t];
    s_tap[SDM_MAX_TAPS + t] = d.tap_dx[phase][t];
    s_tap[2 * SDM_MAX_TAPS + t] = d.tap_w[phase][t];
  }

  // ---- staging geometry: lane -> (row srow of a 64-row pass, physical 16-byte slot tid & 7 of the 128-byte row)
  const int srow = tid >> 3;
  const int cc8 = 8 * ((tid & 7) ^ ((srow >> 1) & 7));   // logical chunk (elements) behind this lane's physical slot
  int a_y[2][2], a_x[2][2], a_row[2][2];                  // source coordinates of the lane's four A rows: [half][pass]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + h * 128 + i * 64 + srow;
      if (m < d.M) {
        int n, y, x;
        if (L.pow2) { n = m >> L.lhw; const int rem = m & (HtWt - 1); y = rem >> L.lw; x = rem & (d.Wt - 1); }
        else { n = m / HtWt; const int rem = m - n * HtWt; y = rem / d.Wt; x = rem - y * d.Wt; }
        a_y[h][i] = y * d.sstride;
        a_x[h][i] = x * d.sstride;
        a_row[h][i] = (n * d.a.H + a_y[h][i]) * d.a.W + a_x[h][i];
      } else {
        a_y[h][i] = -(1 << 20); a_x[h][i] = 0; a_row[h][i] = 0;
      }
    }
  __syncthreads();

  const int ntap = d.ntaps[phase];
  const int kch = K >> 6;                                 // K-tiles per tap
  const int kch0 = (csplit < K ? csplit : K) >> 6;        // ... served by the first concatenated source
  const int nkt = ntap * kch;
  const int S = 4 * nkt;                                  // stagings

  const bf16_t* pA[2][2];
  int stA[2][2];
  const bf16_t* pB[2][2];
  int tpA[2] = {0, 0}, kcA[2] = {0, 0}, tpB[2] = {0, 0}, kcB[2] = {0, 0};
  // A half h: pointers of the segment (tap tp, from K-tile kc of that tap)
  auto seg_a = [&](auto hc, int tp, int kc) {
    constexpr int h = decltype(hc)::value;
    const int dy = s_tap[tp], dx = s_tap[SDM_MAX_TAPS + tp];
    const bool second = kc >= kch0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int sy = a_y[h][i] + dy, sx = a_x[h][i] + dx;
      const bool ok = (unsigned)sy < (unsigned)d.a.H && (unsigned)sx < (unsigned)d.a.W;
      const long long row = (long long)a_row[h][i] + dy * d.a.W + dx;
      const bf16_t* p = second ? a1 + row * d.a.ld1 + (((kc - kch0) << 6) + cc8) : a0 + row * d.a.ld0 + ((kc << 6) + cc8);
      pA[h][i] = ok ? p : zero;
      stA[h][i] = ok ? 64 : 0;
    }
  };
  // B half h at (tap 0, K-tile 0); rows beyond N read the last valid row (their columns are never stored)
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int n = n0 + h * 128 + i * 64 + srow;
      n = n < N ? n : N - 1;
      pB[h][i] = bw + (long long)(TAPIN ? 0 : s_tap[2 * SDM_MAX_TAPS]) * d.b_tap_stride + (long long)n * d.ldb + cc8;
    }
  // TAPIN state: tap tables in lanes, per-row tap validity masks, row base pointers of both concatenated sources
  int lane_dw = 0, lane_tw = 0;
  unsigned mskA[2][2] = {{0u, 0u}, {0u, 0u}};
  const bf16_t* bA0[2][2];
  const bf16_t* bA1[2][2];
  if constexpr (TAPIN) {
    lane_dw = lane < ntap ? s_tap[lane] * d.a.W + s_tap[SDM_MAX_TAPS + lane] : 0;
    lane_tw = lane < ntap ? s_tap[2 * SDM_MAX_TAPS + lane] : 0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        unsigned mk = 0u;
        for (int t = 0; t < ntap; ++t) {
          const int sy = a_y[h][i] + s_tap[t], sx = a_x[h][i] + s_tap[SDM_MAX_TAPS + t];
          if ((unsigned)sy < (unsigned)d.a.H && (unsigned)sx < (unsigned)d.a.W) mk |= 1u << t;
        }
        mskA[h][i] = mk;
        bA0[h][i] = a0 + (long long)a_row[h][i] * d.a.ld0 + cc8;
        bA1[h][i] = a1 ? a1 + (long long)a_row[h][i] * d.a.ld1 + cc8 : zero;
      }
    tpA[0] = tpA[1] = tpB[0] = tpB[1] = 0;
    kcA[0] = kcA[1] = kcB[0] = kcB[1] = 0;
  } else {
    seg_a(IC8(0), 0, 0);
    seg_a(IC8(1), 0, 0);
  }

  char* lds_w = smem8 + wave * 1024;
  // staging s = 4 T + j : j = 0 B half 0, 1 A half 0, 2 B half 1, 3 A half 1 ; slot = s & 7
  auto stage = [&](auto jc, auto slotc) {
    constexpr int j = decltype(jc)::value, slot = decltype(slotc)::value, h = j >> 1;
    char* dst = lds_w + slot * HT8;
    if constexpr (TAPIN) {
      if constexpr ((j & 1) == 0) {
        const long long off = (long long)__builtin_amdgcn_readlane(lane_tw, tpB[h]) * d.b_tap_stride + (kcB[h] << 6);
        __builtin_amdgcn_global_load_lds((glb_ptr)(pB[h][0] + off), (lds_ptr)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr)(pB[h][1] + off), (lds_ptr)(dst + 8192), 16, 0, 0);
        if (++tpB[h] == ntap) { tpB[h] = 0; ++kcB[h]; }
      } else {
        const int tp = tpA[h], kc = kcA[h];
        const long long dw = __builtin_amdgcn_readlane(lane_dw, tp);
        const bool second = kc >= kch0;
        const long long off = second ? dw * d.a.ld1 + ((kc - kch0) << 6) : dw * d.a.ld0 + (kc << 6);
        const bf16_t* q0 = ((mskA[h][0] >> tp) & 1u) ? (second ? bA1[h][0] : bA0[h][0]) + off : zero;
        const bf16_t* q1 = ((mskA[h][1] >> tp) & 1u) ? (second ? bA1[h][1] : bA0[h][1]) + off : zero;
        __builtin_amdgcn_global_load_lds((glb_ptr)q0, (lds_ptr)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr)q1, (lds_ptr)(dst + 8192), 16, 0, 0);
        if (++tpA[h] == ntap) { tpA[h] = 0; ++kcA[h]; }
      }
    } else
    if constexpr ((j & 1) == 0) {
      __builtin_amdgcn_global_load_lds((glb_ptr)pB[h][0], (lds_ptr)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr)pB[h][1], (lds_ptr)(dst + 8192), 16, 0, 0);
      long long step = 64;
      if (++kcB[h] == kch) {                               // next tap's weight slice
        kcB[h] = 0;
        const int t = tpB[h]++;
        if (t + 1 < ntap) step = (long long)(s_tap[2 * SDM_MAX_TAPS + t + 1] - s_tap[2 * SDM_MAX_TAPS + t]) * d.b_tap_stride - (K - 64);
      }
      pB[h][0] += step;
      pB[h][1] += step;
    } else {
      __builtin_amdgcn_global_load_lds((glb_ptr)pA[h][0], (lds_ptr)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr)pA[h][1], (lds_ptr)(dst + 8192), 16, 0, 0);
      if (++kcA[h] == kch) {
        kcA[h] = 0;
        if (++tpA[h] < ntap) seg_a(IC8(h), tpA[h], 0);
      } else if (kcA[h] == kch0) {
        seg_a(IC8(h), tpA[h], kcA[h]);                     // the second concatenated source starts
      } else {
        pA[h][0] += stA[h][0];
        pA[h][1] += stA[h][1];
      }
    }
  };
  auto wait_younger = [&](int n) {   // at most n stagings (2 DMA instructions each) still in flight
    if (n >= 5) wait_vm8<10>();
    else if (n == 4) wait_vm8<8>();
    else if (n == 3) wait_vm8<6>();
    else if (n == 2) wait_vm8<4>();
    else if (n == 1) wait_vm8<2>();
    else wait_vm8<0>();
  };

  // ---- fragment reads: lane l takes row (l & 15) of a 16-row block and 16-byte chunk (l >> 4) of a 32-wide k-step
  const int l15 = lane & 15, lq = lane >> 4, sw = (l15 >> 1) & 7;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem8;
  const unsigned oa0 = (unsigned)((wr * 64 + l15) * 128 + ((lq ^ sw) << 4));   // k-step 0; k-step 1: offset ^ 64
  const unsigned ob0 = (unsigned)((wc * 32 + l15) * 128 + ((lq ^ sw) << 4));
  const unsigned ra[2][2] = {{lds0 + oa0, lds0 + (oa0 ^ 64u)}, {lds0 + oa0 + 65536u, lds0 + (oa0 ^ 64u) + 65536u}};   // [upper 64 KB][k-step]
  const unsigned rb[2][2] = {{lds0 + ob0, lds0 + (ob0 ^ 64u)}, {lds0 + ob0 + 65536u, lds0 + (ob0 ^ 64u) + 65536u}};

  f32x4 fa[2][4];               // A fragments [k-step][16-row block]
  f32x4 fx[2][2], fy[2][2];     // the two B fragment sets [k-step][16-column block]
  f32x4 acc[2][2][4][2];        // [qa][qb][row block][column block]: lane = row (l & 15), columns (l >> 4) * 4 .. + 3
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[i][j][r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto read_a = [&](auto slotc) {
    constexpr int slot = decltype(slotc)::value, up = slot >> 2, off = (slot & 3) * HT8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      dsr8<off>(fa[ks][0], ra[up][ks]);
      dsr8<off + 2048>(fa[ks][1], ra[up][ks]);
      dsr8<off + 4096>(fa[ks][2], ra[up][ks]);
      dsr8<off + 6144>(fa[ks][3], ra[up][ks]);
    }
  };
  auto read_b = [&](auto slotc, f32x4 (&fb)[2][2]) {
    constexpr int slot = decltype(slotc)::value, up = slot >> 2, off = (slot & 3) * HT8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      dsr8<off>(fb[ks][0], rb[up][ks]);
      dsr8<off + 2048>(fb[ks][1], rb[up][ks]);
    }
  };
  auto mfma_q = [&](f32x4 (&q)[4][2], f32x4 (&fb)[2][2]) {
    // the fragments were written by asm reads: opaque here, so that no MFMA is scheduled above the wait in front
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(fa[ks][r]));
#pragma unroll
      for (int c = 0; c < 2; ++c) asm volatile("" : "+v"(fb[ks][c]));
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          q[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ks][c]),
                                                            __builtin_bit_cast(bf16x8, fa[ks][r]), q[r][c], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: stagings 0 .. 6, wait for 0 and 1, B0 of tile 0 into fx, then the offset of the wave halves
  int issued = 0;
  if (0 < S) { stage(IC8(0), IC8(0)); ++issued; }
  if (1 < S) { stage(IC8(1), IC8(1)); ++issued; }
  if (2 < S) { stage(IC8(2), IC8(2)); ++issued; }
  if (3 < S) { stage(IC8(3), IC8(3)); ++issued; }
  if (4 < S) { stage(IC8(0), IC8(4)); ++issued; }
  if (5 < S) { stage(IC8(1), IC8(5)); ++issued; }
  if (6 < S) { stage(IC8(2), IC8(6)); ++issued; }
  T8_STAMP(1);
  wait_younger(issued - 2);
  __builtin_amdgcn_s_barrier();
  T8_STAMP(2);
  read_b(IC8(0), fx);
  if (wr == 1) __builtin_amdgcn_s_barrier();

  // phase P of iteration I is phase f = 8 I + P + 1 of the walk
  auto phase_fn = [&](auto Pc, int f) {
    constexpr int P = decltype(Pc)::value;
    constexpr int j = (P + 1) & 3, rslot = (P + 1) & 7, sslot = (P + 7) & 7, sj = (P + 3) & 3;
    constexpr bool odd = P >= 4;            // tile parity: B0 of the current tile sits in fy (odd) or fx (even)
    // L segment
    if constexpr (j == 1 || j == 3) read_a(IC8(rslot));
    else if constexpr (j == 2) { if constexpr (odd) read_b(IC8(rslot), fx); else read_b(IC8(rslot), fy); }
    else { if (f < S) { if constexpr (odd) read_b(IC8(rslot), fx); else read_b(IC8(rslot), fy); } }
    if (f + 6 < S) stage(IC8(sj), IC8(sslot));
    wait_younger(S - f - 2);
    __builtin_amdgcn_s_barrier();
    // M segment
    wait_lgkm8();
    if constexpr (j == 1) { if constexpr (odd) mfma_q(acc[0][0], fy); else mfma_q(acc[0][0], fx); }
    else if constexpr (j == 2) { if constexpr (odd) mfma_q(acc[0][1], fx); else mfma_q(acc[0][1], fy); }
    else if constexpr (j == 3) { if constexpr (odd) mfma_q(acc[1][1], fx); else mfma_q(acc[1][1], fy); }
    else { if constexpr (odd) mfma_q(acc[1][0], fy); else mfma_q(acc[1][0], fx); }
    __builtin_amdgcn_s_barrier();
  };
  const int niter = (nkt + 1) >> 1;
  for (int I = 0; I < niter; ++I) {
    const int f = 8 * I + 1;
    phase_fn(IC8(0), f);
    phase_fn(IC8(1), f + 1);
    phase_fn(IC8(2), f + 2);
    phase_fn(IC8(3), f + 3);
    if (2 * I + 1 < nkt) {
      phase_fn(IC8(4), f + 4);
      phase_fn(IC8(5), f + 5);
      phase_fn(IC8(6), f + 6);
      phase_fn(IC8(7), f + 7);
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();     // every wave has passed the same number of barriers; the ring is free
  T8_STAMP(3);

  // ---- epilogue.  acc[qa][qb][r][c][e] is output (tile row qa*128 + wr*64 + r*16 + l15, tile column qb*128 + wc*32 + c*16
  // + lq*4 + e): four consecutive columns of one row per lane.
  const long long obo = b0 * d.o_bs0 + b1 * d.o_bs1;
  const long long rbo = b0 * d.r_bs0 + b1 * d.r_bs1;
  bf16_t* out16 = (bf16_t*)d.out;
  bf16_t* pre16 = (bf16_t*)d.out_preact;
  const bf16_t* res16 = (const bf16_t*)d.res;
  const bf16_t* res216 = (const bf16_t*)d.res2;
  const bf16_t* dz16 = (const bf16_t*)d.dact_z;
  float* red = (float*)smem8;                   // small cross-wave reductions (the ring is free)

  if (d.col_stats) {
    // pass 1 of the two-pass column softmax: {max, sum exp} of alpha * acc over the tile's rows, per column; nothing else
    // is stored.  A lane holds 8 rows of each of its 16 columns; lanes l15 = 0..15 hold 16 more each; the two wave rows
    // meet in LDS.
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int qa = 0; qa < 2; ++qa)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool live = s_orow[qa * 128 + wr * 64 + r * 16 + l15] >= 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = acc[qa][qb][r][c][e] * d.alpha;
              asm volatile("" : "+v"(v));        // rounded product: never contracted into the subtraction below
              v = live ? v : -INFINITY;
              acc[qa][qb][r][c][e] = v;
              mx[e] = fmaxf(mx[e], v);
            }
          }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
          for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], __shfl_xor(mx[e], o, 64));
        f32x4 sm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (mx[e] > -INFINITY) {
#pragma unroll
            for (int qa = 0; qa < 2; ++qa)
#pragma unroll
              for (int r = 0; r < 4; ++r) sm[e] += __expf(acc[qa][qb][r][c][e] - mx[e]);
          }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
          for (int e = 0; e < 4; ++e) sm[e] += __shfl_xor(sm[e], o, 64);
        if (l15 == 0) {
          const int cl = qb * 128 + wc * 32 + c * 16 + lq * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            red[(wr * 256 + cl + e) * 2] = mx[e];
            red[(wr * 256 + cl + e) * 2 + 1] = sm[e];
          }
        }
      }
    __syncthreads();
    if (tid < 256) {
      const int col = n0 + tid;
      if (col < N) {
        float mx = red[tid * 2], sm = red[tid * 2 + 1];
        const float m2 = red[(256 + tid) * 2], s2 = red[(256 + tid) * 2 + 1];
        const float mn = fmaxf(mx, m2);
        sm = (mx > -INFINITY ? sm * __expf(mx - mn) : 0.f) + (m2 > -INFINITY ? s2 * __expf(m2 - mn) : 0.f);
        float* o = d.col_stats + (long long)blockIdx.z * d.col_stats_bs + ((long long)tile_m * N + col) * 2;
        o[0] = mn;
        o[1] = sm;
      }
    }
    return;
  }

  // Main epilogue, staged through LDS in two slabs of 128 tile rows (qa = 0, 1; 128 x 256 fp32 = the ring): the
  // accumulator layout would allow direct 8-byte stores, but 32 copies of the epilogue's branches (one per accumulator
  // register group) are ~90 KB of straight-line code -- the instruction fetch of that cost more than the whole K loop's
  // gain (measured: 956 vs 1240 TFLOP/s for the bare loop).  Staged, the epilogue is ONE rolled loop: a thread owns one
  // column quad (tid & 63) for all its rows, bias / column vectors are loaded once, a wave's store covers 512 contiguous
  // bytes of one output row.  LDS image: row r (0..127) at r * 1024 bytes, 16-byte column quad q at slot q ^ (r & 15) (the
  // 16 lanes of a ds_write_b128 group hold 16 different rows of one quad: conflict-free; a reading wave takes one row).
  const int cq = tid & 63;                     // this thread's column quad in the epilogue loop
  const int mycol = n0 + cq * 4;
  const bool col_ok = mycol < N;               // N % 4 == 0 (vector epilogue)
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
  const f32x4 bias4 = (d.bias && col_ok) ? *(const f32x4*)(d.bias + mycol) : zero4;
  const f32x4 cshift4 = (d.col_shift && col_ok) ? *(const f32x4*)(d.col_shift + (long long)blockIdx.z * d.colvec_bs + mycol) : zero4;
  const f32x4 cmul4 = (d.col_mul && col_ok) ? *(const f32x4*)(d.col_mul + (long long)blockIdx.z * d.colvec_bs + mycol) : one4;
  f32x4 gs = zero4, gq = zero4;                // GroupNorm statistics of this thread's 4 columns (over its rows)
  char* Ct = smem8;
  const bool simple = !d.col_shift && !L.out_f32 && d.alpha == 1.0f &&
                      (d.act == SDM_ACT_NONE || d.act == SDM_ACT_SILU) && !L.no_simple_epi;
  const bool extras = res16 || res216 || dz16;                      // (uniform: one scalar branch per trip when absent)
#pragma unroll
  for (int qa = 0; qa < 2; ++qa) {
    if (qa) __syncthreads();                   // slab 0 has been read
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int row = wr * 64 + r * 16 + l15;                        // row & 15 == l15
          const int quad = qb * 32 + wc * 8 + c * 4 + lq;
          *(f32x4*)(Ct + row * 1024 + ((quad ^ l15) << 4)) = acc[qa][qb][r][c];
        }
    __syncthreads();
    if (simple) {
      // the forward conv's and the stride-1 data gradient's epilogue (bias, SiLU or nothing, bf16 rows in tile order, optional
      // pre-activation copy, GroupNorm sums, residuals and the silu'(z) factor) as a straight loop: output pointers advance by 8 rows per trip, no table lookups, no per-option
      // address arithmetic -- the general loop below is ~200 instructions per trip and took 17-22 us per tile
      // (profiles/r04_igemm8_timing.txt), 12-15 % of a 512-channel 3x3 tile and over half of a K = 512 1x1 tile.
      // (output row of a tile row: the row itself for plain geometry, else the table entry -- sub-pixel phases, strided outputs)
      const int mrow = m0 + qa * 128 + wave;
      const long long ooff = obo + mycol;
      // (data-gradient extras -- + residual, + second residual, x silu'(z), in the general loop's order -- take their addresses
      // per trip: pointers kept across the loop spilled registers)
      const bool zcol = d.dact_ncols == 0 || mycol < d.dact_ncols;
      const char* cp = Ct + wave * 1024;
      auto trip = [&](auto silu_c, int it) {
        constexpr bool SILU = decltype(silu_c)::value;
        const int row = it * 8 + wave;
        f32x4 v = *(const f32x4*)(cp + it * 8192 + ((cq ^ (row & 15)) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bias4[e];
        const long long r = L.plain_out ? (long long)(mrow + it * 8) : s_orow[qa * 128 + row];
        if ((L.plain_out ? mrow + it * 8 < d.M : r >= 0) && col_ok) {
          const long long oo = ooff + r * d.ldo;
          if (pre16) {
            u16x4 zo;
#pragma unroll
            for (int e = 0; e < 4; ++e) zo[e] = f2bf(v[e]);
            *(u16x4*)(pre16 + oo) = zo;
          }
          if constexpr (SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = sdm_silu(v[e]);
          }
          if (extras) {
            if (res16) {
              const u16x4 rv = *(const u16x4*)(res16 + rbo + r * d.ldres + mycol);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += bf2f(rv[e]);
            }
            if (res216) {
              const u16x4 rv = *(const u16x4*)(res216 + r * d.ldres2 + mycol);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += bf2f(rv[e]);
            }
            if (dz16 && zcol) {
              const u16x4 zv = *(const u16x4*)(dz16 + r * d.ldz + mycol);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= sdm_dsilu(bf2f(zv[e]));
            }
          }
          u16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
          *(u16x4*)(out16 + oo) = o;
          if (L.gn_stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float rr = bf2f(o[e]);
              gs[e] += rr;
              gq[e] = fmaf(rr, rr, gq[e]);
            }
          }
        }
      };
      if (d.act == SDM_ACT_SILU) {
#pragma unroll 4
        for (int it = 0; it < 16; ++it) trip(std::true_type(), it);
      } else {
#pragma unroll 4
        for (int it = 0; it < 16; ++it) trip(std::false_type(), it);
      }
      continue;
    }
#pragma unroll 2
    for (int it = 0; it < 16; ++it) {
      const int row = it * 8 + wave;            // tile row inside the slab: one row per wave and iteration
      const long long orow = s_orow[qa * 128 + row];
      if (orow < 0 || !col_ok) continue;
      f32x4 v = *(const f32x4*)(Ct + row * 1024 + ((cq ^ (row & 15)) << 4));
      if (d.col_shift) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = v[e] * d.alpha;
          asm volatile("" : "+v"(t));      // the rounded product pass 1 took its maxima from (no fma contraction)
          v[e] = __expf(t + cshift4[e]) * cmul4[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * d.alpha + bias4[e];
      }
      if (pre16) {     // pre-activation copy for the backward (training)
        u16x4 zo;
#pragma unroll
        for (int e = 0; e < 4; ++e) zo[e] = f2bf(v[e]);
        *(u16x4*)(pre16 + obo + orow * d.ldo + mycol) = zo;
      }
      if (!d.col_shift) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = sdm_act(v[e], d.act);
      }
      if (res16) {
        const u16x4 rv = *(const u16x4*)(res16 + rbo + orow * d.ldres + mycol);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bf2f(rv[e]);
      }
      if (res216) {
        const u16x4 rv = *(const u16x4*)(res216 + orow * d.ldres2 + mycol);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bf2f(rv[e]);
      }
      if (dz16 && (d.dact_ncols == 0 || mycol < d.dact_ncols)) {   // data gradient: times silu'(z) of the producer
        const u16x4 zv = *(const u16x4*)(dz16 + orow * d.ldz + mycol);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= sdm_dsilu(bf2f(zv[e]));
      }
      if (L.out_f32) {
        *(f32x4*)(d.out + obo + orow * d.ldo + mycol) = v;
      } else {
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        *(u16x4*)(out16 + obo + orow * d.ldo + mycol) = o;
        if (L.gn_stats) {       // statistics of the values the next layer will read (after rounding to bf16)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float rr = bf2f(o[e]);
            gs[e] += rr;
            gq[e] = fmaf(rr, rr, gq[e]);
          }
        }
      }
    }
  }
  if (L.gn_stats) {
    // the 8 waves hold partial sums of the same 256 columns (rows it * 8 + wave): fold them through LDS (the staged tile is
    // dead), then one fp32 atomic pair per column and workgroup into [image][channel][2].  The host guarantees that the
    // tile's rows lie in ONE image and that output rows are tile rows (plain 3x3 stride-1 conv).
    __syncthreads();
    *(f32x4*)(red + (tid * 2) * 4) = gs;
    *(f32x4*)(red + (tid * 2 + 1) * 4) = gq;
    __syncthreads();
    if (tid < 256 && n0 + tid < N && s_orow[0] >= 0) {
      const int q = tid >> 2, e = tid & 3;
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        a += red[((w * 64 + q) * 2) * 4 + e];
        b += red[((w * 64 + q) * 2 + 1) * 4 + e];
      }
      const long long img = s_orow[0] / ((long long)d.Ho * d.Wo);
      float* dst = d.gn_sums + (img * N + n0 + tid) * 2;
      atomicAdd(dst, a);
      atomicAdd(dst + 1, b);
    }
  }
  T8_STAMP(4);
}

int sdm_launch_igemm8_bf16(const BfLaunch& L0, int ntm, hipStream_t st) {
  constexpr size_t lds = (size_t)RING8 + 256 * sizeof(long long) + 3 * SDM_MAX_TAPS * sizeof(int);
  BfLaunch L = L0;
  L.kchunks = L.d.K / 64;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)igemm8_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)igemm8_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  static const char* se_s = getenv("SDM_IGEMM8_SIMPLE_EPI");  // "0": the general epilogue loop everywhere (A/B)
  L.no_simple_epi = (se_s && se_s[0] == '0') ? 1 : 0;
  static const char* tapin_s = getenv("SDM_BF16_TAPIN");     // "1": taps innermost in the K walk (see the kernel)
  const bool tapin = tapin_s && tapin_s[0] == '1';
  dim3 grid(ntm * L.ntn, L.d.nphase, L.d.batch);
  double taps = 0;
  for (int p = 0; p < L.d.nphase; ++p) taps += L.d.ntaps[p];
  {
    const double images = (double)L.d.M / ((double)L.d.Ht * L.d.Wt);
    const double in_b = images * L.d.a.H * L.d.a.W * L.d.K, w_b = (double)L.d.K * L.d.N * taps, out_b = images * L.d.Ho * L.d.Wo * L.d.N;
    sdm_prof_shape(L.d.M, L.d.N, L.d.K, (int)taps, 1, 2.0 * L.d.batch * (in_b + w_b + out_b));
  }
  sdm_prof_pre(st);
  if (tapin) SDM_LAUNCH(igemm8_bf16_kernel<true>, grid, dim3(512), lds, st, L);
  else SDM_LAUNCH(igemm8_bf16_kernel<false>, grid, dim3(512), lds, st, L);
  sdm_prof_post(st, 0, 2.0 * L.d.M * L.d.N * L.d.K * taps * L.d.batch);
  SDM_CHECK_LAUNCH("igemm8_bf16");
  return SDM_OK;
}
