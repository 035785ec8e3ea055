// bf16 storage path of the U-Net forward (sampling): activations and weights are bf16 in HBM, every sum is fp32
// (matrix-core accumulators, GroupNorm statistics, softmax), biases / modulation tables / sampler state stay fp32.
//
//   v_mfma_f32_32x32x16_bf16: lane l (r = l&31, h = l>>5) supplies A[row r][k = 8h..8h+7] and B[k = 8h..8h+7][col r]
//   -- one 16-byte chunk of a K-contiguous row each -- and the 32x32 fp32 result has col = lane&31,
//   row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), exactly the layout of the fp32 path's 32x32x2 instruction.
//
// A K-step is 64 bf16 = 128 bytes per tile row, so the LDS image (128-byte rows, 16-byte chunks XOR-swizzled by
// (row>>1)&7, filled by global_load_lds_dwordx4) is byte-for-byte the one of the fp32 kernel in sdm_igemm.hip; what
// changes is the arithmetic density: a K-step is 4 MFMAs (128 cycles) per 32x32 tile instead of 16 (1024 cycles),
// so operand movement, not the matrix pipe, sets the pace (see the tile configurations below).
#include <stdlib.h>

#define SDM_FAST_SIGMOID
#include "sdm_common.h"

#include "sdm_bf16_int.h"

#define BKE 64   // reduction elements per K-step (128 bytes)
#define ROWF 32  // floats per LDS tile row

static inline int ilog2_exact_bf(long v) {
  if (v <= 0 || (v & (v - 1)) != 0) return -1;
  int l = 0;
  while ((1L << l) < v) ++l;
  return l;
}

__device__ float g_zero_chunk_bf[4];

// timing build only (-DSDM_IGEMM8_TIMING, see sdm_igemm8.hip): phase stamps of conv3_row_bf16_kernel's workgroups
#ifdef SDM_IGEMM8_TIMING
#define TR_MAXWG 65536
__device__ unsigned long long g_tr[TR_MAXWG][4];
#define TR_STAMP(i)                                                                                          \
  do {                                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < TR_MAXWG) g_tr[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime();   \
  } while (0)
extern "C" int sdm_debug_row_times(unsigned long long* host, int nwg) {
  if (nwg > TR_MAXWG) nwg = TR_MAXWG;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tr), (size_t)nwg * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#else
#define TR_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ int xcd_contiguous_bf(int bid, int total) {   // see sdm_igemm.hip
  const int q = total >> 3, r = total & 7, x = bid & 7, i = bid >> 3;
  return x < r ? x * (q + 1) + i : r * (q + 1) + (x - r) * q + i;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_bf() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Fragment reads the compiler does not see (ds_read_b128 in inline asm) with hand-counted waits.  Left to the compiler,
// the wait in front of an MFMA group is s_waitcnt lgkmcnt(0) -- it also waits for the NEXT group's fragments, issued just
// before (its scoreboard cannot count across the DMA-issuing branch of the loop) -- so every other group paid a full LDS
// round trip.  LDS reads return in order: "at most N outstanding" with N = the reads issued after the set needed is exact.
template <int OFF>
__device__ __forceinline__ void ds_read128_bf(f32x4& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkmcnt_bf() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// out[m][n] = act(alpha * sum_tap sum_k A[row(m,tap)][k] * B[tap][n][k] + bias[n]) (+ res) ; see sdm_igemm_desc.
// 256 threads = 2x2 waves, each wave owns (BM/2)x(BN/2).
// FAST: K and the concat split are multiples of the 64-element K-step, so a K-step never straddles the reduction end or
// the two concatenated sources; the DMA addresses are then per-lane pointers advanced by one add per K-step (rows outside
// the image point at a zero chunk with step 0) and rebuilt only when the (tap, source) segment changes.
template <int BM, int BN, int NT>
__device__ __forceinline__ void splitk_finish_tile_bf(const BfLaunch& L, const long long* s_orow, int n0) {
  const sdm_igemm_desc& d = L.d;
  const int N = d.N;
  const int b0 = blockIdx.z / d.bdiv, b1 = blockIdx.z % d.bdiv;
  const long long obo = b0 * d.o_bs0 + b1 * d.o_bs1, rbo = b0 * d.r_bs0 + b1 * d.r_bs1;
  float* wsb = d.ws + (long long)blockIdx.z * L.out_rows * N;
  bf16_t* out16 = (bf16_t*)d.out;
  const bf16_t* res16 = (const bf16_t*)d.res;
  const bf16_t* res216 = (const bf16_t*)d.res2;
  for (int e = threadIdx.x; e < BM * BN; e += NT) {
    const int r = e / BN, col = n0 + (e % BN);
    const long long orow = s_orow[r];
    if (orow < 0 || col >= N) continue;
    float* w = wsb + orow * N + col;
    float v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *w = 0.f;
    const long long oo = obo + orow * d.ldo + col;
    v = v * d.alpha + (d.bias ? d.bias[col] : 0.f);
    if (d.out_preact) ((bf16_t*)d.out_preact)[oo] = f2bf(v);
    v = sdm_act(v, d.act);
    if (res16) v += bf2f(res16[rbo + orow * d.ldres + col]);
    if (res216) v += bf2f(res216[orow * d.ldres2 + col]);
    if (d.dact_z && (d.dact_ncols == 0 || col < d.dact_ncols)) v *= sdm_dsilu(bf2f(((const bf16_t*)d.dact_z)[orow * d.ldz + col]));
    if (L.out_f32) d.out[oo] = v;
    else out16[oo] = f2bf(v);
  }
}

template <int BM, int BN, int WM, int WN, int NBUF, bool FAST, bool XS, int KE = 64, bool PP = false>
__global__ __launch_bounds__(64 * WM * WN) void igemm_bf16_kernel(const BfLaunch L) {
  const sdm_igemm_desc& d = L.d;
  constexpr int NT = 64 * WM * WN;              // threads: WM x WN waves, each owning (BM/WM) x (BN/WN)
  constexpr int RF = KE / 2;                    // floats per LDS tile row (a K-step is KE bf16 = 128 or 64 bytes per row)
  constexpr int CPR = KE / 8;                   // 16-byte chunks per row
  constexpr int KG = KE / 16;                   // MFMA k-groups per K-step
  static_assert(KE == 64 || KE == 32, "K-step");
  constexpr int RA = NT / CPR;                  // tile rows covered by one pass of the NT threads
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr int AIT = BM / RA, BIT = BN / RA;
  static_assert(RA % 16 == 0 && BM % RA == 0 && BN % RA == 0, "pass geometry");
  constexpr int A_TILE = BM * RF, B_TILE = BN * RF;
  constexpr int G = AIT + BIT;
  constexpr int PD = NBUF - 1;
  static_assert(PD >= 1 && PD <= 5, "ring depth");
  extern __shared__ __align__(16) float smem[];
  float* As = smem;
  float* Bs = smem + NBUF * A_TILE;
  long long* s_orow = (long long*)(Bs + NBUF * B_TILE);
  int* s_tap = (int*)(s_orow + BM);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, h = lane >> 5, l31 = lane & 31;
  const int tile_id = xcd_contiguous_bf(blockIdx.x, gridDim.x), ntm_ = gridDim.x / L.ntn;
  const int tile_n = L.mfast ? tile_id / ntm_ : tile_id % L.ntn, tile_m = L.mfast ? tile_id % ntm_ : tile_id / L.ntn;
  const int phase = blockIdx.y / L.splitk, ks = blockIdx.y % L.splitk;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int b0 = blockIdx.z / d.bdiv, b1 = blockIdx.z % d.bdiv;
  const bf16_t* a0 = (const bf16_t*)d.a.p0 + b0 * d.a_bs0 + b1 * d.a_bs1;
  const bf16_t* a1 = (const bf16_t*)d.a.p1;
  const bf16_t* bw = (const bf16_t*)d.b + b0 * d.b_bs0 + b1 * d.b_bs1;
  const int HtWt = d.Ht * d.Wt;
  const int K = d.K, N = d.N, csplit = d.a.csplit;
  // XOR swizzle of the 16-byte chunks inside a row, so that the 16 lanes of a ds_read_b128 group fall on 16 different
  // slots of the 256-byte bank row: 128-byte rows (two per bank row): chunk ^ ((row >> 1) & 7); 64-byte rows (four per bank
  // row): chunk ^ ((row >> 2) & 3)
  const int arow = tid / CPR;
  const int cc8 = KE == 64 ? 8 * ((tid & 7) ^ ((arow >> 1) & 7)) : 8 * ((tid & 3) ^ ((arow >> 2) & 3));   // logical chunk (in elements) behind this lane's physical slot
  const int sw_l = KE == 64 ? (l31 >> 1) & 7 : (l31 >> 2) & 3;
  const bf16_t* zero = (const bf16_t*)g_zero_chunk_bf;

  for (int r = tid; r < BM; r += NT) {
    const int m = m0 + r;
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
    s_orow[r] = o;
  }
  if (tid < SDM_MAX_TAPS) {
    s_tap[tid] = d.tap_dy[phase][tid];
    s_tap[SDM_MAX_TAPS + tid] = d.tap_dx[phase][tid];
    s_tap[2 * SDM_MAX_TAPS + tid] = d.tap_w[phase][tid];
  }
  int a_ys[AIT], a_xs[AIT];
  long long a_base[AIT];
#pragma unroll
  for (int i = 0; i < AIT; ++i) {
    const int m = m0 + arow + RA * i;
    if (m < d.M) {
      int n, y, x;
      if (L.pow2) { n = m >> L.lhw; const int rem = m & (HtWt - 1); y = rem >> L.lw; x = rem & (d.Wt - 1); }
      else { n = m / HtWt; const int rem = m - n * HtWt; y = rem / d.Wt; x = rem - y * d.Wt; }
      a_ys[i] = y * d.sstride;
      a_xs[i] = x * d.sstride;
      a_base[i] = ((long long)n * d.a.H + a_ys[i]) * d.a.W + a_xs[i];
    } else {
      a_ys[i] = -(1 << 20); a_xs[i] = 0; a_base[i] = 0;
    }
  }
  __syncthreads();

  const int ntap = d.ntaps[phase];
  const int kch = L.kchunks;
  const int T = ntap * kch;
  const int it0 = (int)((long long)T * ks / L.splitk), it1 = (int)((long long)T * (ks + 1) / L.splitk);

  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  const bf16_t* pa0[AIT];
  const bf16_t* pa1[AIT];
  bool aok[AIT];
  const bf16_t* pb[BIT];
  bool bok[BIT];
  int astep[AIT], bstep[BIT];          // FAST: elements to advance per K-step (64, or 0 for the zero chunk)
  const int kch0 = (csplit < K ? csplit : K) / KE;   // FAST: K-steps served by the first source
  // general path: per-tap pointers, every K-step re-derives source / validity / tail
  auto setup_tap = [&](int tp) {
    const int dy = s_tap[tp], dx = s_tap[SDM_MAX_TAPS + tp];
    const bf16_t* bt = bw + (long long)s_tap[2 * SDM_MAX_TAPS + tp] * d.b_tap_stride;
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const int sy = a_ys[i] + dy, sx = a_xs[i] + dx;
      aok[i] = (unsigned)sy < (unsigned)d.a.H && (unsigned)sx < (unsigned)d.a.W;
      const long long row = a_base[i] + dy * d.a.W + dx;
      pa0[i] = a0 + row * d.a.ld0 + cc8;
      pa1[i] = a1 + row * d.a.ld1 + (cc8 - csplit);
    }
#pragma unroll
    for (int i = 0; i < BIT; ++i) {
      const int n = n0 + arow + RA * i;
      bok[i] = n < N;
      pb[i] = bt + (long long)n * d.ldb + cc8;
    }
  };
  // FAST path: pointers of segment (tap tp, starting at K-step kc of that tap)
  auto setup_seg = [&](int tp, int kc) {
    const int dy = s_tap[tp], dx = s_tap[SDM_MAX_TAPS + tp];
    const bf16_t* bt = bw + (long long)s_tap[2 * SDM_MAX_TAPS + tp] * d.b_tap_stride + kc * KE + cc8;
    const bool second = kc >= kch0;
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const int sy = a_ys[i] + dy, sx = a_xs[i] + dx;
      const bool ok = (unsigned)sy < (unsigned)d.a.H && (unsigned)sx < (unsigned)d.a.W;
      const long long row = a_base[i] + dy * d.a.W + dx;
      const bf16_t* p = second ? a1 + row * d.a.ld1 + ((kc - kch0) * KE + cc8) : a0 + row * d.a.ld0 + (kc * KE + cc8);
      pa0[i] = ok ? p : zero;
      astep[i] = ok ? KE : 0;
    }
#pragma unroll
    for (int i = 0; i < BIT; ++i) {
      const int n = n0 + arow + RA * i;
      const bool ok = n < N;
      pb[i] = ok ? bt + (long long)n * d.ldb : zero;
      bstep[i] = ok ? KE : 0;
    }
  };

  auto issue = [&](int c0, int buf) {
    float* Ad = As + buf * A_TILE + __builtin_amdgcn_readfirstlane(wave) * 256;
    float* Bd = Bs + buf * B_TILE + __builtin_amdgcn_readfirstlane(wave) * 256;
    if (FAST) {
#pragma unroll
      for (int i = 0; i < AIT; ++i) {
        __builtin_amdgcn_global_load_lds((glb_ptr)pa0[i], (lds_ptr)(Ad + i * RA * RF), 16, 0, 0);
        pa0[i] += astep[i];
      }
#pragma unroll
      for (int i = 0; i < BIT; ++i) {
        __builtin_amdgcn_global_load_lds((glb_ptr)pb[i], (lds_ptr)(Bd + i * RA * RF), 16, 0, 0);
        pb[i] += bstep[i];
      }
      return;
    }
    const int c = c0 + cc8;
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const bool use = aok[i] && c < K;
      const bf16_t* p = (c < csplit ? pa0[i] : pa1[i]) + c0;
      p = use ? p : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr)p, (lds_ptr)(Ad + i * RA * RF), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BIT; ++i) {
      const bool use = bok[i] && c < K;
      const bf16_t* p = use ? pb[i] + c0 : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr)p, (lds_ptr)(Bd + i * RA * RF), 16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int tp = it0 / kch, kc = it0 - tp * kch;
  int issued = it0;
  auto issue_next = [&]() {
    if (issued < it1) {
      if (FAST) {
        if (issued == it0) {
          setup_seg(tp, kc);
        } else if (++kc == kch) {
          kc = 0;
          setup_seg(++tp, 0);
        } else if (kc == kch0) {
          setup_seg(tp, kc);     // second concatenated source starts
        }
      } else {
        if (issued == it0) {
          setup_tap(tp);
        } else if (++kc == kch) {
          kc = 0;
          setup_tap(++tp);
        }
      }
      issue(kc * KE, (issued - it0) % NBUF);
    }
    ++issued;
  };
  auto wait_tile = [&](int younger) {      // wait until at most `younger` K-steps' DMA groups of this lane are still in flight
    if (PD >= 5 && younger >= 5) wait_vmcnt_bf<(PD >= 5 ? 5 : 0) * G>();
    else if (PD >= 4 && younger == 4) wait_vmcnt_bf<(PD >= 4 ? 4 : 0) * G>();
    else if (PD >= 3 && younger == 3) wait_vmcnt_bf<(PD >= 3 ? 3 : 0) * G>();
    else if (PD >= 2 && younger == 2) wait_vmcnt_bf<(PD >= 2 ? 2 : 0) * G>();
    else if (younger == 1) wait_vmcnt_bf<G>();
    else wait_vmcnt_bf<0>();
  };
  // fragment reads run one k-group ahead of the matrix instructions (two register sets, order pinned): left to itself
  // the compiler reads a group, drains the LDS queue and only then issues its MFMAs
  f32x4 af[2][TM], bf[2][TN];
  auto ldfrag = [&](const float* Ab, const float* Bb, int kg, int st) {
    const int ph = ((kg * 2 + h) ^ sw_l) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[st][i] = *(const f32x4*)(Ab + i * 32 * RF + ph);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[st][j] = *(const f32x4*)(Bb + j * 32 * RF + ph);
  };
  auto mfmas = [&](int st) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[st][i]),
                                                            __builtin_bit_cast(bf16x8, bf[st][j]), acc[i][j], 0, 0, 0);
  };
  const int frag_off_a = (wm * TM * 32 + l31) * RF, frag_off_b = (wn * TN * 32 + l31) * RF;

  if (XS) {
    const unsigned lds_a = (unsigned)(uintptr_t)(lds_ptr)(As + frag_off_a), lds_b = (unsigned)(uintptr_t)(lds_ptr)(Bs + frag_off_b);
    auto ldfrag_x = [&](unsigned a_addr, unsigned b_addr, int kg, int st) {
      const unsigned ph = (unsigned)(((kg * 2 + h) ^ sw_l) << 4);
      const unsigned pa = a_addr + ph, pb = b_addr + ph;
      ds_read128_bf<0>(af[st][0], pa);
      if constexpr (TM > 1) ds_read128_bf<32 * RF * 4>(af[st][1], pa);
      if constexpr (TM > 2) ds_read128_bf<2 * 32 * RF * 4>(af[st][2], pa);
      if constexpr (TM > 3) ds_read128_bf<3 * 32 * RF * 4>(af[st][3], pa);
      ds_read128_bf<0>(bf[st][0], pb);
      if constexpr (TN > 1) ds_read128_bf<32 * RF * 4>(bf[st][1], pb);
      if constexpr (TN > 2) ds_read128_bf<2 * 32 * RF * 4>(bf[st][2], pb);
      if constexpr (TN > 3) ds_read128_bf<3 * 32 * RF * 4>(bf[st][3], pb);
    };
    auto pin = [&](int st) {     // the set's registers are "rewritten" here: its MFMAs cannot move above the wait in front
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(af[st][i]));
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bf[st][j]));
    };
    static_assert(TM <= 4 && TN <= 4, "fragment sets");
    static_assert(!PP || (NT == 512 && KG == 4), "ping-pong: 8 waves, four k-groups per K-step");
    if constexpr (PP) {
      // Ping-pong of the two 4-wave halves of an 8-wave tile (waves w and w + 4 share a SIMD).  A K-step is two half-steps of
      // two k-groups: LOAD = read the half-step's 12 fragments into the two register sets, MFMA = its 16 matrix instructions.
      // Both halves meet at ONE barrier per K-step, but at different points of their programs:
      //   half 0:  B | DMA  L0 M0 L1 M1 | B ...
      //   half 1:  B | M1'  DMA L0 M0 L1 | B ...      (M1' = the last half-step of the PREVIOUS tile, read before the barrier)
      // so after the barrier one wave of every SIMD has 16 MFMAs to issue while its partner issues the DMA and reads
      // fragments, and they keep alternating until the next barrier: the matrix pipe is fed across the hand-off.  The barrier
      // still means what it meant: tile `it` has landed for everybody, nobody reads tile it-1 any more (half 1 read its
      // last fragments of it before arriving), so the DMA of tile it+PD may overwrite that slot.
#pragma unroll
      for (int p = 0; p < PD; ++p) issue_next();
      const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
      auto loadh = [&](unsigned a_cur, unsigned b_cur, int half) {
        ldfrag_x(a_cur, b_cur, 2 * half, 0);
        ldfrag_x(a_cur, b_cur, 2 * half + 1, 1);
      };
      auto mfmah = [&]() {
        wait_lgkmcnt_bf<0>();
        pin(0);
        pin(1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
      };
      for (int it = it0; it < it1; ++it) {
        const int buf = (it - it0) % NBUF;
        const unsigned a_cur = lds_a + (unsigned)(buf * A_TILE * 4), b_cur = lds_b + (unsigned)(buf * B_TILE * 4);
        const int younger = (it1 - 1 - it) < (PD - 1) ? (it1 - 1 - it) : (PD - 1);
        wait_tile(younger);
        __builtin_amdgcn_s_barrier();
        if (grp == 1 && it > it0) mfmah();       // M1' of the previous tile
        issue_next();
        loadh(a_cur, b_cur, 0);
        mfmah();
        loadh(a_cur, b_cur, 1);
        if (grp == 0) mfmah();
        else wait_lgkmcnt_bf<0>();               // this wave's reads of tile `it` are complete before it meets the others
      }
      if (grp == 1) mfmah();
    } else {
    // Cross-step schedule.  The per-step barrier sits in the MIDDLE of a K-step, behind three issued MFMA groups (the
      // matrix pipe works through its backlog while the waves meet), and covers two things at once: tile it+1 has landed
      // for every wave, and every wave has finished reading tile it (its last fragment group is read one group ahead).
      // So the first fragment group of tile it+1 is read under the last MFMA group of tile it -- no exposed LDS latency at
      // the step boundary -- and the ring slot of tile it takes the next DMA half a step earlier: all NBUF slots are in
      // flight instead of NBUF-1.
#pragma unroll
      for (int p = 0; p < NBUF; ++p) issue_next();
      {
        const int y0 = (it1 - 1 - it0) < PD ? (it1 - 1 - it0) : PD;
        wait_tile(y0);
        __builtin_amdgcn_s_barrier();
      }
        ldfrag_x(lds_a, lds_b, 0, 0);
      for (int it = it0; it < it1; ++it) {
        const int buf = (it - it0) % NBUF, nbuf = (it + 1 - it0) % NBUF;
        const unsigned a_cur = lds_a + (unsigned)(buf * A_TILE * 4), b_cur = lds_b + (unsigned)(buf * B_TILE * 4);
        const bool more = it + 1 < it1;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
          if (kg < KG - 1) {
            ldfrag_x(a_cur, b_cur, kg + 1, (kg + 1) & 1);
            wait_lgkmcnt_bf<TM + TN>();            // everything older than the reads just issued: set kg & 1 is complete
          } else if (more) {
            // (set kg & 1 was complete at the mid-step wait below)
            ldfrag_x(lds_a + (unsigned)(nbuf * A_TILE * 4), lds_b + (unsigned)(nbuf * B_TILE * 4), 0, 0);
          } else {
            wait_lgkmcnt_bf<0>();
          }
          pin(kg & 1);
          __builtin_amdgcn_sched_barrier(0);
          mfmas(kg & 1);
          __builtin_amdgcn_sched_barrier(0);
          if (kg == KG - 2 && more) {
            const int y = (it1 - 2 - it) < (PD - 1) ? (it1 - 2 - it) : (PD - 1);
            wait_tile(y);
            wait_lgkmcnt_bf<0>();                                  // this wave's reads of tile `it` are complete
            __builtin_amdgcn_s_barrier();
            issue_next();                                          // into the slot of tile `it`
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  } else {
#pragma unroll
  for (int p = 0; p < PD; ++p) issue_next();

  for (int it = it0; it < it1; ++it) {
    const int buf = (it - it0) % NBUF;
    const int younger = (it1 - 1 - it) < (PD - 1) ? (it1 - 1 - it) : (PD - 1);
    wait_tile(younger);
    __builtin_amdgcn_s_barrier();   // tile `it` visible to all waves; ring slot of tile it-1 is free again
    const float* Ab = As + buf * A_TILE + frag_off_a;
    const float* Bb = Bs + buf * B_TILE + frag_off_b;
    ldfrag(Ab, Bb, 0, 0);
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      if (kg < KG - 1) ldfrag(Ab, Bb, kg + 1, (kg + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(kg & 1);
      __builtin_amdgcn_sched_barrier(0);
      // the DMA of K-step it+PD is issued in the shadow of the first MFMA group (the matrix pipe runs them
      // asynchronously), not between the barrier and the first fragment reads
      if (kg == 0) issue_next();
    }
  }
  }

  // ---- epilogue
  const long long obo = b0 * d.o_bs0 + b1 * d.o_bs1;
  const long long rbo = b0 * d.r_bs0 + b1 * d.r_bs1;
  bf16_t* out16 = (bf16_t*)d.out;
  bf16_t* pre16 = (bf16_t*)d.out_preact;
  const bf16_t* res16 = (const bf16_t*)d.res;
  const bf16_t* res216 = (const bf16_t*)d.res2;
  const bf16_t* dz16 = (const bf16_t*)d.dact_z;
  if (d.col_stats) {
    // pass 1 of the two-pass column softmax: {max, sum exp} over this tile's rows for each of its columns; nothing else
    // is stored.  A lane holds ONE column of each of its TN 32x32 blocks (rows: 16 registers x TM blocks, the other half
    // of the rows on lane ^ 32), the WM waves of a tile column meet in LDS.
    __syncthreads();                         // the ring is free
    float* red = smem;                       // [WM][BN][2]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          float v = acc[i][j][r] * d.alpha;
          asm volatile("" : "+v"(v));        // rounded product: never contracted into the subtraction below
          v = s_orow[row] >= 0 ? v : -INFINITY;
          acc[i][j][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sm = 0.f;
      if (mx > -INFINITY) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) sm += __expf(acc[i][j][r] - mx);
      }
      sm += __shfl_xor(sm, 32, 64);
      if (h == 0) {
        const int cl = wn * TN * 32 + j * 32 + l31;
        red[(wm * BN + cl) * 2] = mx;
        red[(wm * BN + cl) * 2 + 1] = sm;
      }
    }
    __syncthreads();
    for (int cl = tid; cl < BN; cl += NT) {
      const int col = n0 + cl;
      if (col >= N) continue;
      float mx = red[cl * 2], sm = red[cl * 2 + 1];
#pragma unroll
      for (int w = 1; w < WM; ++w) {
        const float m2 = red[(w * BN + cl) * 2], s2 = red[(w * BN + cl) * 2 + 1];
        const float mn = fmaxf(mx, m2);
        sm = (mx > -INFINITY ? sm * __expf(mx - mn) : 0.f) + (m2 > -INFINITY ? s2 * __expf(m2 - mn) : 0.f);
        mx = mn;
      }
      float* o = d.col_stats + (long long)blockIdx.z * d.col_stats_bs + ((long long)tile_m * N + col) * 2;
      o[0] = mx;
      o[1] = sm;
    }
    return;
  }
  if (L.splitk == 1 && L.vec_epi) {
    // Staged through LDS: the accumulator layout gives a lane ONE column and 16 scattered rows of a 32x32 tile, i.e.
    // 2-byte stores; the ring (no longer needed) is reused as an fp32 [BM][BN] tile so that every lane finishes 4
    // consecutive columns and stores 8 bytes (16 for fp32 scores), rows fully coalesced.
    // A tile larger than the ring (256x256) is staged in SL row slabs, each the i-th 32-row blocks of every wave row.
    constexpr int SL = ((size_t)BM * BN <= (size_t)NBUF * (BM + BN) * RF) ? 1 : ((size_t)BM * BN / 2 <= (size_t)NBUF * (BM + BN) * RF ? 2 : TM);
    constexpr int TMS = TM / SL;             // 32-row blocks per wave and slab
    static_assert(TM % SL == 0 && (size_t)(BM / SL) * BN <= (size_t)NBUF * (BM + BN) * RF, "output slab must fit in the ring");
    float* Ct = smem;
    constexpr int C4 = BN / 4;
    static_assert(NT % C4 == 0, "a thread keeps one column quad across its rows");
    f32x4 gs = {0.f, 0.f, 0.f, 0.f}, gq = {0.f, 0.f, 0.f, 0.f};     // GroupNorm statistics of this thread's 4 columns
    // a thread keeps its column quad over all its rows (NT % C4 == 0): the bias is loaded once, not per row
    const int mycol = n0 + (tid % C4) * 4;
    const f32x4 bias4 = (d.bias && mycol < N) ? *(const f32x4*)(d.bias + mycol) : f32x4{0.f, 0.f, 0.f, 0.f};
    // pass 2 of the two-pass column softmax: per-(batch entry, column) shift before and factor after the activation
    const f32x4 cshift4 = (d.col_shift && mycol < N) ? *(const f32x4*)(d.col_shift + (long long)blockIdx.z * d.colvec_bs + mycol)
                                                     : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 cmul4 = (d.col_mul && mycol < N) ? *(const f32x4*)(d.col_mul + (long long)blockIdx.z * d.colvec_bs + mycol)
                                                 : f32x4{1.f, 1.f, 1.f, 1.f};
    const bool simple_epi = !d.col_shift && !L.out_f32 && d.alpha == 1.0f && (d.act == SDM_ACT_NONE || d.act == SDM_ACT_SILU) &&
                            !L.no_simple_epi;
    const bool extras = res16 || res216 || dz16;
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      __syncthreads();                       // every wave is done with the last K-step's fragments / the previous slab
#pragma unroll
      for (int ii = 0; ii < TMS; ++ii)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int prow = wm * TMS * 32 + ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            Ct[prow * BN + wn * TN * 32 + j * 32 + l31] = acc[sl * TMS + ii][j][r];
          }
      __syncthreads();
      if (simple_epi) {
        // straight-line form of the loop below (alpha = 1, act in {none, SiLU}, bf16 output, no column factors): the thread's
        // column quad is fixed, option addresses only where the option is present (see conv3_row_bf16_kernel / igemm8)
        const int c4 = (tid % C4) * 4, pr0 = tid / C4;
        constexpr int TRIPS = (BM / SL) * C4 / NT;
        static_assert(((BM / SL) * C4) % NT == 0, "whole trips");
        if (mycol < N) {
          auto trip = [&](auto silu_c, int k) {
            constexpr bool SILU = decltype(silu_c)::value;
            const int prow = pr0 + k * (NT / C4);
            const int row = SL == 1 ? prow : (prow / (TMS * 32)) * (TM * 32) + sl * (TMS * 32) + prow % (TMS * 32);
            const long long orow = s_orow[row];
            if (orow >= 0) {
              f32x4 v = *(const f32x4*)(Ct + prow * BN + c4);
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] += bias4[q];
              if (pre16) {
                u16x4 zo;
#pragma unroll
                for (int q = 0; q < 4; ++q) zo[q] = f2bf(v[q]);
                *(u16x4*)(pre16 + obo + orow * d.ldo + mycol) = zo;
              }
              if constexpr (SILU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = sdm_silu(v[q]);
              }
              if (extras) {
                if (res16) {
                  const u16x4 rv = *(const u16x4*)(res16 + rbo + orow * d.ldres + mycol);
#pragma unroll
                  for (int q = 0; q < 4; ++q) v[q] += bf2f(rv[q]);
                }
                if (res216) {
                  const u16x4 rv = *(const u16x4*)(res216 + orow * d.ldres2 + mycol);
#pragma unroll
                  for (int q = 0; q < 4; ++q) v[q] += bf2f(rv[q]);
                }
                if (dz16 && (d.dact_ncols == 0 || mycol < d.dact_ncols)) {
                  const u16x4 zv = *(const u16x4*)(dz16 + orow * d.ldz + mycol);
#pragma unroll
                  for (int q = 0; q < 4; ++q) v[q] *= sdm_dsilu(bf2f(zv[q]));
                }
              }
              u16x4 o;
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = f2bf(v[q]);
              *(u16x4*)(out16 + obo + orow * d.ldo + mycol) = o;
              if (L.gn_stats) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const float r = bf2f(o[q]);
                  gs[q] += r;
                  gq[q] = fmaf(r, r, gq[q]);
                }
              }
            }
          };
          if (d.act == SDM_ACT_SILU) {
#pragma unroll 4
            for (int k = 0; k < TRIPS; ++k) trip(std::true_type(), k);
          } else {
#pragma unroll 4
            for (int k = 0; k < TRIPS; ++k) trip(std::false_type(), k);
          }
        }
        continue;
      }
#pragma unroll 2
      for (int e = tid; e < (BM / SL) * C4; e += NT) {
        const int prow = e / C4, c4 = (e - prow * C4) * 4;
        const int row = SL == 1 ? prow : (prow / (TMS * 32)) * (TM * 32) + sl * (TMS * 32) + prow % (TMS * 32);
        const int col = n0 + c4;
        const long long orow = s_orow[row];
        if (orow < 0 || col >= N) continue;
        f32x4 v = *(const f32x4*)(Ct + prow * BN + c4);
        if (d.col_shift) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float t = v[q] * d.alpha;
            asm volatile("" : "+v"(t));      // the rounded product pass 1 took its maxima from (no fma contraction)
            v[q] = __expf(t + cshift4[q]) * cmul4[q];
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = v[q] * d.alpha + bias4[q];
        }
        if (pre16) {     // pre-activation copy for the backward (training)
          u16x4 zo;
#pragma unroll
          for (int q = 0; q < 4; ++q) zo[q] = f2bf(v[q]);
          *(u16x4*)(pre16 + obo + orow * d.ldo + col) = zo;
        }
        if (!d.col_shift) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = sdm_act(v[q], d.act);
        }
        if (res16) {
          const u16x4 rv = *(const u16x4*)(res16 + rbo + orow * d.ldres + col);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += bf2f(rv[q]);
        }
        if (res216) {
          const u16x4 rv = *(const u16x4*)(res216 + orow * d.ldres2 + col);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += bf2f(rv[q]);
        }
        if (dz16 && (d.dact_ncols == 0 || col < d.dact_ncols)) {   // data gradient: times silu'(z) of the producer
          const u16x4 zv = *(const u16x4*)(dz16 + orow * d.ldz + col);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] *= sdm_dsilu(bf2f(zv[q]));
        }
        if (L.out_f32) {
          *(f32x4*)(d.out + obo + orow * d.ldo + col) = v;
        } else {
          u16x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = f2bf(v[q]);
          *(u16x4*)(out16 + obo + orow * d.ldo + col) = o;
          if (L.gn_stats) {       // statistics of the values the next layer will read (after rounding to bf16)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float r = bf2f(o[q]);
              gs[q] += r;
              gq[q] = fmaf(r, r, gq[q]);
            }
          }
        }
      }
    }
    if (L.gn_stats) {
      // NT / C4 threads share a column quad (rows e / C4 differ by NT / C4): fold them through LDS (the staged tile is
      // dead now), then one fp32 atomic pair per column and workgroup into [image][channel][2].  The host guarantees that
      // a tile's rows lie in ONE image (Ho*Wo % BM == 0) and that output rows are tile rows (plain 3x3 stride-1 conv).
      constexpr int GT = NT / C4;
      __syncthreads();
      float* red = smem;
      *(f32x4*)(red + (tid * 2) * 4) = gs;
      *(f32x4*)(red + (tid * 2 + 1) * 4) = gq;
      __syncthreads();
      if (tid < BN && n0 + tid < N && s_orow[0] >= 0) {
        const int cg = tid >> 2, q = tid & 3;
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int g = 0; g < GT; ++g) {
          a += red[((g * C4 + cg) * 2) * 4 + q];
          b += red[((g * C4 + cg) * 2 + 1) * 4 + q];
        }
        const long long img = s_orow[0] / ((long long)d.Ho * d.Wo);
        float* dst = d.gn_sums + (img * N + n0 + tid) * 2;
        atomicAdd(dst, a);
        atomicAdd(dst + 1, b);
      }
    }
    return;
  }
  if constexpr (BM > 128) return;     // the large tiles are only planned with the vector epilogue and without split-K
  if (L.splitk > 1) {                 // partial sums: nothing but the atomics
    float* wsb = d.ws + (long long)blockIdx.z * L.out_rows * N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + l31;
        if (col >= N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long orow = s_orow[wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
          if (orow >= 0) atomicAdd(wsb + orow * N + col, acc[i][j][r]);
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + l31;
      if (col >= N) continue;
      const float bias = d.bias ? d.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const long long orow = s_orow[row];
        if (orow < 0) continue;
        float v = acc[i][j][r];
        if (L.splitk > 1) {
          atomicAdd(&d.ws[((long long)blockIdx.z * L.out_rows + orow) * N + col], v);
          continue;
        }
        v = v * d.alpha + bias;
        if (pre16) pre16[obo + orow * d.ldo + col] = f2bf(v);
        v = sdm_act(v, d.act);
        if (res16) v += bf2f(res16[rbo + orow * d.ldres + col]);
        if (res216) v += bf2f(res216[orow * d.ldres2 + col]);
        if (dz16 && (d.dact_ncols == 0 || col < d.dact_ncols)) v *= sdm_dsilu(bf2f(dz16[orow * d.ldz + col]));
        if (L.out_f32) d.out[obo + orow * d.ldo + col] = v;
        else out16[obo + orow * d.ldo + col] = f2bf(v);
      }
    }
  }
}

// --------------------------------------------------------------------------------------------
// 3x3 stride-1 convs (forward and data gradient): one activation patch per kernel ROW.
//
// What bounds igemm_bf16_kernel is the volume moved L2 -> LDS (profiles/r02_bf16_tiles.txt), and the three taps of one
// kernel row (dx = -1, 0, +1) read the same BM consecutive pixels shifted by one.  Here a tile is BM consecutive pixels of
// one image; for a kernel row dy and a 64-channel chunk the BM + 2 pixels  t0 + dy W - 1 .. t0 + dy W + BM  are loaded
// ONCE (pixels outside the image from a zero chunk) and the three taps read their fragments from that image at row offset
// 1 + dx; lanes whose output pixel sits at x = 0 / W - 1 zero their fragment for dx = -1 / +1 (the patch holds the wrapped
// neighbour there).  Per K-step (one tap, 64 channels) only the weight tile moves: a third of the activation bytes, the
// K-step itself (32 MFMAs per wave and barrier at 256x256) unchanged.  LDS image as in igemm_bf16_kernel (128-byte rows,
// chunk ^ (row >> 1) & 7: conflict-free for any start row).  A full 9-tap halo (profiles: tools/scratch/halo) needed
// 32-channel steps to fit LDS and lost on the matrix side; one row at a time fits with 64-channel steps.
// --------------------------------------------------------------------------------------------
struct RowLaunch {
  sdm_igemm_desc d;
  int ntn, W, wmask, gn_stats;
  int no_simple_epi;    // SDM_IGEMM8_SIMPLE_EPI=0: the general epilogue loop (A/B)
  int dy[3];            // kernel-row offsets in the order they are walked
  int tapw[3][3];       // weight tap index of (row index, dx = -1, 0, +1)
};

template <int BM, int BN, int WM, int WN, int NBS = 2>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4 ? 2 : 1)) void conv3_row_bf16_kernel(const RowLaunch L) {
  const sdm_igemm_desc& d = L.d;
  constexpr int NT = 64 * WM * WN, NWAVE = WM * WN;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr int PROWS = BM + 2;                           // patch pixels
  constexpr int PINST = (PROWS + 7) / 8;                  // 1-KiB DMA instructions per patch (8 rows of 128 B each)
  constexpr int PW = (PINST + NWAVE - 1) / NWAVE;         // ... per wave
  constexpr int PBYTES = 128 + PINST * 1024;              // one guard row in front (read by masked lanes only)
  constexpr int BIT = BN / (8 * NWAVE);                   // weight-tile DMA instructions per wave (BN rows of 128 B)
  constexpr int BSTAGE = BN * 128;
  extern __shared__ __align__(16) char smem_r[];
  TR_STAMP(0);
  char* Pb = smem_r;                                      // 2 patch buffers
  char* Bs = smem_r + 2 * PBYTES;                         // NBS weight stages
  // NBS > 2 (the 8-wave 256x128 tile): the weight stage of K-step s + NBS - 1 is issued in K-step s, the wait in front of a
  // barrier is COUNTED (only the stage the coming step reads must have landed), and the fragment reads are inline asm -- next to
  // LDS-DMA the compiler puts s_waitcnt vmcnt(0) in front of every builtin LDS read (§4d), which would drain the ring each step.
  constexpr int DB = NBS - 1;                             // weight stages in flight behind the one being waited for + 1

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, h = lane >> 5, l31 = lane & 31;
  // (Persistent workgroups -- a grid of the resident slots, each walking several tiles -- were built and measured neutral,
  // 771 vs 771 TFLOP/s at 128 x 128x128 x 128 channels, profiles/r04_row_persist_ab.txt: with two workgroups per CU the hand-over
  // between workgroups is already hidden by the neighbour.  Not kept: 30 more registers and 60 spilled scalars.)
  const int tile_id = xcd_contiguous_bf(blockIdx.x, gridDim.x);
  const int tile_n = tile_id % L.ntn, tile_m = tile_id / L.ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int W = L.W, H = d.a.H, HW = H * W;
  const int img = m0 / HW, t0 = m0 - img * HW;             // tile start as a pixel index inside its image
  const int K = d.K, N = d.N, csplit = d.a.csplit;
  const int KC = K / BKE, kc0 = (csplit < K ? csplit : K) / BKE;
  const bf16_t* a0 = (const bf16_t*)d.a.p0;
  const bf16_t* a1 = (const bf16_t*)d.a.p1;
  const bf16_t* bw = (const bf16_t*)d.b;
  const bf16_t* zero = (const bf16_t*)g_zero_chunk_bf;
  if (tid < 64) ((unsigned*)(Pb + (tid >> 5) * PBYTES))[tid & 31] = 0u;     // the zero row in front of either patch
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  // ---- DMA roles: patch instruction k * NWAVE + wave covers patch rows 8 * inst .. +7; lane -> row lane >> 3, physical
  // chunk lane & 7, logical chunk = physical ^ swizzle(row)
  const int prl = lane >> 3;
  int pchunk[PW];
#pragma unroll
  for (int k = 0; k < PW; ++k) {
    const int row = (k * NWAVE + wave) * 8 + prl;
    pchunk[k] = 8 * ((lane & 7) ^ ((row >> 1) & 7));
  }
  const bf16_t* bp[BIT];
  bool bok[BIT];
#pragma unroll
  for (int k = 0; k < BIT; ++k) {
    const int n = (k * NWAVE + wave) * 8 + prl;
    bok[k] = n0 + n < N;
    bp[k] = bw + (long long)(n0 + n) * d.ldb + 8 * ((lane & 7) ^ ((n >> 1) & 7));
  }
  auto issue_patch = [&](int ri, int kc, int buf) {        // kernel row ri, chunk kc -> patch buffer buf
    const int lin0 = t0 + L.dy[ri] * W - 1;                // image pixel index of patch row 0
    const bool second = kc >= kc0;
    const bf16_t* src = second ? a1 : a0;
    const long long ld = second ? d.a.ld1 : d.a.ld0;
    const int ch = (second ? kc - kc0 : kc) * BKE;
    char* dst = Pb + buf * PBYTES + 128 + wave * 1024;
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const int inst = k * NWAVE + wave;
      if (inst < PINST) {
        const int lin = lin0 + inst * 8 + prl;
        const bool ok = (unsigned)lin < (unsigned)HW && inst * 8 + prl < PROWS;
        const bf16_t* p = ok ? src + ((long long)img * HW + lin) * ld + ch + pchunk[k] : zero;
        __builtin_amdgcn_global_load_lds((glb_ptr)p, (lds_ptr)(dst + k * NWAVE * 1024), 16, 0, 0);
      }
    }
  };
  auto issue_b = [&](int ri, int dxi, int kc, int slot) {
    const long long off = (long long)L.tapw[ri][dxi] * d.b_tap_stride + kc * BKE;
    char* dst = Bs + slot * BSTAGE + wave * 1024;
#pragma unroll
    for (int k = 0; k < BIT; ++k) {
      const bf16_t* p = bok[k] ? bp[k] + off : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr)p, (lds_ptr)(dst + k * NWAVE * 1024), 16, 0, 0);
    }
  };
  // ---- fragment roles
  // patch row of (tile pixel t, dx) is t + 1 + dx; lanes whose pixel has no left / right neighbour read the zeroed guard
  // row in front of the patch instead (one select per step, nothing per k-group)
  int t128[TM];
  unsigned deadm = 0;                                       // bit 2i: no left neighbour, bit 2i + 1: no right neighbour
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int t = wm * TM * 32 + i * 32 + l31;              // tile pixel
    const int tx = (t0 + t) & L.wmask;
    t128[i] = t * 128;
    if (tx == 0) deadm |= 1u << (2 * i);
    if (tx == W - 1) deadm |= 2u << (2 * i);
  }
  int boff[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) boff[j] = (wn * TN * 32 + j * 32 + l31) * 128;
  const int sw_b = (l31 >> 1) & 7;                          // weight rows: (n >> 1) & 7 == (l31 >> 1) & 7 (32-row blocks)

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  __syncthreads();
  const int NP = 3 * KC;                                    // patches: (kernel row, chunk), chunk fastest
  TR_STAMP(1);
  const int S = 3 * NP;                                     // K-steps (= weight stages): step s = patch s / 3, column tap s % 3
  auto issue_stage = [&](int s2) {                          // weight stage of K-step s2 (if it exists) into slot s2 % NBS
    if (s2 < S) {
      const int pj = s2 / 3, dxj = s2 - pj * 3;
      const int rj = pj / KC, kj = pj - rj * KC;
      issue_b(rj, dxj, kj, s2 % NBS);
    }
  };
  issue_patch(0, 0, 0);
  if constexpr (NBS > 2) {
#pragma unroll
    for (int j = 0; j < DB; ++j) issue_stage(j);
  } else {
    issue_b(0, 0, 0, 0);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem_r;
  int ri = 0, kc = 0;                                       // (kernel row, chunk) of the patch being computed
  for (int pi = 0; pi < NP; ++pi) {
    int nri = ri, nkc = kc + 1;                             // ... of the next patch
    if (nkc == KC) { nkc = 0; ++nri; }
    const char* Pq = Pb + (pi & 1) * PBYTES + 128;
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) {
      const int s = pi * 3 + dxi;
      if constexpr (NBS > 2) {
        // issue order inside a step: [next patch if dxi == 0], weight stage s + DB.  Behind the stage this step reads (issued
        // DB steps ago) lie DB - 1 weight stages (BIT instructions each) and, if one of the steps s - DB + 1 .. s - 1 was a
        // dxi == 0 step, that step's patch (>= 4 instructions per wave: PINST = 17 or 33 over 4 or 8 waves).
        constexpr int PMIN = PINST / NWAVE;
        const bool patch_behind = (DB == 2) ? (dxi == 1) : (dxi == 1 || dxi == 2);     // (dxi is a constant after unrolling)
        if (s + DB >= S) wait_vmcnt_bf<0>();
        else if (patch_behind) wait_vmcnt_bf<(DB - 1) * BIT + PMIN>();
        else wait_vmcnt_bf<(DB - 1) * BIT>();
      } else {
        wait_vmcnt_bf<0>();
      }
      __builtin_amdgcn_s_barrier();                         // stage s (and patch pi) visible; stage s - 1 / patch pi - 1 free
      if constexpr (NBS > 2) {
        if (dxi == 0 && pi + 1 < NP) issue_patch(nri, nkc, (pi + 1) & 1);
        issue_stage(s + DB);
      } else {
        if (dxi < 2) issue_b(ri, dxi + 1, kc, (s + 1) & 1);
        else if (pi + 1 < NP) issue_b(nri, 0, nkc, (s + 1) & 1);
        if (dxi == 0 && pi + 1 < NP) issue_patch(nri, nkc, (pi + 1) & 1);
      }
      const char* Bq = Bs + (s % NBS) * BSTAGE;
      f32x4 af[2][TM], bf[2][TN];
      int aoff[TM];                                         // (the chunk swizzle (q >> 1) & 7 is bits 8..10 of q * 128)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const bool dead = dxi == 0 ? (deadm >> (2 * i)) & 1u : (dxi == 2 ? (deadm >> (2 * i + 1)) & 1u : false);
        aoff[i] = dead ? -128 : t128[i] + dxi * 128;
      }
      if constexpr (NBS > 2) {
        const unsigned pq0 = lds0 + (unsigned)(Pq - smem_r), bq0 = lds0 + (unsigned)(Bq - smem_r);
        auto ldfrag_asm = [&](int kg, int st) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const unsigned ad = pq0 + (unsigned)(aoff[i] + (((kg * 2 + h) ^ ((aoff[i] >> 8) & 7)) << 4));
            asm volatile("ds_read_b128 %0, %1" : "=v"(af[st][i]) : "v"(ad) : "memory");
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const unsigned ad = bq0 + (unsigned)(boff[j] + (((kg * 2 + h) ^ sw_b) << 4));
            asm volatile("ds_read_b128 %0, %1" : "=v"(bf[st][j]) : "v"(ad) : "memory");
          }
        };
        ldfrag_asm(0, 0);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
          if (kg < 3) {
            ldfrag_asm(kg + 1, (kg + 1) & 1);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN) : "memory");   // the set read one trip ago has landed
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
#pragma unroll
          for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(af[kg & 1][i]));
#pragma unroll
          for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bf[kg & 1][j]));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[kg & 1][i]),
                                                                  __builtin_bit_cast(bf16x8, bf[kg & 1][j]), acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
      auto ldfrag = [&](int kg, int st) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[st][i] = *(const f32x4*)(Pq + aoff[i] + (((kg * 2 + h) ^ ((aoff[i] >> 8) & 7)) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[st][j] = *(const f32x4*)(Bq + boff[j] + (((kg * 2 + h) ^ sw_b) << 4));
      };
      ldfrag(0, 0);
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        if (kg < 3) ldfrag(kg + 1, (kg + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[kg & 1][i]),
                                                                __builtin_bit_cast(bf16x8, bf[kg & 1][j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
    }
    ri = nri;
    kc = nkc;
  }
  if constexpr (NBS > 2) __syncthreads();      // (asm reads: every wave is done with the ring before the epilogue reuses it)

  // ---- epilogue (the vector epilogue of igemm_bf16_kernel; output row of tile row r is m0 + r)
  TR_STAMP(2);
  bf16_t* out16 = (bf16_t*)d.out;
  bf16_t* pre16 = (bf16_t*)d.out_preact;
  const bf16_t* res16 = (const bf16_t*)d.res;
  const bf16_t* res216 = (const bf16_t*)d.res2;
  const bf16_t* dz16 = (const bf16_t*)d.dact_z;
  constexpr size_t LDSB = (size_t)2 * PBYTES + NBS * BSTAGE;
  constexpr int SL = ((size_t)BM * BN * 4 <= LDSB) ? 1 : 2;
  constexpr int TMS = TM / SL;
  static_assert(TM % SL == 0 && (size_t)(BM / SL) * BN * 4 <= LDSB, "output slab must fit");
  float* Ct = (float*)smem_r;
  constexpr int C4 = BN / 4;
  static_assert(NT % C4 == 0, "a thread keeps one column quad");
  f32x4 gs = {0.f, 0.f, 0.f, 0.f}, gq = {0.f, 0.f, 0.f, 0.f};
  const int mycol = n0 + (tid % C4) * 4;           // a thread keeps its column quad over all its rows: bias loaded once
  const f32x4 bias4 = (d.bias && mycol < N) ? *(const f32x4*)(d.bias + mycol) : f32x4{0.f, 0.f, 0.f, 0.f};
  const bool simple_epi = d.alpha == 1.0f && (d.act == SDM_ACT_NONE || d.act == SDM_ACT_SILU) && !L.no_simple_epi;
  const bool extras = res16 || res216 || dz16;
#pragma unroll
  for (int sl = 0; sl < SL; ++sl) {
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < TMS; ++ii)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int prow_ = wm * TMS * 32 + ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          Ct[prow_ * BN + wn * TN * 32 + j * 32 + l31] = acc[sl * TMS + ii][j][r];
        }
    __syncthreads();
    if (simple_epi) {
      // straight-line form of the loop below for alpha = 1 and act in {none, SiLU} (every conv of the U-Net): the thread's
      // column quad is fixed, rows advance by NT / C4 per trip, option addresses are taken only where the option is present
      // (the general loop is ~100 instructions per trip; same change and measurements as igemm8's, profiles/r04_igemm8_epi_ab.txt)
      const int c4 = (tid % C4) * 4, pr0 = tid / C4;
      constexpr int TRIPS = (BM / SL) * C4 / NT;
      static_assert(((BM / SL) * C4) % NT == 0, "whole trips");
      if (mycol < N) {
        auto trip = [&](auto silu_c, int k) {
          constexpr bool SILU = decltype(silu_c)::value;
          const int prow_ = pr0 + k * (NT / C4);
          const int row = SL == 1 ? prow_ : (prow_ / (TMS * 32)) * (TM * 32) + sl * (TMS * 32) + prow_ % (TMS * 32);
          const long long orow = m0 + row;
          f32x4 v = *(const f32x4*)(Ct + prow_ * BN + c4);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += bias4[q];
          if (pre16) {
            u16x4 zo;
#pragma unroll
            for (int q = 0; q < 4; ++q) zo[q] = f2bf(v[q]);
            *(u16x4*)(pre16 + orow * d.ldo + mycol) = zo;
          }
          if constexpr (SILU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = sdm_silu(v[q]);
          }
          if (extras) {
            if (res16) {
              const u16x4 rv = *(const u16x4*)(res16 + orow * d.ldres + mycol);
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] += bf2f(rv[q]);
            }
            if (res216) {
              const u16x4 rv = *(const u16x4*)(res216 + orow * d.ldres2 + mycol);
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] += bf2f(rv[q]);
            }
            if (dz16 && (d.dact_ncols == 0 || mycol < d.dact_ncols)) {
              const u16x4 zv = *(const u16x4*)(dz16 + orow * d.ldz + mycol);
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] *= sdm_dsilu(bf2f(zv[q]));
            }
          }
          u16x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = f2bf(v[q]);
          *(u16x4*)(out16 + orow * d.ldo + mycol) = o;
          if (L.gn_stats) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float r = bf2f(o[q]);
              gs[q] += r;
              gq[q] = fmaf(r, r, gq[q]);
            }
          }
        };
        if (d.act == SDM_ACT_SILU) {
#pragma unroll 4
          for (int k = 0; k < TRIPS; ++k) trip(std::true_type(), k);
        } else {
#pragma unroll 4
          for (int k = 0; k < TRIPS; ++k) trip(std::false_type(), k);
        }
      }
      continue;
    }
#pragma unroll 2
    for (int e = tid; e < (BM / SL) * C4; e += NT) {
      const int prow_ = e / C4, c4 = (e - prow_ * C4) * 4;
      const int row = SL == 1 ? prow_ : (prow_ / (TMS * 32)) * (TM * 32) + sl * (TMS * 32) + prow_ % (TMS * 32);
      const int col = n0 + c4;
      const long long orow = m0 + row;
      if (col >= N) continue;
      f32x4 v = *(const f32x4*)(Ct + prow_ * BN + c4);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = v[q] * d.alpha + bias4[q];
      if (pre16) {
        u16x4 zo;
#pragma unroll
        for (int q = 0; q < 4; ++q) zo[q] = f2bf(v[q]);
        *(u16x4*)(pre16 + orow * d.ldo + col) = zo;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = sdm_act(v[q], d.act);
      if (res16) {
        const u16x4 rv = *(const u16x4*)(res16 + orow * d.ldres + col);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += bf2f(rv[q]);
      }
      if (res216) {
        const u16x4 rv = *(const u16x4*)(res216 + orow * d.ldres2 + col);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += bf2f(rv[q]);
      }
      if (dz16 && (d.dact_ncols == 0 || col < d.dact_ncols)) {
        const u16x4 zv = *(const u16x4*)(dz16 + orow * d.ldz + col);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] *= sdm_dsilu(bf2f(zv[q]));
      }
      u16x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = f2bf(v[q]);
      *(u16x4*)(out16 + orow * d.ldo + col) = o;
      if (L.gn_stats) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float r = bf2f(o[q]);
          gs[q] += r;
          gq[q] = fmaf(r, r, gq[q]);
        }
      }
    }
  }
  if (L.gn_stats) {
    constexpr int GT = NT / C4;
    __syncthreads();
    float* red = (float*)smem_r;
    *(f32x4*)(red + (tid * 2) * 4) = gs;
    *(f32x4*)(red + (tid * 2 + 1) * 4) = gq;
    __syncthreads();
    if (tid < BN && n0 + tid < N) {
      const int cg = tid >> 2, q = tid & 3;
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int g = 0; g < GT; ++g) {
        a += red[((g * C4 + cg) * 2) * 4 + q];
        b += red[((g * C4 + cg) * 2 + 1) * 4 + q];
      }
      float* dst = d.gn_sums + ((long long)img * N + n0 + tid) * 2;
      atomicAdd(dst, a);
      atomicAdd(dst + 1, b);
    }
  }
  TR_STAMP(3);
}

// finishes a split-K contraction (ws [batch entry][out row][N] raw fp32 sums, left zero-filled); rows = per entry
__global__ __launch_bounds__(256) void splitk_epilogue_bf16_kernel(const sdm_igemm_desc d, long long rows, int out_f32) {
  const long long total = rows * d.batch * d.N;
  bf16_t* out16 = (bf16_t*)d.out;
  const bf16_t* res16 = (const bf16_t*)d.res;
  const bf16_t* res216 = (const bf16_t*)d.res2;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long grow = sdm_idiv(e, d.N);
    const int col = (int)(e - grow * d.N);
    const long long b = sdm_idiv(grow, rows), orow = grow - b * rows;
    const long long b0 = sdm_idiv(b, d.bdiv), b1 = b - b0 * d.bdiv;
    const long long oo = b0 * d.o_bs0 + b1 * d.o_bs1 + orow * d.ldo + col;
    float v = d.ws[e] * d.alpha + (d.bias ? d.bias[col] : 0.f);
    d.ws[e] = 0.f;
    if (d.out_preact) ((bf16_t*)d.out_preact)[oo] = f2bf(v);
    v = sdm_act(v, d.act);
    if (res16) v += bf2f(res16[b0 * d.r_bs0 + b1 * d.r_bs1 + orow * d.ldres + col]);
    if (res216) v += bf2f(res216[orow * d.ldres2 + col]);
    if (d.dact_z && (d.dact_ncols == 0 || col < d.dact_ncols)) v *= sdm_dsilu(bf2f(((const bf16_t*)d.dact_z)[orow * d.ldz + col]));
    if (out_f32) d.out[oo] = v;
    else out16[oo] = f2bf(v);
  }
}

static int bf_validate(const sdm_igemm_desc* d, int out_f32) {
  SDM_CHECK_ARG(d != nullptr, "igemm_bf16: null desc");
  SDM_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "igemm_bf16: bad M/N/K %d %d %d", d->M, d->N, d->K);
  SDM_CHECK_ARG(d->batch >= 1 && d->bdiv >= 1, "igemm_bf16: bad batch");
  SDM_CHECK_ARG(d->a.p0 && d->b && (d->out || d->col_stats), "igemm_bf16: null operand");
  SDM_CHECK_ARG(d->Ht > 0 && d->Wt > 0 && d->M % (d->Ht * d->Wt) == 0, "igemm_bf16: M %d not a multiple of Ht*Wt", d->M);
  SDM_CHECK_ARG(d->nphase >= 1 && d->nphase <= SDM_MAX_PHASE, "igemm_bf16: nphase");
  for (int p = 0; p < d->nphase; ++p) SDM_CHECK_ARG(d->ntaps[p] >= 1 && d->ntaps[p] <= SDM_MAX_TAPS, "igemm_bf16: ntaps");
  SDM_CHECK_ARG(d->a.p1 != nullptr || d->a.csplit >= d->K, "igemm_bf16: csplit < K without second source");
  SDM_CHECK_ARG(d->K % 8 == 0 && (d->a.p1 == nullptr || d->a.csplit % 8 == 0), "igemm_bf16: K and the concat split must be multiples of 8");
  SDM_CHECK_ARG(aligned16(d->a.p0) && d->a.ld0 % 8 == 0 && d->a_bs0 % 8 == 0 && d->a_bs1 % 8 == 0 &&
                    (d->a.p1 == nullptr || (aligned16(d->a.p1) && d->a.ld1 % 8 == 0)),
                "igemm_bf16: A rows must be 16-byte aligned (pitches multiples of 8 elements)");
  SDM_CHECK_ARG(aligned16(d->b) && d->ldb % 8 == 0 && d->b_tap_stride % 8 == 0 && d->b_bs0 % 8 == 0 && d->b_bs1 % 8 == 0,
                "igemm_bf16: B rows must be 16-byte aligned");
  SDM_CHECK_ARG(d->b_kmajor == 0, "igemm_bf16: K-major B is not supported (transpose the operand: sdm_transpose_bf16)");
  SDM_CHECK_ARG(d->a.scale == nullptr, "igemm_bf16: no affine-on-load on bf16 rows (materialise it: sdm_affine_residual_bf16)");
  SDM_CHECK_ARG(!(out_f32 && (d->res || d->res2 || d->out_preact)), "igemm_bf16: fp32 output takes no residual / pre-activation copy");
  SDM_CHECK_ARG(d->ostride >= 1 && d->sstride >= 1 && d->Ho > 0 && d->Wo > 0, "igemm_bf16: bad strides/output dims");
  SDM_CHECK_ARG((d->act >= 0 && d->act <= 2) || (d->act == SDM_ACT_EXP && d->col_shift), "igemm_bf16: bad act");
  SDM_CHECK_ARG((d->col_shift == nullptr) == (d->col_mul == nullptr), "igemm_bf16: col_shift and col_mul come together");
  SDM_CHECK_ARG(d->col_shift == nullptr || (d->act == SDM_ACT_EXP && !d->bias && !d->res && !d->res2 && !d->out_preact &&
                                            !d->dact_z && aligned16(d->col_shift) && aligned16(d->col_mul) && d->colvec_bs % 4 == 0),
                "igemm_bf16: the column shift / factor epilogue is exp(alpha*acc + shift) * factor alone (16-byte aligned vectors)");
  SDM_CHECK_ARG(d->col_stats == nullptr || (!d->col_shift && !d->bias && !d->res && !d->res2 && !d->out_preact && !d->dact_z &&
                                            !d->defer_epilogue && d->nphase == 1),
                "igemm_bf16: column statistics take alpha only");
  SDM_CHECK_ARG(!d->defer_epilogue || (d->batch == 1 && !d->res && !d->res2 && !d->out_preact && !d->dact_z && !out_f32),
                "igemm_bf16: a deferred epilogue carries only alpha / bias / activation (batch 1, bf16 output)");
  return SDM_OK;
}

// Tile configurations.  Measured on MI355X (profiles/r01_bf16_probe.txt, profiles/r02_bf16_tiles.txt): operand movement
// L2 -> LDS (64 flop per tile byte at 128x128, 128 at 256x256) and LDS fragment reads (one ds_read_b128 per MFMA at a
// 64x64 wave tile, 0.75 at 128x64) set the pace, not the matrix pipe, and every wave needs a partner on its SIMD to hide
// DMA latency and barriers (one wave per SIMD -- 4-wave 256x128 / 256x256 tiles, or ring 3 with one workgroup -- measured
// 0.7-0.9x of the configurations below).
//   cfg 0: 128x128, 4 waves (2x2), ring 2 (64 KB): two workgroups per CU
//   cfg 1:  64x64,  4 waves (2x2), ring 3 (48 KB): three workgroups per CU; small problems, with split-K
//   cfg 2: 256x256, 8 waves (2x4, wave tile 128x64), ring 2 (128 KB): one workgroup per CU, two waves per SIMD; 1.17x the
//          rate of cfg 0 once a launch has >= 256 such tiles (721 -> 843 TFLOP/s at 512 channels, 630 -> 751 at 256)
//   cfg 3: as cfg 2 with 4x2 waves (wave tile 64x128): same rate within noise, force override only
struct BfCfg { int bm, bn; };
static const BfCfg kBfCfgs[4] = {{128, 128}, {64, 64}, {256, 256}, {256, 256}};

static void bf_plan(const sdm_igemm_desc* d, bool big_ok, int* cfg, int* splitk, size_t* ws_need) {
  static const char* force = getenv("SDM_BF16_FORCE");   // "cfg,splitk" tuning override
  const long ph = (long)d->nphase * d->batch;
  int maxtaps = 0;
  for (int p = 0; p < d->nphase; ++p) maxtaps = d->ntaps[p] > maxtaps ? d->ntaps[p] : maxtaps;
  const int T = maxtaps * cdiv(d->K, BKE);
  const long t128 = (long)cdiv(d->M, 128) * cdiv(d->N, 128) * ph;
  const long t64 = (long)cdiv(d->M, 64) * cdiv(d->N, 64) * ph;
  int c = t128 >= 320 ? 0 : 1, sk = 1;
  if (d->N <= 64) c = 1;     // the 3-channel head conv: a 128-wide tile would be 98 % padding, a 64-wide one 95 %
  if (c == 0 && big_ok) {
    // rounds of resident workgroups (512 slots of 128x128, 256 of 256x256); a 256x256 tile is 4 tile units at 1.17x the
    // rate of a co-resident pair: 4 / (2 * 1.17) = 1.71 rounds of cfg 0
    static const char* nobig = getenv("SDM_BF16_BIG");
    const long t256 = (long)cdiv(d->M, 256) * cdiv(d->N, 256) * ph;
    const double cost128 = (double)cdiv(t128, 512), cost256 = 1.71 * (double)cdiv(t256, 256);
    if (cost256 < cost128 && !(nobig && nobig[0] == '0')) c = 2;
  }
  if (c == 1 && t64 < 512) {
    // Split K only where it pays for its finishing launch.  A 64-deep K-step of a 64x64 tile is ~0.3 us (operand fetch
    // bound), the finishing pass (a dependent launch reading atomically summed fp32) ~5 us of a sampling step's chain
    // (profiles/r03_sample_n1_bf16_kernel_stats.csv: 37 finishing launches at 4.9 us were 20 % of the N=1 step, most of them
    // behind K walks of 8-18 steps that a single workgroup finishes sooner than that).
    static const char* ts_s = getenv("SDM_BF16_TSTEP");
    static const char* te_s = getenv("SDM_BF16_SPLIT_COST");
    const double ts = ts_s ? atof(ts_s) : 0.30, tepi = te_s ? atof(te_s) : 5.0;
    int cap = T / 4;
    if (cap > 64) cap = 64;
    double best = (double)cdiv(t64, 512) * T * ts;
    for (int k2 = 2; k2 <= cap; ++k2) {
      const double t = (double)cdiv(t64 * k2, 512) * ((double)T / k2) * ts + tepi;
      if (t < best) { best = t; sk = k2; }
    }
  }
  // batched problems (attention cores): the scratch is indexed per batch entry, so splitting works, but at the batch
  // sizes where these GEMMs are tiny the sampling step is launch-bound and the extra finishing launch costs more than
  // the shorter K walk saves (measured: 837 -> 801 steps/s at batch 4) -- left to the force override
  if (d->batch > 1) sk = 1;
  if (d->col_stats || d->col_shift) sk = 1;      // the column epilogues see whole sums
  if (force) {
    int fc = 0, fs = 1;
    if (sscanf(force, "%d,%d", &fc, &fs) == 2 && fc >= 0 && fc <= 3 && (fc < 2 || big_ok)) { c = fc; sk = fc >= 2 ? 1 : (fs < 1 ? 1 : (fs > T ? T : fs)); }
    if (d->col_stats || d->col_shift) sk = 1;
  }
  *cfg = c;
  *splitk = sk;
  const long out_rows = (long)(d->M / (d->Ht * d->Wt)) * d->Ho * d->Wo;
  *ws_need = sk > 1 ? (size_t)out_rows * d->batch * d->N * sizeof(float) : 0;
}

template <int BM, int BN, int WM, int WN, int NBUF, bool FAST, bool XS = false, int KE = 64, bool PP = false>
static int launch_bf(const BfLaunch& L0, int ntm, hipStream_t st) {
  constexpr size_t lds = (size_t)NBUF * (BM + BN) * (KE / 2) * sizeof(float) + BM * sizeof(long long) + 3 * SDM_MAX_TAPS * sizeof(int);
  BfLaunch L = L0;
  L.kchunks = cdiv(L.d.K, KE);
  static bool attr_set = false;
  auto kern = igemm_bf16_kernel<BM, BN, WM, WN, NBUF, FAST, XS, KE, PP>;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid(ntm * L.ntn, L.d.nphase * L.splitk, L.d.batch);
  double taps = 0;
  for (int p = 0; p < L.d.nphase; ++p) taps += L.d.ntaps[p];
  {
    const double images = (double)L.d.M / ((double)L.d.Ht * L.d.Wt);
    const double in_b = images * L.d.a.H * L.d.a.W * L.d.K, w_b = (double)L.d.K * L.d.N * taps, out_b = images * L.d.Ho * L.d.Wo * L.d.N;
    sdm_prof_shape(L.d.M, L.d.N, L.d.K, (int)taps, L.splitk, 2.0 * L.d.batch * (in_b + w_b + out_b));
  }
  sdm_prof_pre(st);
  SDM_LAUNCH(kern, grid, dim3(64 * WM * WN), lds, st, L);
  sdm_prof_post(st, BM >= 128 ? 0 : 1, 2.0 * L.d.M * L.d.N * L.d.K * taps * L.d.batch);
  SDM_CHECK_LAUNCH("igemm_bf16");
  return SDM_OK;
}

// row-patch kernel: plain 3x3 stride-1 geometry (forward or data gradient), image width a power of two, tiles inside one
// image, unsplit, vector epilogue.  Returns 0 (not eligible) or the tile edge (128: the kernel replaces the 128x128 tiles of
// igemm_bf16_kernel; a 256x256 variant ran out of registers and lost to the plain 256x256 tile, profiles/r02_bf16_tiles.txt).
static int row_tile(const sdm_igemm_desc* d, int out_f32, int vec_epi, RowLaunch* R) {
  static const char* sw = getenv("SDM_BF16_ROW");          // "0": off
  if (sw && sw[0] == '0') return 0;
  if (out_f32 || !vec_epi || d->batch != 1 || d->nphase != 1 || d->sstride != 1 || d->ostride != 1 || d->defer_epilogue) return 0;
  if (d->ntaps[0] != 9 || d->Ht != d->a.H || d->Wt != d->a.W || d->Ho != d->Ht || d->Wo != d->Wt) return 0;
  const int W = d->a.W, H = d->a.H;
  if (W < 2 || (W & (W - 1)) != 0 || ((long)H * W) % 128 != 0) return 0;
  if (d->K % BKE != 0 || (d->a.p1 != nullptr && d->a.csplit % BKE != 0) || d->N < 64) return 0;
  int seen[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int t = 0; t < 9; ++t) {
    const int ty = d->tap_dy[0][t], tx = d->tap_dx[0][t];
    if (ty < -1 || ty > 1 || tx < -1 || tx > 1 || seen[ty + 1][tx + 1]) return 0;
    seen[ty + 1][tx + 1] = 1;
    R->tapw[ty + 1][tx + 1] = d->tap_w[0][t];
  }
  for (int i = 0; i < 3; ++i) R->dy[i] = i - 1;
  R->W = W;
  R->wmask = W - 1;
  return 128;
}

template <int BM, int BN, int WM, int WN, int NBS = 2>
static int launch_row(const sdm_igemm_desc* d, RowLaunch& R, int gn_stats, hipStream_t st) {
  constexpr int PINST = (BM + 2 + 7) / 8;
  constexpr size_t lds = (size_t)2 * (128 + PINST * 1024) + (size_t)NBS * BN * 128;
  static bool attr_set = false;
  auto kern = conv3_row_bf16_kernel<BM, BN, WM, WN, NBS>;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  R.d = *d;
  R.ntn = cdiv(d->N, BN);
  R.gn_stats = gn_stats;
  static const char* se_s = getenv("SDM_IGEMM8_SIMPLE_EPI");
  R.no_simple_epi = (se_s && se_s[0] == '0') ? 1 : 0;
  {
    const double images = (double)d->M / ((double)d->Ht * d->Wt);
    const double in_b = images * d->a.H * d->a.W * d->K, w_b = (double)d->K * d->N * 9, out_b = images * d->Ho * d->Wo * d->N;
    sdm_prof_shape(d->M, d->N, d->K, 9, 1, 2.0 * (in_b + w_b + out_b));
  }
  sdm_prof_pre(st);
  SDM_LAUNCH(kern, dim3((d->M / BM) * R.ntn), dim3(64 * WM * WN), lds, st, R);
  sdm_prof_post(st, 0, 2.0 * d->M * d->N * d->K * 9);
  SDM_CHECK_LAUNCH("conv3_row_bf16");
  return SDM_OK;
}

extern "C" size_t sdm_igemm_bf16_workspace_bytes(const sdm_igemm_desc* d) {
  if (bf_validate(d, 0) != SDM_OK) return 0;
  int b, sk;
  size_t need;
  bf_plan(d, false, &b, &sk, &need);     // an upper bound: the large tiles never split K
  return need;
}

extern "C" int sdm_igemm_bf16_tile_m(const sdm_igemm_desc* d) {
  if (bf_validate(d, 0) != SDM_OK) return -1;
  static const char* nofast = getenv("SDM_BF16_FAST");
  const bool fast = d->K % BKE == 0 && (d->a.p1 == nullptr || d->a.csplit % BKE == 0) && !(nofast && nofast[0] == '0');
  int cfg, sk;
  size_t need;
  // the large tiles need the vector epilogue; pass 1 stores no tile and pass 2's caller guarantees it (N % 4 == 0, aligned
  // rows), so both passes of a column softmax plan the same tiles
  bf_plan(d, fast && d->N % 4 == 0, &cfg, &sk, &need);
  return kBfCfgs[cfg].bm;
}

__global__ __launch_bounds__(256) void colstats_combine_kernel(const float* __restrict__ st, int ntiles, int N,
                                                               float* __restrict__ neg_max, float* __restrict__ inv_sum) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  const float* p = st + (long long)blockIdx.y * ntiles * N * 2;
  float mx = -INFINITY;
  for (int t = 0; t < ntiles; ++t) mx = fmaxf(mx, p[((long long)t * N + j) * 2]);
  float sm = 0.f;
  for (int t = 0; t < ntiles; ++t) {
    const float m = p[((long long)t * N + j) * 2], l = p[((long long)t * N + j) * 2 + 1];
    if (m > -INFINITY) sm += l * __expf(m - mx);
  }
  neg_max[(long long)blockIdx.y * N + j] = -mx;
  inv_sum[(long long)blockIdx.y * N + j] = 1.f / sm;
}
extern "C" int sdm_attn_colstats_combine(const float* col_stats, int batch, int ntiles, int N, float* neg_max, float* inv_sum,
                                         sdm_stream_t s) {
  SDM_CHECK_ARG(col_stats && neg_max && inv_sum && batch > 0 && ntiles > 0 && N > 0, "attn_colstats_combine: bad args");
  SDM_LAUNCH(colstats_combine_kernel, dim3(cdiv(N, 256), batch), dim3(256), 0, (hipStream_t)s, col_stats, ntiles, N,
                     neg_max, inv_sum);
  SDM_CHECK_LAUNCH("attn_colstats_combine");
  return SDM_OK;
}

extern "C" int sdm_igemm_bf16(const sdm_igemm_desc* d, int out_f32, sdm_stream_t stream) {
  int rc = bf_validate(d, out_f32);
  if (rc != SDM_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  BfLaunch L;
  L.d = *d;
  L.out_f32 = out_f32 ? 1 : 0;
  L.kchunks = cdiv(d->K, BKE);
  L.out_rows = (long long)(d->M / (d->Ht * d->Wt)) * d->Ho * d->Wo;
  L.lw = ilog2_exact_bf(d->Wt);
  L.lhw = ilog2_exact_bf((long)d->Ht * d->Wt);
  L.pow2 = L.lw >= 0 && L.lhw >= 0;
  L.plain_out = d->nphase == 1 && d->ostride == 1 && d->Ho == d->Ht && d->Wo == d->Wt && d->ph_oy[0] == 0 && d->ph_ox[0] == 0;
  {
    const size_t esz = out_f32 ? 4 : 2;
    auto al = [](const void* q, size_t a) { return (((uintptr_t)q) & (a - 1)) == 0; };
    static const char* noepi = getenv("SDM_BF16_VEC_EPI");
    L.vec_epi = d->col_stats ? d->N % 4 == 0 :
                d->N % 4 == 0 && al(d->out, 4 * esz) && d->ldo % 4 == 0 && d->o_bs0 % 4 == 0 && d->o_bs1 % 4 == 0 &&
                (d->bias == nullptr || al(d->bias, 16)) &&
                (d->res == nullptr || (al(d->res, 8) && d->ldres % 4 == 0 && d->r_bs0 % 4 == 0 && d->r_bs1 % 4 == 0)) &&
                (d->res2 == nullptr || (al(d->res2, 8) && d->ldres2 % 4 == 0)) &&
                (d->out_preact == nullptr || al(d->out_preact, 8)) &&
                (d->dact_z == nullptr || (al(d->dact_z, 8) && d->ldz % 4 == 0 && d->dact_ncols % 4 == 0)) &&
                !(noepi && noepi[0] == '0');
  }
  static const char* nofast = getenv("SDM_BF16_FAST");
  const bool fast = d->K % BKE == 0 && (d->a.p1 == nullptr || d->a.csplit % BKE == 0) && !(nofast && nofast[0] == '0');
  int cfg;
  size_t need;
  bf_plan(d, fast && L.vec_epi, &cfg, &L.splitk, &need);
  if (L.splitk > 1 && (d->ws == nullptr || d->ws_bytes < need)) L.splitk = 1;
  if (d->col_shift && !L.vec_epi)
    return sdm_set_error(SDM_ERR_ARG, "igemm_bf16: the column shift / factor epilogue needs N %% 4 == 0 and aligned output rows");
  const int bm = kBfCfgs[cfg].bm, bn = kBfCfgs[cfg].bn;
  const int ntm = cdiv(d->M, bm);
  L.ntn = cdiv(d->N, bn);
  {
    int taps_all = 0;
    for (int ph = 0; ph < d->nphase; ++ph) taps_all += d->ntaps[ph];
    const long c = cdiv((long)ntm * L.ntn, 8);
    const double a_tile = (double)bm * d->K, b_tile = (double)bn * d->K * taps_all;
    const double nfast = (double)cdiv(c, L.ntn) * a_tile + (double)(c < L.ntn ? c : L.ntn) * b_tile;
    const double mfast = (double)(c < ntm ? c : ntm) * a_tile + (double)cdiv(c, ntm) * b_tile;
    L.mfast = mfast < nfast ? 1 : 0;
  }
  // GroupNorm statistics in the epilogue: vector epilogue, no split, plain row order, tiles inside one image
  L.gn_stats = d->gn_sums != nullptr && !out_f32 && L.vec_epi && L.splitk == 1 && d->batch == 1 && d->nphase == 1 &&
               d->ostride == 1 && d->sstride == 1 && d->Ht == d->Ho && d->Wt == d->Wo && d->N % 4 == 0 &&
               ((long)d->Ho * d->Wo) % kBfCfgs[cfg].bm == 0;
  static const char* rowsw = getenv("SDM_BF16_ROW");        // "0": never, "1": wherever eligible (tests)
  // the planner sends launches of fewer than 320 128x128 tiles to the 64x64 tiles; the row-patch kernel is worth it from one
  // tile per CU on (8192 x 512, 512 channels: 487 -> 640 TFLOP/s; at 128 tiles the 64x64 tiles win, 534 vs 322)
  const bool row_mid = cfg == 1 && L.splitk == 1 && (long)cdiv(d->M, 128) * cdiv(d->N, 128) >= 256;
  if ((cfg == 0 && L.splitk == 1) || row_mid || (rowsw && rowsw[0] == '1')) {
    RowLaunch R;
    if (row_tile(d, out_f32, L.vec_epi, &R)) {      // 3x3 stride-1 geometry: one activation patch per kernel row
      const int gs = d->gn_sums != nullptr && d->N % 4 == 0;      // a tile lies inside one image by construction
      // 128 output channels at >= 2 rounds of 256-pixel tiles: an 8-wave 256x128 tile (one weight slice per 256 pixels: 1.6x less
      // L2 -> LDS traffic per flop than two 128x128 workgroups) exists behind SDM_BF16_ROW256=1 and is SLOWER (750 vs 772
      // TFLOP/s at 128 x 128x128, 761 vs 772 at 32 x 256x256, profiles/r04_row256_ab.txt): the row kernel is paced by its
      // one-barrier-per-K-step schedule, not by the feed.  Parity-tested (tests: row_patch ... forced), off by default.
      static const char* r256 = getenv("SDM_BF16_ROW256");
      const bool tall = d->N <= 128 && ((long)d->a.H * d->a.W) % 256 == 0 && d->M / 256 >= 512 && r256 && r256[0] >= '1';
      // "3" / "4": the same tile with a 3- / 4-slot weight ring, counted waits and asm fragment reads
      if (tall && r256[0] == '3') rc = launch_row<256, 128, 4, 2, 3>(d, R, gs, st);
      else if (tall && r256[0] == '4') rc = launch_row<256, 128, 4, 2, 4>(d, R, gs, st);
      else rc = tall ? launch_row<256, 128, 4, 2>(d, R, gs, st) : launch_row<128, 128, 2, 2>(d, R, gs, st);
      if (rc != SDM_OK) return rc;
      return gs ? SDM_STATS_DONE : SDM_OK;
    }
  }
  // cross-step schedule (barrier in the middle of a K-step, first fragments of the next tile read under the last MFMA
  // group, all ring slots in flight): SDM_BF16_XS=0 selects the step-boundary barrier
  static const char* xs_s = getenv("SDM_BF16_XS");
  const bool xs = fast && !(xs_s && xs_s[0] == '0');
  static const char* pp_s = getenv("SDM_BF16_PP");         // 8-wave tiles: ping-pong of the two 4-wave halves ("0": off)
  L.pp = (pp_s && pp_s[0] == '1') ? 1 : 0;
  static const char* se_s = getenv("SDM_IGEMM8_SIMPLE_EPI");   // "0": the general epilogue loops everywhere (A/B)
  L.no_simple_epi = (se_s && se_s[0] == '0') ? 1 : 0;
  if (cfg == 0) rc = fast ? (xs ? launch_bf<128, 128, 2, 2, 2, true, true>(L, ntm, st) : launch_bf<128, 128, 2, 2, 2, true>(L, ntm, st))
                          : launch_bf<128, 128, 2, 2, 2, false>(L, ntm, st);
  else if (cfg == 1) {
    // EXPERIMENT, opt-in (SDM_BF16_DEEP=1): at most one workgroup per CU in the grid (the N=1..4 sampling step) -> ring 6
    // (96 KB, five K-steps in flight instead of two).  Measured neutral on MI355X (N=1 bf16 step 0.727 -> 0.732 ms, N=4
    // 0.937 -> 0.957): these launches are not bound by tile-fetch latency per K-step but by the fixed chain of a launch
    // (kernel-argument / table / first-tile / store-drain round trips, ~7 us whatever the K walk; tools/scratch/tiny_igemm.py).
    static const char* deep = getenv("SDM_BF16_DEEP");
    const long nblocks = (long)ntm * L.ntn * d->nphase * L.splitk * d->batch;
    if (fast && nblocks <= 256 && deep && deep[0] == '1') rc = launch_bf<64, 64, 2, 2, 6, true>(L, ntm, st);
    else rc = fast ? (xs ? launch_bf<64, 64, 2, 2, 3, true, true>(L, ntm, st) : launch_bf<64, 64, 2, 2, 3, true>(L, ntm, st))
                   : launch_bf<64, 64, 2, 2, 3, false>(L, ntm, st);
  }
  else if (cfg == 2 && fast && L.vec_epi && L.splitk == 1) {
    // the phase-interleaved kernel of sdm_igemm8.hip (default); SDM_BF16_8PH=0: the one-barrier-per-K-step tile below (A/B)
    static const char* ph8_s = getenv("SDM_BF16_8PH");
    // SDM_BF16_KE=32: 32-element K-steps on a four-slot ring (three 32 KB tiles in flight instead of one 64 KB tile)
    static const char* ke_s = getenv("SDM_BF16_KE");
    if (!(ph8_s && ph8_s[0] == '0')) rc = sdm_launch_igemm8_bf16(L, ntm, st);
    else if (xs && L.pp) rc = launch_bf<256, 256, 2, 4, 2, true, true, 64, true>(L, ntm, st);
    else if (xs && ke_s && atoi(ke_s) == 32) rc = launch_bf<256, 256, 2, 4, 4, true, true, 32>(L, ntm, st);
    else rc = xs ? launch_bf<256, 256, 2, 4, 2, true, true>(L, ntm, st) : launch_bf<256, 256, 2, 4, 2, true>(L, ntm, st);
  }
  else if (cfg == 3 && fast && L.vec_epi && L.splitk == 1) rc = xs ? launch_bf<256, 256, 4, 2, 2, true, true>(L, ntm, st) : launch_bf<256, 256, 4, 2, 2, true>(L, ntm, st);
  else rc = sdm_set_error(SDM_ERR_ARG, "igemm_bf16: tile configuration %d needs the FAST path, the vector epilogue and no split-K", cfg);
  if (rc != SDM_OK) return rc;
  if (L.splitk > 1 && d->defer_epilogue) return SDM_DEFERRED;   // raw fp32 sums stay in ws for sdm_gn_apply_fused_bf16
  if (L.gn_stats) return SDM_STATS_DONE;
  if (L.splitk > 1) {
    const long out_rows = (long)(d->M / (d->Ht * d->Wt)) * d->Ho * d->Wo;
    const long long total = (long long)out_rows * d->batch * d->N;
    int grid = (int)((total + 1023) / 1024);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    SDM_LAUNCH(splitk_epilogue_bf16_kernel, dim3(grid), dim3(256), 0, st, *d, (long long)out_rows, L.out_f32);
    SDM_CHECK_LAUNCH("splitk_epilogue_bf16");
  }
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// fp32 -> bf16 (weights, once per checkpoint), round to nearest even
// --------------------------------------------------------------------------------------------
__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n, int vec) {
  if (vec) {     // 16-byte aligned: 8 elements per lane and pass (two 16-byte loads, one 16-byte store)
    const size_t n8 = n >> 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
      const f32x4 a = *(const f32x4*)(src + 8 * i), b = *(const f32x4*)(src + 8 * i + 4);
      u16x8 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) { o[k] = f2bf(a[k]); o[4 + k] = f2bf(b[k]); }
      *(u16x8*)(dst + 8 * i) = o;
    }
    for (size_t i = (n8 << 3) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = f2bf(src[i]);
    return;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = f2bf(src[i]);
}

extern "C" int sdm_cast_bf16(const float* src, uint16_t* dst, size_t n, sdm_stream_t s) {
  SDM_CHECK_ARG(src && dst && n > 0, "cast_bf16: bad args");
  const int vec = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
  size_t grid = ((vec ? n / 8 + 1 : n) + 255) / 256;
  if (grid > 8192) grid = 8192;
  SDM_LAUNCH(cast_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)s, src, dst, n, vec);
  SDM_CHECK_LAUNCH("cast_bf16");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// edge convolutions: stem fp32 NCHW image -> bf16 rows (+SiLU); head bf16 rows -> fp32 NCHW (+Tanh).  fp32 weights.
// --------------------------------------------------------------------------------------------
#define STEM_PIX_BF 32
template <int CIN>
__global__ __launch_bounds__(256) void stem_conv_bf16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                             bf16_t* __restrict__ preact, int images, int H, int W, int Cout,
                                                             int act) {
  constexpr int K = CIN * 9;
  __shared__ float patch[STEM_PIX_BF][K];
  const long long P = (long long)images * H * W;
  const long long p0 = (long long)blockIdx.x * STEM_PIX_BF;
  for (int e = threadIdx.x; e < STEM_PIX_BF * K; e += 256) {
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
  __syncthreads();
  for (int co = threadIdx.x; co < Cout; co += 256) {
    float wr[K];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) wr[ci * 9 + tap] = w[((long long)tap * Cout + co) * CIN + ci];
    const float b = bias ? bias[co] : 0.f;
    for (int pi = 0; pi < STEM_PIX_BF; ++pi) {
      const long long p = p0 + pi;
      if (p >= P) break;
      float a = b;
#pragma unroll
      for (int k = 0; k < K; ++k) a = fmaf(patch[pi][k], wr[k], a);
      if (preact) preact[p * Cout + co] = f2bf(a);
      out[p * Cout + co] = f2bf(sdm_act(a, act));
    }
  }
}

extern "C" int sdm_stem_conv_bf16(const float* x, const float* w, const float* bias, uint16_t* out, uint16_t* out_preact,
                                  int images, int Cin, int H, int W, int Cout, int act, sdm_stream_t s) {
  SDM_CHECK_ARG(x && w && out && images > 0 && H > 0 && W > 0 && Cout > 0, "stem_conv_bf16: bad args");
  SDM_CHECK_ARG(Cin >= 1 && Cin <= 8, "stem_conv_bf16: Cin %d not in [1,8]", Cin);
  const long long P = (long long)images * H * W;
  // matrix-core form (sdm_edge_bf16.hip) whenever the shape qualifies; SDM_STEM_MFMA=0 keeps the one-channel-per-thread kernel
  static const char* mf = getenv("SDM_STEM_MFMA");
  if (!(mf && mf[0] == '0')) {
    const int rc = sdm_launch_stem_mfma_bf16(x, w, bias, out, out_preact, images, Cin, H, W, Cout, act, (hipStream_t)s);
    if (rc == SDM_OK) {
      SDM_CHECK_LAUNCH("stem_conv_mfma_bf16");
      return SDM_OK;
    }
  }
  const dim3 grid(cdiv(P, STEM_PIX_BF));
#define STEM_BF(C) SDM_LAUNCH(stem_conv_bf16_kernel<C>, grid, dim3(256), 0, (hipStream_t)s, x, w, bias, out, out_preact, images, H, W, Cout, act)
  switch (Cin) {
    case 1: STEM_BF(1); break;
    case 2: STEM_BF(2); break;
    case 3: STEM_BF(3); break;
    case 4: STEM_BF(4); break;
    case 5: STEM_BF(5); break;
    case 6: STEM_BF(6); break;
    case 7: STEM_BF(7); break;
    default: STEM_BF(8); break;
  }
#undef STEM_BF
  SDM_CHECK_LAUNCH("stem_conv_bf16");
  return SDM_OK;
}

template <int COUT>
__global__ __launch_bounds__(256) void head_conv_bf16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
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
      const bf16_t* xr = x + (((long long)n * H + sy) * W + sx) * Cin;
      for (int c = lane; c < Cin; c += 64) {
        const float xv = bf2f(xr[c]);
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

extern "C" int sdm_head_conv_bf16(const uint16_t* x, const float* w, const float* bias, float* out, int images, int Cin,
                                  int H, int W, int Cout, int act, sdm_stream_t s) {
  SDM_CHECK_ARG(x && w && out && images > 0 && H > 0 && W > 0 && Cin > 0, "head_conv_bf16: bad args");
  SDM_CHECK_ARG(Cout >= 1 && Cout <= 8, "head_conv_bf16: Cout %d not in [1,8]", Cout);
  const long long P = (long long)images * H * W;
  int g = (int)((P + 3) / 4);
  if (g > 8192) g = 8192;
  const dim3 grid(g);
#define HEAD_BF(C) SDM_LAUNCH(head_conv_bf16_kernel<C>, grid, dim3(256), 0, (hipStream_t)s, x, w, bias, out, images, Cin, H, W, act)
  switch (Cout) {
    case 1: HEAD_BF(1); break;
    case 2: HEAD_BF(2); break;
    case 3: HEAD_BF(3); break;
    case 4: HEAD_BF(4); break;
    case 5: HEAD_BF(5); break;
    case 6: HEAD_BF(6); break;
    case 7: HEAD_BF(7); break;
    default: HEAD_BF(8); break;
  }
#undef HEAD_BF
  SDM_CHECK_LAUNCH("head_conv_bf16");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// GroupNorm statistics (fp32 partials, fp64 totals) of bf16 rows folded with the AdaGN modulation; same tables as
// sdm_gn_affine.  One workgroup per (image, group).
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gn_affine_bf16_kernel(const bf16_t* __restrict__ y, long long ldy, int HW, int C,
                                                              int groups, float eps, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ mod,
                                                              int mod_ns, float* __restrict__ scale, float* __restrict__ shift,
                                                              float* __restrict__ mean_o, float* __restrict__ rstd_o, int vec) {
  __shared__ double red[2][16];
  const int n = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cpg = C / groups;
  const bf16_t* base = y + (long long)n * HW * ldy + g * cpg;
  double s1 = 0.0, s2 = 0.0;
  const long long total = (long long)HW * cpg;
  if (vec) {   // cpg % 4 == 0, 8-byte aligned runs
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
          const u16x4 v = *(const u16x4*)(base + pix * ldy + c);
          const float a = bf2f(v.x), b = bf2f(v.y), cc = bf2f(v.z), dd = bf2f(v.w);
          p1 += (a + b) + (cc + dd);
          p2 += (a * a + b * b) + (cc * cc + dd * dd);
        }
      }
      s1 += p1;
      s2 += p2;
    }
  } else {
    for (long long e = threadIdx.x; e < total; e += 1024) {
      const long long pix = sdm_idiv(e, cpg);
      const int c = (int)(e - pix * cpg);
      const double v = bf2f(base[pix * ldy + c]);
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
  if (threadIdx.x == 0 && mean_o) {      // the backward of the training path needs the statistics
    mean_o[n * groups + g] = mu;
    rstd_o[n * groups + g] = rstd;
  }
  for (int c = threadIdx.x; c < cpg; c += 1024) {
    const int ch = g * cpg + c;
    const float ga = gamma[ch], be = beta[ch];
    const float sm = mod ? mod[(long long)n * mod_ns + ch] : 1.f;
    scale[(long long)n * C + ch] = rstd * ga * sm;
    shift[(long long)n * C + ch] = mod ? (be - mu * rstd * ga) * sm + sm : (be - mu * rstd * ga);
  }
}

extern "C" int sdm_gn_affine_bf16(const uint16_t* y, int64_t ldy, int images, int HW, int C, int groups, float eps,
                                  const float* gamma, const float* beta, const float* mod, int mod_nstride, float* scale,
                                  float* shift, float* mean, float* rstd, sdm_stream_t s) {
  SDM_CHECK_ARG(y && gamma && beta && scale && shift && (mean == nullptr) == (rstd == nullptr), "gn_affine_bf16: null pointer");
  SDM_CHECK_ARG(images > 0 && HW > 0 && groups > 0 && C % groups == 0, "gn_affine_bf16: C %d not divisible by groups %d", C, groups);
  const int vec = ((C / groups) % 4 == 0) && ldy % 4 == 0 && (((uintptr_t)y) & 7) == 0;
  SDM_LAUNCH(gn_affine_bf16_kernel, dim3(images * groups), dim3(1024), 0, (hipStream_t)s, y, (long long)ldy, HW, C,
                     groups, eps, gamma, beta, mod, mod_nstride, scale, shift, mean, rstd, vec);
  SDM_CHECK_LAUNCH("gn_affine_bf16");
  return SDM_OK;
}

// out = scale*y + shift (+ cat(x0,x1)) on bf16 rows; x0 may be NULL (pure affine: the next conv's input)
__global__ __launch_bounds__(256) void affine_residual_bf16_kernel(const bf16_t* __restrict__ y, long long ldy,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, int ans,
                                                                   const bf16_t* __restrict__ x0, long long ldx0,
                                                                   const bf16_t* __restrict__ x1, long long ldx1, int csplit,
                                                                   bf16_t* __restrict__ out, long long ldo, long long rows,
                                                                   int HW, int C) {
  const int C8 = C >> 3;
  const long long total = rows * C8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = sdm_idiv(e, C8);
    const int c = (int)(e - r * C8) * 8;
    const int n = (int)sdm_idiv(r, HW);
    const u16x8 v = *(const u16x8*)(y + r * ldy + c);
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = bf2f(v[i]);
    if (scale) {
      const float* a = scale + (long long)n * ans + c;
      const float* b = shift + (long long)n * ans + c;
      const f32x4 a0 = *(const f32x4*)a, a1 = *(const f32x4*)(a + 4), b0 = *(const f32x4*)b, b1 = *(const f32x4*)(b + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f[i] = f[i] * a0[i] + b0[i];
        f[4 + i] = f[4 + i] * a1[i] + b1[i];
      }
    }
    if (x0) {
      const u16x8 xv = (c < csplit) ? *(const u16x8*)(x0 + r * ldx0 + c) : *(const u16x8*)(x1 + r * ldx1 + (c - csplit));
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] += bf2f(xv[i]);
    }
    u16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f2bf(f[i]);
    *(u16x8*)(out + r * ldo + c) = o;
  }
}

extern "C" int sdm_affine_residual_bf16(const uint16_t* y, int64_t ldy, const float* scale, const float* shift,
                                        int aff_nstride, const uint16_t* x0, int64_t ldx0, const uint16_t* x1, int64_t ldx1,
                                        int csplit, uint16_t* out, int64_t ldo, int images, int HW, int C, sdm_stream_t s) {
  SDM_CHECK_ARG(y && out && images > 0 && HW > 0 && C > 0, "affine_residual_bf16: bad args");
  SDM_CHECK_ARG(x0 == nullptr || x1 != nullptr || csplit >= C, "affine_residual_bf16: csplit < C without second source");
  SDM_CHECK_ARG(C % 8 == 0 && csplit % 8 == 0 && ldy % 8 == 0 && ldo % 8 == 0 && aff_nstride % 4 == 0 && aligned16(y) &&
                    aligned16(out) && (x0 == nullptr || (aligned16(x0) && ldx0 % 8 == 0)) &&
                    (x1 == nullptr || (aligned16(x1) && ldx1 % 8 == 0)) &&
                    (scale == nullptr || (shift && aligned16(scale) && aligned16(shift))),
                "affine_residual_bf16: rows must be 16-byte aligned, C a multiple of 8");
  const long long rows = (long long)images * HW;
  const long long work = rows * (C / 8);
  int grid = (int)((work + 255) / 256);
  if (grid > 8192) grid = 8192;
  SDM_LAUNCH(affine_residual_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, y, (long long)ldy, scale, shift,
                     aff_nstride, x0, (long long)ldx0, x1, (long long)ldx1, csplit, out, (long long)ldo, rows, HW, C);
  SDM_CHECK_LAUNCH("affine_residual_bf16");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// column softmax of fp32 scores [batch][S][lds] -> bf16 probabilities [batch][S][ldp] (softmax over rows i for every
// column j, custom_layers.py:147).  Columns j in [S, ldp) are written as zero (they are reduction padding for P.V).
// --------------------------------------------------------------------------------------------
// CSM_W row phases (waves) per workgroup, lane = column: S = 4096 x 8 images is 1.3 GB of traffic per call, and with 4
// phases of one outstanding load each it streamed at 1.3 TB/s; 16 phases x 4 independent rows per lane keep 64 loads per
// column block in flight.
#define CSM_W 16
__global__ __launch_bounds__(64 * CSM_W) void col_softmax_bf16_kernel(const float* __restrict__ sc, bf16_t* __restrict__ P, int S,
                                                                      long long lds, long long ldp) {
  __shared__ float sm[CSM_W][64], ss[CSM_W][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const float* base = sc + (long long)blockIdx.y * S * lds;
  bf16_t* pb = P + (long long)blockIdx.y * S * ldp;
  float m = -INFINITY, l = 0.f;
  if (j < S) {
    int i = wave;
    for (; i + 3 * CSM_W < S; i += 4 * CSM_W) {       // four independent rows per trip: their loads overlap
      const float v0 = base[(long long)i * lds + j], v1 = base[(long long)(i + CSM_W) * lds + j];
      const float v2 = base[(long long)(i + 2 * CSM_W) * lds + j], v3 = base[(long long)(i + 3 * CSM_W) * lds + j];
      const float vm = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
      if (vm > m) {
        l *= __expf(m - vm);
        m = vm;
      }
      l += (__expf(v0 - m) + __expf(v1 - m)) + (__expf(v2 - m) + __expf(v3 - m));
    }
    for (; i < S; i += CSM_W) {
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
  for (int w = 1; w < CSM_W; ++w) M = fmaxf(M, sm[w][lane]);
  float Lsum = 0.f;
#pragma unroll
  for (int w = 0; w < CSM_W; ++w)
    if (ss[w][lane] > 0.f) Lsum += ss[w][lane] * __expf(sm[w][lane] - M);
  if (j < S) {
    const float inv = 1.f / Lsum;
    int i = wave;
    for (; i + 3 * CSM_W < S; i += 4 * CSM_W) {
      const float v0 = base[(long long)i * lds + j], v1 = base[(long long)(i + CSM_W) * lds + j];
      const float v2 = base[(long long)(i + 2 * CSM_W) * lds + j], v3 = base[(long long)(i + 3 * CSM_W) * lds + j];
      pb[(long long)i * ldp + j] = f2bf(__expf(v0 - M) * inv);
      pb[(long long)(i + CSM_W) * ldp + j] = f2bf(__expf(v1 - M) * inv);
      pb[(long long)(i + 2 * CSM_W) * ldp + j] = f2bf(__expf(v2 - M) * inv);
      pb[(long long)(i + 3 * CSM_W) * ldp + j] = f2bf(__expf(v3 - M) * inv);
    }
    for (; i < S; i += CSM_W) pb[(long long)i * ldp + j] = f2bf(__expf(base[(long long)i * lds + j] - M) * inv);
  } else if (j < ldp) {
    for (int i = wave; i < S; i += CSM_W) pb[(long long)i * ldp + j] = 0;
  }
}

extern "C" int sdm_col_softmax_bf16(const float* scores, uint16_t* P, int batch, int S, int64_t lds, int64_t ldp,
                                    sdm_stream_t s) {
  SDM_CHECK_ARG(scores && P && batch > 0 && S > 0 && lds >= S && ldp >= S, "col_softmax_bf16: bad args");
  SDM_LAUNCH(col_softmax_bf16_kernel, dim3(cdiv(ldp, 64), batch), dim3(64 * CSM_W), 0, (hipStream_t)s, scores, P, S,
                     (long long)lds, (long long)ldp);
  SDM_CHECK_LAUNCH("col_softmax_bf16");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// batched transpose of bf16 matrices: dst[b][c][r] = src[b][r][c]  (r < R, c < Cc); dst columns [R, ldd) zero-filled.
// Used for V in P.V: the matrix-core B operand must be contiguous along the reduction (token) index.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, long long lds, long long sb0,
                                                             long long sb1, int bdiv, bf16_t* __restrict__ dst, long long ldd,
                                                             long long dbs, int R, int Cc) {
  __shared__ bf16_t tile[64][66];
  const int b = blockIdx.z;
  const bf16_t* s = src + (b / bdiv) * sb0 + (b % bdiv) * sb1;
  bf16_t* dpt = dst + b * dbs;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? s[(long long)r * lds + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cc && r < ldd) dpt[(long long)c * ldd + r] = tile[tx][i];
  }
}

extern "C" int sdm_transpose_bf16(const uint16_t* src, int64_t ld_src, int64_t src_bs0, int64_t src_bs1, int bdiv,
                                  uint16_t* dst, int64_t ld_dst, int64_t dst_bs, int batch, int R, int Cc, sdm_stream_t s) {
  SDM_CHECK_ARG(src && dst && batch > 0 && bdiv > 0 && R > 0 && Cc > 0 && ld_dst >= R && ld_src >= Cc, "transpose_bf16: bad args");
  SDM_LAUNCH(transpose_bf16_kernel, dim3(cdiv(Cc, 64), cdiv(ld_dst, 64), batch), dim3(256), 0, (hipStream_t)s, src,
                     (long long)ld_src, (long long)src_bs0, (long long)src_bs1, bdiv, dst, (long long)ld_dst,
                     (long long)dst_bs, R, Cc);
  SDM_CHECK_LAUNCH("transpose_bf16");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// fp32 rows [images*HW][ld] (first C columns) -> NCHW fp32 [images][C][HW]: the edge of the matrix-core head conv
// (out_layers.1 evaluated by sdm_igemm_bf16 with fp32 output, then laid out as the reference returns it).
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_to_nchw_kernel(const float* __restrict__ rows, long long ld, float* __restrict__ out,
                                                           int C, long long HW, long long P) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
    const long long n = sdm_idiv(p, HW), rem = p - n * HW;
    for (int c = 0; c < C; ++c) out[(n * C + c) * HW + rem] = rows[p * ld + c];
  }
}

extern "C" int sdm_rows_to_nchw(const float* rows, int64_t ld, float* out_nchw, int images, int C, int H, int W, sdm_stream_t s) {
  SDM_CHECK_ARG(rows && out_nchw && images > 0 && C > 0 && H > 0 && W > 0 && ld >= C, "rows_to_nchw: bad args");
  const long long HW = (long long)H * W, P = HW * images;
  int grid = (int)((P + 255) / 256);
  if (grid > 8192) grid = 8192;
  SDM_LAUNCH(rows_to_nchw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, rows, (long long)ld, out_nchw, C, HW, P);
  SDM_CHECK_LAUNCH("rows_to_nchw");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// Fused conv-block tail on bf16 rows (no-grad forwards; see gn_apply_fused_kernel in sdm_pointwise.hip): src is either
// the conv's finished bf16 rows or the fp32 raw split-K sums left by sdm_igemm_bf16(defer_epilogue); out and the
// residual are bf16; statistics of the VALUES THE NEXT LAYER SEES (i.e. after rounding y to bf16, like the unfused path).
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 gnf16_value(const void* __restrict__ src, long long idx, int raw, float alpha, const f32x4 b,
                                             int act) {
  f32x4 v;
  if (raw) {
    v = *(const f32x4*)((const float*)src + idx);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = bf2f(f2bf(sdm_act(v[i] * alpha + b[i], act)));
  } else {
    const u16x4 u = *(const u16x4*)((const bf16_t*)src + idx);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = bf2f(u[i]);
  }
  return v;
}

__global__ __launch_bounds__(1024) void gn_apply_fused_bf16_kernel(void* __restrict__ src, long long lds_, int raw, float alpha,
                                                                   const float* __restrict__ bias, int act, int HW, int C,
                                                                   int groups, float eps, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta,
                                                                   const float* __restrict__ mod, int mod_ns,
                                                                   const bf16_t* __restrict__ x0, long long ldx0,
                                                                   const bf16_t* __restrict__ x1, long long ldx1, int csplit,
                                                                   bf16_t* __restrict__ out, long long ldo, int hold) {
  __shared__ double red[2][16];
  __shared__ float s_stat[2];
  const int n = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cpg = C / groups, q4 = cpg >> 2;
  const long long base = (long long)n * HW * lds_ + g * cpg;
  const long long tot4 = (long long)HW * q4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  double s1 = 0.0, s2 = 0.0;
  const bool held = hold && tot4 <= 1024 * 8;      // small maps: the values stay in registers between the two sweeps (see gn_apply_fused_kernel)
  f32x4 hv[8];
  if (held) {
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long e = threadIdx.x + (long long)u * 1024;
      hv[u] = zero4;
      if (e < tot4) {
        const long long pix = sdm_idiv(e, q4);
        const int c = (int)(e - pix * q4) * 4;
        const f32x4 b = (raw && bias) ? *(const f32x4*)(bias + g * cpg + c) : zero4;
        hv[u] = gnf16_value(src, base + pix * lds_ + c, raw, alpha, b, act);
        p1 += (hv[u].x + hv[u].y) + (hv[u].z + hv[u].w);
        p2 += (hv[u].x * hv[u].x + hv[u].y * hv[u].y) + (hv[u].z * hv[u].z + hv[u].w * hv[u].w);
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
        const f32x4 v = gnf16_value(src, base + pix * lds_ + c, raw, alpha, b, act);
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
  }
  __syncthreads();
  const float mu = s_stat[0], rstd = s_stat[1];
  auto emit = [&](long long e, const f32x4 v) {
    const long long pix = sdm_idiv(e, q4);
    const int c = (int)(e - pix * q4) * 4;
    const int ch = g * cpg + c;
    const long long si = base + pix * lds_ + c;
    if (raw) *(f32x4*)((float*)src + si) = zero4;        // the split-K scratch goes back zero-filled
    const f32x4 ga = *(const f32x4*)(gamma + ch), be = *(const f32x4*)(beta + ch);
    f32x4 o;
    if (mod) {
      const f32x4 sm = *(const f32x4*)(mod + (long long)n * mod_ns + ch);
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = v[i] * (rstd * ga[i] * sm[i]) + ((be[i] - mu * rstd * ga[i]) * sm[i] + sm[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = v[i] * (rstd * ga[i]) + (be[i] - mu * rstd * ga[i]);
    }
    const long long row = (long long)n * HW + pix;
    if (x0) {
      const u16x4 xv = (ch < csplit) ? *(const u16x4*)(x0 + row * ldx0 + ch) : *(const u16x4*)(x1 + row * ldx1 + (ch - csplit));
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] += bf2f(xv[i]);
    }
    u16x4 ov;
#pragma unroll
    for (int i = 0; i < 4; ++i) ov[i] = f2bf(o[i]);
    *(u16x4*)(out + row * ldo + ch) = ov;
  };
  if (held) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long e = threadIdx.x + (long long)u * 1024;
      if (e < tot4) emit(e, hv[u]);
    }
  } else {
    for (long long e = threadIdx.x; e < tot4; e += 1024) {
      const long long pix = sdm_idiv(e, q4);
      const int c = (int)(e - pix * q4) * 4;
      const f32x4 b = (raw && bias) ? *(const f32x4*)(bias + g * cpg + c) : zero4;
      emit(e, gnf16_value(src, base + pix * lds_ + c, raw, alpha, b, act));
    }
  }
}

extern "C" int sdm_gn_apply_fused_bf16(const void* src, int64_t ld_src, int src_raw, float alpha, const float* bias, int act,
                                       int images, int HW, int C, int groups, float eps, const float* gamma,
                                       const float* beta, const float* mod, int mod_nstride, const uint16_t* x0,
                                       int64_t ldx0, const uint16_t* x1, int64_t ldx1, int csplit, uint16_t* out, int64_t ldo,
                                       sdm_stream_t s) {
  SDM_CHECK_ARG(src && gamma && beta && out && images > 0 && HW > 0 && groups > 0 && C % groups == 0, "gn_apply_fused_bf16: bad args");
  SDM_CHECK_ARG(x0 == nullptr || x1 != nullptr || csplit >= C, "gn_apply_fused_bf16: csplit < C without second source");
  const int cpg = C / groups;
  auto al = [](const void* q, size_t a) { return (((uintptr_t)q) & (a - 1)) == 0; };
  SDM_CHECK_ARG(cpg % 4 == 0 && ld_src % 4 == 0 && ldo % 4 == 0 && al(src, src_raw ? 16 : 8) && al(out, 8) && aligned16(gamma) &&
                    aligned16(beta) && (mod == nullptr || (aligned16(mod) && mod_nstride % 4 == 0)) &&
                    (bias == nullptr || aligned16(bias)) && (x0 == nullptr || (al(x0, 8) && ldx0 % 4 == 0 && csplit % 4 == 0)) &&
                    (x1 == nullptr || (al(x1, 8) && ldx1 % 4 == 0)),
                "gn_apply_fused_bf16: needs channels-per-group %% 4 == 0 and aligned rows / tables");
  SDM_CHECK_ARG(act >= 0 && act <= 2, "gn_apply_fused_bf16: bad act");
  SDM_LAUNCH(gn_apply_fused_bf16_kernel, dim3(images * groups), dim3(1024), 0, (hipStream_t)s, (void*)src,
                     (long long)ld_src, src_raw, alpha, bias, act, HW, C, groups, eps, gamma, beta, mod, mod_nstride,
                     (const bf16_t*)x0, (long long)ldx0, (const bf16_t*)x1, (long long)ldx1, csplit, (bf16_t*)out, (long long)ldo,
                     sdm_gn_hold());
  SDM_CHECK_LAUNCH("gn_apply_fused_bf16");
  return SDM_OK;
}


// --------------------------------------------------------------------------------------------
// GroupNorm fold from per-(image, channel) sums emitted by the producing conv's epilogue (sdm_igemm_bf16 with gn_sums):
// the statistics need no pass over the activation any more.  One wave per (image, group); clears the sums it consumed
// (zero-filled contract, like the split-K scratch).  Same tables as sdm_gn_affine.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void gn_finalize_kernel(float* __restrict__ sums, int HW, int C, int groups, float eps,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mod, int mod_ns, float* __restrict__ scale,
                                                         float* __restrict__ shift, float* __restrict__ mean_o,
                                                         float* __restrict__ rstd_o) {
  const int n = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cpg = C / groups;
  double s1 = 0.0, s2 = 0.0;
  for (int c = threadIdx.x; c < cpg; c += 64) {
    float* p = sums + ((long long)n * C + g * cpg + c) * 2;
    s1 += p[0];
    s2 += p[1];
    p[0] = 0.f;
    p[1] = 0.f;
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  const double total = (double)HW * cpg;
  const double mean = s1 / total;
  double var = s2 / total - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float mu = (float)mean;
  if (threadIdx.x == 0 && mean_o) {
    mean_o[n * groups + g] = mu;
    rstd_o[n * groups + g] = rstd;
  }
  for (int c = threadIdx.x; c < cpg; c += 64) {
    const int ch = g * cpg + c;
    const float ga = gamma[ch], be = beta[ch];
    const float sm = mod ? mod[(long long)n * mod_ns + ch] : 1.f;
    scale[(long long)n * C + ch] = rstd * ga * sm;
    shift[(long long)n * C + ch] = mod ? (be - mu * rstd * ga) * sm + sm : (be - mu * rstd * ga);
  }
}

extern "C" int sdm_gn_finalize(float* sums, int images, int HW, int C, int groups, float eps, const float* gamma,
                               const float* beta, const float* mod, int mod_nstride, float* scale, float* shift, float* mean,
                               float* rstd, sdm_stream_t s) {
  SDM_CHECK_ARG(sums && gamma && beta && scale && shift && (mean == nullptr) == (rstd == nullptr), "gn_finalize: null pointer");
  SDM_CHECK_ARG(images > 0 && HW > 0 && groups > 0 && C % groups == 0, "gn_finalize: C %d not divisible by groups %d", C, groups);
  SDM_LAUNCH(gn_finalize_kernel, dim3(images * groups), dim3(64), 0, (hipStream_t)s, sums, HW, C, groups, eps, gamma,
                     beta, mod, mod_nstride, scale, shift, mean, rstd);
  SDM_CHECK_LAUNCH("gn_finalize");
  return SDM_OK;
}
