// Implicit-GEMM contraction kernels on the gfx950 fp32 matrix cores.
//
//   v_mfma_f32_32x32x2_f32:  lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31];
//   the 32x32 result has col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
//
// Activations are NHWC "pixel rows x channels", so a conv tap is a shifted row gather with the
// reduction (channel) dimension contiguous; the weight slice of a tap is a plain [c_out][c_in]
// matrix.  Both operand tiles go global -> registers (16-byte loads, with the pending
// GroupNorm/AdaGN affine applied in registers) -> LDS, double-buffered; fragments are read
// with ds_read_b128 from rows padded to 36 floats (conflict-free) or, for operands whose
// reduction index is the row index, with ds_read_b32 down the columns (also conflict-free).
#include <math.h>
#include <stdlib.h>

#include "sdm_common.h"

#define BK 32
#define PITCH 36

struct IgemmLaunch {
  sdm_igemm_desc d;
  int splitk, ntn, kchunks;
  int mfast;   // tile order inside one XCD's run: 0 = N-tiles fastest (A rows shared in L2), 1 = M-tiles fastest (B shared)
  long long out_rows;   // output rows of ONE batch entry (split-K scratch is [batch][out_rows][N])
  // prologue shortcuts (host-derived, wave-uniform): every feature map of the U-Net has power-of-two sides, and most
  // launches write their rows in tile order -- the general per-row divisions cost ~350 instructions of a ~1800-instruction
  // prologue, a measurable part of the ~7 us a launch of a few K-steps takes (tools/scratch/tiny_igemm.py)
  int pow2, lw, lhw;    // Wt == 1 << lw and Ht * Wt == 1 << lhw
  int plain_out;        // output row == tile-grid row (one phase, unit output stride, Ho x Wo == Ht x Wt)
};

static inline int ilog2_exact(long v) {   // log2 of a power of two, else -1
  if (v <= 0 || (v & (v - 1)) != 0) return -1;
  int l = 0;
  while ((1L << l) < v) ++l;
  return l;
}
static inline void igemm_geometry(IgemmLaunch& L) {
  const sdm_igemm_desc& d = L.d;
  L.lw = ilog2_exact(d.Wt);
  L.lhw = ilog2_exact((long)d.Ht * d.Wt);
  L.pow2 = L.lw >= 0 && L.lhw >= 0;
  L.plain_out = d.nphase == 1 && d.ostride == 1 && d.Ho == d.Ht && d.Wo == d.Wt && d.ph_oy[0] == 0 && d.ph_ox[0] == 0;
}

// Workgroups go to the 8 XCDs round-robin by linear id and every XCD has its own L2.  Remap blockIdx.x so that one
// XCD owns a contiguous run of the tile order: tiles that share an operand then meet in one L2 instead of each
// XCD fetching its own copy from HBM.  (Bijective for any grid size; the grid's y/z dimensions -- taps, split-K
// slices, images -- keep their XCD when gridDim.x is a multiple of 8, which the planner's tile counts mostly are.)
__device__ __forceinline__ int xcd_contiguous(int bid, int total) {
  const int q = total >> 3, r = total & 7, x = bid & 7, i = bid >> 3;
  return x < r ? x * (q + 1) + i : r * (q + 1) + (x - r) * q + i;
}

// One K-step = (tap, 32-channel chunk).  Per-tap state (row pointers, validity) is rebuilt only when the tap
// changes; inside a tap the loads are pointer + running channel offset, so the steady-state loop is
// [LDS store | barrier | issue next loads | 4 x (fragment reads + MFMAs)].
template <int BM, int BN, int WM, int WN, bool BKM, bool VEC, bool AFF>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmLaunch L) {
  const sdm_igemm_desc& d = L.d;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int AIT = BM / 32, BIT = BN / 32;
  constexpr int A_TILE = BM * PITCH;
  constexpr int B_TILE = BKM ? BK * BN : BN * PITCH;
  constexpr int CPR = BN / 4;     // K-major B: 16-byte chunks per k-row
  constexpr int RPP = 256 / CPR;  // rows per pass
  extern __shared__ __align__(16) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_TILE;
  long long* s_orow = (long long*)(Bs + 2 * B_TILE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, h = lane >> 5, l31 = lane & 31;
  const int tile_id = L.mfast == 2 ? (int)blockIdx.x : xcd_contiguous(blockIdx.x, gridDim.x), ntm_ = gridDim.x / L.ntn;
  const int tile_n = L.mfast == 1 ? tile_id / ntm_ : tile_id % L.ntn, tile_m = L.mfast == 1 ? tile_id % ntm_ : tile_id / L.ntn;
  const int phase = blockIdx.y / L.splitk, ks = blockIdx.y % L.splitk;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int b0 = blockIdx.z / d.bdiv, b1 = blockIdx.z % d.bdiv;
  const float* a0 = d.a.p0 + b0 * d.a_bs0 + b1 * d.a_bs1;
  const float* a1 = d.a.p1;
  const float* bw = d.b + b0 * d.b_bs0 + b1 * d.b_bs1;
  const int HtWt = d.Ht * d.Wt;
  const int K = d.K, N = d.N, csplit = d.a.csplit;
  const int cc4 = 4 * (tid & 7);           // channel offset of this thread's 16-byte chunk inside a K-step
  const int arow = tid >> 3;               // tile row (A, and B when K-contiguous) handled in pass 0

  for (int r = tid; r < BM; r += 256) {
    int m = m0 + r;
    long long o = -1;
    if (m < d.M) {
      int n = m / HtWt, rem = m - n * HtWt;
      int y = rem / d.Wt, x = rem - y * d.Wt;
      int oy = y * d.ostride + d.ph_oy[phase], ox = x * d.ostride + d.ph_ox[phase];
      if (oy < d.Ho && ox < d.Wo) o = ((long long)n * d.Ho + oy) * d.Wo + ox;
    }
    s_orow[r] = o;
  }

  int a_ys[AIT], a_xs[AIT], a_n[AIT];
  long long a_base[AIT];
#pragma unroll
  for (int i = 0; i < AIT; ++i) {
    int m = m0 + arow + 32 * i;
    if (m < d.M) {
      int n = m / HtWt, rem = m - n * HtWt;
      int y = rem / d.Wt, x = rem - y * d.Wt;
      a_n[i] = n;
      a_ys[i] = y * d.sstride;
      a_xs[i] = x * d.sstride;
      a_base[i] = ((long long)n * d.a.H + a_ys[i]) * d.a.W + a_xs[i];
    } else {
      a_n[i] = 0; a_ys[i] = -(1 << 20); a_xs[i] = 0; a_base[i] = 0;
    }
  }

  const int ntap = d.ntaps[phase];
  const int kch = L.kchunks;
  const int T = ntap * kch;
  const int it0 = (int)((long long)T * ks / L.splitk), it1 = (int)((long long)T * (ks + 1) / L.splitk);

  // ---- per-tap state
  const float* pa0[AIT];
  const float* pa1[AIT];
  bool aok[AIT];
  const float* pb[BIT];
  bool bok[BIT];
  auto setup_tap = [&](int tp) {
    const int dy = d.tap_dy[phase][tp], dx = d.tap_dx[phase][tp];
    const float* bt = bw + (long long)d.tap_w[phase][tp] * d.b_tap_stride;
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const int sy = a_ys[i] + dy, sx = a_xs[i] + dx;
      aok[i] = (unsigned)sy < (unsigned)d.a.H && (unsigned)sx < (unsigned)d.a.W;
      const long long row = a_base[i] + dy * d.a.W + dx;
      pa0[i] = a0 + row * d.a.ld0 + cc4;
      pa1[i] = a1 + row * d.a.ld1 + (cc4 - csplit);
    }
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < BIT; ++i) {
        const int n = n0 + arow + 32 * i;
        bok[i] = n < N;
        pb[i] = bt + (long long)n * d.ldb + cc4;
      }
    } else {
#pragma unroll
      for (int i = 0; i < BIT; ++i) {
        const int n = n0 + 4 * (tid % CPR);
        bok[i] = n < N;
        pb[i] = bt + (long long)(tid / CPR + RPP * i) * d.ldb + n;
      }
    }
  };

  f32x4 ra[AIT], rb[BIT];
  f32x4 rsc[AFF ? AIT : 1], rsh[AFF ? AIT : 1];
  unsigned amask = 0, bmask = 0;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // load() only ISSUES the global loads of a K-step (VEC path: branch-free, an invalid chunk reads an always-mapped
  // dummy address); the zero-select and the pending affine are applied in store(), one K-step later, so the loads
  // stay in flight under the MFMAs of the current step.
  auto load = [&](int c0) {
    const int c = c0 + cc4;
    amask = 0;
    bmask = 0;
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const bool use = aok[i] && c < K;
      amask |= use ? (1u << i) : 0u;
      if (VEC) {
        const float* p = (c < csplit ? pa0[i] : pa1[i]) + c0;
        p = use ? p : d.a.p0;
        ra[i] = *(const f32x4*)p;
        if (AFF) {  // host guarantees K % 4 == 0 and aligned tables
          const long long so = use ? (long long)a_n[i] * d.a.aff_nstride + c : 0;
          rsc[i] = *(const f32x4*)(d.a.scale + so);
          rsh[i] = *(const f32x4*)(d.a.shift + so);
        }
      } else {
        f32x4 v = zero4;
        if (use) {
          const float* p = (c < csplit ? pa0[i] : pa1[i]) + c0;
          v.x = p[0];
          if (c + 1 < K) v.y = p[1];
          if (c + 2 < K) v.z = p[2];
          if (c + 3 < K) v.w = p[3];
        }
        ra[i] = v;
        if (AFF) {
          const long long so = use ? (long long)a_n[i] * d.a.aff_nstride + c : 0;
          rsc[i] = *(const f32x4*)(d.a.scale + so);
          rsh[i] = *(const f32x4*)(d.a.shift + so);
        }
      }
    }
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < BIT; ++i) {
        const bool use = bok[i] && c < K;
        bmask |= use ? (1u << i) : 0u;
        if (VEC) {
          const float* p = use ? pb[i] + c0 : d.b;
          rb[i] = *(const f32x4*)p;
        } else {
          f32x4 v = zero4;
          if (use) {
            const float* p = pb[i] + c0;
            v.x = p[0];
            if (c + 1 < K) v.y = p[1];
            if (c + 2 < K) v.z = p[2];
            if (c + 3 < K) v.w = p[3];
          }
          rb[i] = v;
        }
      }
    } else {
      const long long koff = (long long)c0 * d.ldb;
#pragma unroll
      for (int i = 0; i < BIT; ++i) {
        const int kk = c0 + tid / CPR + RPP * i;
        const bool use = bok[i] && kk < K;
        bmask |= use ? (1u << i) : 0u;
        if (VEC) {
          const float* p = use ? pb[i] + koff : d.b;
          rb[i] = *(const f32x4*)p;
        } else {
          f32x4 v = zero4;
          if (use) {
            const float* p = pb[i] + koff;
            const int n = n0 + 4 * (tid % CPR);
            v.x = p[0];
            if (n + 1 < N) v.y = p[1];
            if (n + 2 < N) v.z = p[2];
            if (n + 3 < N) v.w = p[3];
          }
          rb[i] = v;
        }
      }
    }
  };

  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      f32x4 v = ra[i];
      if (AFF) v = v * rsc[i] + rsh[i];
      v = (amask >> i) & 1u ? v : zero4;           // out-of-image taps stay exactly zero AFTER the affine
      *(f32x4*)&As[buf * A_TILE + (arow + 32 * i) * PITCH + cc4] = v;
    }
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < BIT; ++i)
        *(f32x4*)&Bs[buf * B_TILE + (arow + 32 * i) * PITCH + cc4] = (bmask >> i) & 1u ? rb[i] : zero4;
    } else {
#pragma unroll
      for (int i = 0; i < BIT; ++i)
        *(f32x4*)&Bs[buf * B_TILE + (tid / CPR + RPP * i) * BN + 4 * (tid % CPR)] = (bmask >> i) & 1u ? rb[i] : zero4;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Software pipeline (one barrier per K-step):
  //   registers hold tile it+1, LDS[buf] holds tile it.
  //   first-half MFMAs | store tile it+1 -> LDS[buf^1] | issue loads of tile it+2 | second-half MFMAs | barrier
  // so the LDS stores and the next global loads sit BETWEEN matrix instructions of the same wave.
  auto mma_half = [&](int buf, int half) {
    const float* Ab = As + buf * A_TILE + (wm * TM * 32 + l31) * PITCH + h * 4;
    const float* Bb = BKM ? Bs + buf * B_TILE + (h * 4) * BN + wn * TN * 32 + l31
                          : Bs + buf * B_TILE + (wn * TN * 32 + l31) * PITCH + h * 4;
#pragma unroll
    for (int kq = 0; kq < BK / 16; ++kq) {
      const int kg = half * (BK / 16) + kq;
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(Ab + i * 32 * PITCH + kg * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!BKM) {
          bf[j] = *(const f32x4*)(Bb + j * 32 * PITCH + kg * 8);
        } else {
          bf[j].x = Bb[(kg * 8 + 0) * BN + j * 32];
          bf[j].y = Bb[(kg * 8 + 1) * BN + j * 32];
          bf[j].z = Bb[(kg * 8 + 2) * BN + j * 32];
          bf[j].w = Bb[(kg * 8 + 3) * BN + j * 32];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
  };

  int tp = it0 / kch, kc = it0 - tp * kch;
  auto advance_and_load = [&](int next_it) {
    if (next_it < it1) {
      if (++kc == kch) {
        kc = 0;
        setup_tap(++tp);
      }
      load(kc * BK);
    } else {
      amask = 0;
      bmask = 0;
    }
  };
  if (it0 < it1) {
    setup_tap(tp);
    load(kc * BK);
    store(0);
    advance_and_load(it0 + 1);
  }
  __syncthreads();
  for (int it = it0; it < it1; ++it) {
    const int buf = (it - it0) & 1;
    mma_half(buf, 0);
    store(buf ^ 1);
    advance_and_load(it + 2);
    mma_half(buf, 1);
    __syncthreads();
  }

  // ---- epilogue
  const long long obo = b0 * d.o_bs0 + b1 * d.o_bs1;
  const long long rbo = b0 * d.r_bs0 + b1 * d.r_bs1;
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
        if (d.out_preact) d.out_preact[obo + orow * d.ldo + col] = v;
        v = sdm_act(v, d.act);
        if (d.res) v += d.res[rbo + orow * d.ldres + col];
        if (d.res2) v += d.res2[orow * d.ldres2 + col];
        if (d.dact_z && (d.dact_ncols == 0 || col < d.dact_ncols)) v *= sdm_dsilu(d.dact_z[orow * d.ldz + col]);
        d.out[obo + orow * d.ldo + col] = v;
      }
    }
  }
}

// --------------------------------------------------------------------------------------------
// LDS-DMA variant (no pending affine, 16-byte-aligned operands): operand tiles go global -> LDS directly with
// global_load_lds_dwordx4 (no VGPR round trip, no select, no ds_write).  One wave-instruction writes 1 KiB
// lane-linearly, i.e. 8 rows of 32 floats; rows are therefore unpadded and bank conflicts are avoided by an XOR
// swizzle of the 16-byte chunk index with (row>>1)&7, applied to the per-lane SOURCE address and again on the
// fragment read.  Out-of-image taps / rows read a 16-byte zero word instead.  NBUF LDS buffers form a ring: the
// tile of K-step it+NBUF-1 is issued right after the barrier of step `it`, a counted s_waitcnt keeps the younger
// tiles in flight across the barrier (one raw s_barrier per K-step).
// --------------------------------------------------------------------------------------------
__device__ float g_zero_chunk[4];

// timing build only (-DSDM_IGEMM8_TIMING, see sdm_igemm8.hip): phase stamps of igemm_glds_kernel's workgroups (fp32)
#ifdef SDM_IGEMM8_TIMING
#define TF_MAXWG 65536
__device__ unsigned long long g_tf[TF_MAXWG][4];
#define TF_STAMP(i)                                                                                                            \
  do {                                                                                                                         \
    const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                                       \
    if (threadIdx.x == 0 && wg_ < TF_MAXWG) g_tf[wg_][i] = __builtin_amdgcn_s_memrealtime();                                    \
  } while (0)
extern "C" int sdm_debug_f32_clear() {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tf)) != hipSuccess) return -1;
  return hipMemset(p, 0, sizeof(g_tf)) == hipSuccess ? 0 : -1;
}
extern "C" int sdm_debug_f32_times(unsigned long long* host, int nwg) {
  if (nwg > TF_MAXWG) nwg = TF_MAXWG;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tf), (size_t)nwg * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#else
#define TF_STAMP(i) do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int WM, int WN, bool BKM, int NBUF, int BKT>
__global__ __launch_bounds__(256) void igemm_glds_kernel(const IgemmLaunch L) {
  const sdm_igemm_desc& d = L.d;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int CA = BKT / 4;                     // 16-byte chunks per K-contiguous tile row
  constexpr int RA = 256 / CA;                    // tile rows covered by one pass of the 256 threads
  constexpr int AIT = BM / RA;                    // passes for the A tile
  constexpr int CPR = BN / 4, RPP = 256 / CPR;    // K-major B: chunks per k-row, k-rows per pass
  constexpr int BIT = BKM ? BKT / RPP : BN / RA;  // passes for the B tile
  constexpr int A_TILE = BM * BKT;
  constexpr int B_TILE = BN * BKT;
  constexpr int G = AIT + BIT;      // LDS-DMA instructions per thread per K-step
  constexpr int PD = NBUF - 1;      // prefetch distance in K-steps
  extern __shared__ __align__(16) float smem[];
  float* As = smem;
  float* Bs = smem + NBUF * A_TILE;
  long long* s_orow = (long long*)(Bs + NBUF * B_TILE);
  int* s_tap = (int*)(s_orow + BM);
  TF_STAMP(0);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, h = lane >> 5, l31 = lane & 31;
  const int tile_id = L.mfast == 2 ? (int)blockIdx.x : xcd_contiguous(blockIdx.x, gridDim.x), ntm_ = gridDim.x / L.ntn;
  const int tile_n = L.mfast == 1 ? tile_id / ntm_ : tile_id % L.ntn, tile_m = L.mfast == 1 ? tile_id % ntm_ : tile_id / L.ntn;
  const int phase = blockIdx.y / L.splitk, ks = blockIdx.y % L.splitk;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int b0 = blockIdx.z / d.bdiv, b1 = blockIdx.z % d.bdiv;
  const float* a0 = d.a.p0 + b0 * d.a_bs0 + b1 * d.a_bs1;
  const float* a1 = d.a.p1;
  const float* bw = d.b + b0 * d.b_bs0 + b1 * d.b_bs1;
  const int HtWt = d.Ht * d.Wt;
  const int K = d.K, N = d.N, csplit = d.a.csplit;
  const int arow = tid / CA;
  // logical 16-byte chunk this lane fetches for its physical slot (tid % CA): slot ^ key(row); key = (row>>1)&7 for
  // 128-byte rows (BKT 32), row&15 for 256-byte rows (BKT 64); rows arow + RA*i share it
  const int cc4 = 4 * ((tid % CA) ^ (BKT == 32 ? ((arow >> 1) & 7) : (arow & 15)));
  const int sw_l = BKT == 32 ? ((l31 >> 1) & 7) : (l31 & 15);  // swizzle key of the fragment rows this lane reads
  const float* zero = g_zero_chunk;

  for (int r = tid; r < BM; r += 256) {
    int m = m0 + r;
    long long o = -1;
    if (m < d.M) {
      if (L.plain_out) {
        o = m;
      } else {
        int n, y, x;
        if (L.pow2) { n = m >> L.lhw; const int rem = m & (HtWt - 1); y = rem >> L.lw; x = rem & (d.Wt - 1); }
        else { n = m / HtWt; const int rem = m - n * HtWt; y = rem / d.Wt; x = rem - y * d.Wt; }
        int oy = y * d.ostride + d.ph_oy[phase], ox = x * d.ostride + d.ph_ox[phase];
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
    int m = m0 + arow + RA * i;
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
  TF_STAMP(1);
  __syncthreads();   // s_orow / s_tap visible

  const int ntap = d.ntaps[phase];
  const int kch = L.kchunks;
  const int T = ntap * kch;
  const int it0 = (int)((long long)T * ks / L.splitk), it1 = (int)((long long)T * (ks + 1) / L.splitk);

  const float* pa0[AIT];
  const float* pa1[AIT];
  bool aok[AIT];
  const float* pb[BIT];
  bool bok[BIT];
  auto setup_tap = [&](int tp) {
    const int dy = s_tap[tp], dx = s_tap[SDM_MAX_TAPS + tp];
    const float* bt = bw + (long long)s_tap[2 * SDM_MAX_TAPS + tp] * d.b_tap_stride;
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const int sy = a_ys[i] + dy, sx = a_xs[i] + dx;
      aok[i] = (unsigned)sy < (unsigned)d.a.H && (unsigned)sx < (unsigned)d.a.W;
      const long long row = a_base[i] + dy * d.a.W + dx;
      pa0[i] = a0 + row * d.a.ld0 + cc4;
      pa1[i] = a1 + row * d.a.ld1 + (cc4 - csplit);
    }
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < BIT; ++i) {
        const int n = n0 + arow + RA * i;
        bok[i] = n < N;
        pb[i] = bt + (long long)n * d.ldb + cc4;
      }
    } else {
#pragma unroll
      for (int i = 0; i < BIT; ++i) {
        const int n = n0 + 4 * (tid % CPR);
        bok[i] = n < N;
        pb[i] = bt + (long long)(tid / CPR + RPP * i) * d.ldb + n;
      }
    }
  };

  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  // issue the DMA of one K-step (tap state already set) into ring slot `buf`
  auto issue = [&](int c0, int buf) {
    const int c = c0 + cc4;
    float* Ad = As + buf * A_TILE + __builtin_amdgcn_readfirstlane(wave) * 256;
    float* Bd = Bs + buf * B_TILE + __builtin_amdgcn_readfirstlane(wave) * 256;
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const bool use = aok[i] && c < K;
      const float* p = (c < csplit ? pa0[i] : pa1[i]) + c0;
      p = use ? p : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr)p, (lds_ptr)(Ad + i * RA * BKT), 16, 0, 0);
    }
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < BIT; ++i) {
        const bool use = bok[i] && c < K;
        const float* p = use ? pb[i] + c0 : zero;
        __builtin_amdgcn_global_load_lds((glb_ptr)p, (lds_ptr)(Bd + i * RA * BKT), 16, 0, 0);
      }
    } else {
      const long long koff = (long long)c0 * d.ldb;
#pragma unroll
      for (int i = 0; i < BIT; ++i) {
        const int kk = c0 + tid / CPR + RPP * i;
        const bool use = bok[i] && kk < K;
        const float* p = use ? pb[i] + koff : zero;
        __builtin_amdgcn_global_load_lds((glb_ptr)p, (lds_ptr)(Bd + i * RPP * BN), 16, 0, 0);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // issue-side K-step cursor
  int tp = it0 / kch, kc = it0 - tp * kch;
  int issued = it0;
  auto issue_next = [&]() {   // issues K-step `issued` (if any) into its ring slot and advances the cursor
    if (issued < it1) {
      if (issued == it0) {
        setup_tap(tp);
      } else if (++kc == kch) {
        kc = 0;
        setup_tap(++tp);
      }
      issue(kc * BKT, (issued - it0) % NBUF);
    }
    ++issued;
  };
#pragma unroll
  for (int p = 0; p < PD; ++p) issue_next();

  for (int it = it0; it < it1; ++it) {
    const int buf = (it - it0) % NBUF;
    // tiles issued after `it` may stay in flight; everything older must have landed
    const int younger = (it1 - 1 - it) < (PD - 1) ? (it1 - 1 - it) : (PD - 1);
    if (younger >= 2) wait_vmcnt<2 * G>();
    else if (younger == 1) wait_vmcnt<G>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // tile `it` visible to all waves; ring slot of tile it-1 is free again
    issue_next();                   // K-step it+PD
    const float* Ab = As + buf * A_TILE + (wm * TM * 32 + l31) * BKT;
    const float* Bb = BKM ? Bs + buf * B_TILE + (h * 4) * BN + wn * TN * 32 + l31
                          : Bs + buf * B_TILE + (wn * TN * 32 + l31) * BKT;
#pragma unroll
    for (int kg = 0; kg < BKT / 8; ++kg) {
      const int ph = ((kg * 2 + h) ^ sw_l) * 4;   // swizzled chunk of this lane's fragment rows
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(Ab + i * 32 * BKT + ph);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!BKM) {
          bf[j] = *(const f32x4*)(Bb + j * 32 * BKT + ph);
        } else {
          bf[j].x = Bb[(kg * 8 + 0) * BN + j * 32];
          bf[j].y = Bb[(kg * 8 + 1) * BN + j * 32];
          bf[j].z = Bb[(kg * 8 + 2) * BN + j * 32];
          bf[j].w = Bb[(kg * 8 + 3) * BN + j * 32];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue (same contract as igemm_kernel)
  TF_STAMP(2);
  if (L.splitk > 1) {       // partial sums: nothing but the atomics
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
    TF_STAMP(3);
    return;
  }
  const long long obo = b0 * d.o_bs0 + b1 * d.o_bs1;
  const long long rbo = b0 * d.r_bs0 + b1 * d.r_bs1;
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
        if (d.out_preact) d.out_preact[obo + orow * d.ldo + col] = v;
        v = sdm_act(v, d.act);
        if (d.res) v += d.res[rbo + orow * d.ldres + col];
        if (d.res2) v += d.res2[orow * d.ldres2 + col];
        if (d.dact_z && (d.dact_ncols == 0 || col < d.dact_ncols)) v *= sdm_dsilu(d.dact_z[orow * d.ldz + col]);
        d.out[obo + orow * d.ldo + col] = v;
      }
    }
  }
  TF_STAMP(3);
}

template <int BM, int BN, int WM, int WN, bool BKM, int NBUF, int BKT>
static int launch_igemm_glds(const IgemmLaunch& L, int ntm, hipStream_t st) {
  constexpr size_t lds = (size_t)NBUF * (BM + BN) * BKT * sizeof(float) + BM * sizeof(long long) + 3 * SDM_MAX_TAPS * sizeof(int);
  static bool attr_set = false;
  auto kern = igemm_glds_kernel<BM, BN, WM, WN, BKM, NBUF, BKT>;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid(ntm * L.ntn, L.d.nphase * L.splitk, L.d.batch);
  double taps = 0;
  for (int p = 0; p < L.d.nphase; ++p) taps += L.d.ntaps[p];
  {  // algorithmic bytes: every input row, weight and output element once (SURVEY 8d rule)
    const double images = (double)L.d.M / ((double)L.d.Ht * L.d.Wt);
    const double in_b = images * L.d.a.H * L.d.a.W * L.d.K, w_b = (double)L.d.K * L.d.N * taps, out_b = images * L.d.Ho * L.d.Wo * L.d.N;
    sdm_prof_shape(L.d.M, L.d.N, L.d.K, (int)taps, L.splitk, 4.0 * L.d.batch * (in_b + w_b + out_b));
  }
  sdm_prof_pre(st);
  SDM_LAUNCH(kern, grid, dim3(256), lds, st, L);
  sdm_prof_post(st, BM == 128 ? (BN == 128 ? 0 : 5) : (BM == 64 ? 1 : 2), 2.0 * L.d.M * L.d.N * L.d.K * taps * L.d.batch);
  SDM_CHECK_LAUNCH("igemm_glds");
  return SDM_OK;
}

// Finishes a split-K contraction: ws holds raw sums indexed [batch entry][out row][N]; `rows` = out rows per entry.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const sdm_igemm_desc d, long long rows, int vec) {
  const long long all_rows = rows * d.batch;
  if (vec) {   // N % 4 == 0 and every row pointer 16-byte aligned: four columns per lane
    const int N4 = d.N >> 2;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < all_rows * N4; e += (long long)gridDim.x * 256) {
      const long long grow = sdm_idiv(e, N4);
      const int col = (int)(e - grow * N4) * 4;
      const long long b = sdm_idiv(grow, rows), orow = grow - b * rows;
      const long long b0 = sdm_idiv(b, d.bdiv), b1 = b - b0 * d.bdiv;
      const long long oo = b0 * d.o_bs0 + b1 * d.o_bs1 + orow * d.ldo + col;
      f32x4* w = (f32x4*)(d.ws + grow * d.N + col);
      f32x4 v = *w * d.alpha;
      *w = zero4;      // leave the scratch zero-filled for the next split-K launch
      if (d.bias) v += *(const f32x4*)(d.bias + col);
      if (d.out_preact) *(f32x4*)(d.out_preact + oo) = v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = sdm_act(v[i], d.act);
      if (d.res) v += *(const f32x4*)(d.res + b0 * d.r_bs0 + b1 * d.r_bs1 + orow * d.ldres + col);
      if (d.res2) v += *(const f32x4*)(d.res2 + orow * d.ldres2 + col);
      if (d.dact_z && (d.dact_ncols == 0 || col < d.dact_ncols)) {   // dact_ncols is a multiple of 4 on this path
        const f32x4 zv = *(const f32x4*)(d.dact_z + orow * d.ldz + col);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= sdm_dsilu(zv[i]);
      }
      *(f32x4*)(d.out + oo) = v;
    }
    return;
  }
  const long long total = all_rows * d.N;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long grow = sdm_idiv(e, d.N);
    const int col = (int)(e - grow * d.N);
    const long long b = sdm_idiv(grow, rows), orow = grow - b * rows;
    const long long b0 = sdm_idiv(b, d.bdiv), b1 = b - b0 * d.bdiv;
    const long long oo = b0 * d.o_bs0 + b1 * d.o_bs1 + orow * d.ldo + col;
    float v = d.ws[e] * d.alpha + (d.bias ? d.bias[col] : 0.f);
    d.ws[e] = 0.f;  // leave the scratch zero-filled for the next split-K launch
    if (d.out_preact) d.out_preact[oo] = v;
    v = sdm_act(v, d.act);
    if (d.res) v += d.res[b0 * d.r_bs0 + b1 * d.r_bs1 + orow * d.ldres + col];
    if (d.res2) v += d.res2[orow * d.ldres2 + col];
    if (d.dact_z && (d.dact_ncols == 0 || col < d.dact_ncols)) v *= sdm_dsilu(d.dact_z[orow * d.ldz + col]);
    d.out[oo] = v;
  }
}

template <int BM, int BN, int WM, int WN, bool BKM, bool VEC, bool AFF>
static int launch_igemm(const IgemmLaunch& L, int ntm, hipStream_t st) {
  constexpr size_t lds = (2 * BM * PITCH + 2 * (BKM ? BK * BN : BN * PITCH)) * sizeof(float) + BM * sizeof(long long);
  static bool attr_set = false;
  auto kern = igemm_kernel<BM, BN, WM, WN, BKM, VEC, AFF>;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid(ntm * L.ntn, L.d.nphase * L.splitk, L.d.batch);
  double taps = 0;
  for (int p = 0; p < L.d.nphase; ++p) taps += L.d.ntaps[p];
  {  // algorithmic bytes: every input row, weight and output element once (SURVEY 8d rule)
    const double images = (double)L.d.M / ((double)L.d.Ht * L.d.Wt);
    const double in_b = images * L.d.a.H * L.d.a.W * L.d.K, w_b = (double)L.d.K * L.d.N * taps, out_b = images * L.d.Ho * L.d.Wo * L.d.N;
    sdm_prof_shape(L.d.M, L.d.N, L.d.K, (int)taps, L.splitk, 4.0 * L.d.batch * (in_b + w_b + out_b));
  }
  sdm_prof_pre(st);
  SDM_LAUNCH(kern, grid, dim3(256), lds, st, L);
  sdm_prof_post(st, BM == 128 ? (BN == 128 ? 0 : 5) : (BM == 64 ? 1 : 2), 2.0 * L.d.M * L.d.N * L.d.K * taps * L.d.batch);
  SDM_CHECK_LAUNCH("igemm");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// Skinny-M weight streaming (M <= 64 tile-grid rows: the 4x4 / 8x8 layers at small batch and the N=1 sampling step).
// EXPERIMENT, off by default (see skinny_eligible): correct, but slower than the tiled kernels as measured.
// There the contraction is a pure weight stream (9-38 MB per launch against <= 256 KB of activations; AI 8-32 FLOP/B,
// SURVEY App. B), and the tiled kernels above are latency-bound on it: a workgroup keeps one or two 8 KB weight tiles in
// flight, 4 MB chip-wide, which at ~1.5 us loaded HBM latency cannot exceed ~2-3 TB/s (measured 1.5 TB/s = 19 % of HBM).
// Here NOTHING goes through LDS on the way in: a workgroup owns 32 output channels and one K-slice, its four waves split
// that slice, and every lane streams its own weight row (the MFMA B operand: lane j,h <- W[n0+j][k+4h..k+4h+3], one 16-byte
// load per 8 reduction steps) through a rotating queue of SK_D register slots -- 8 KB in flight per wave, 64 KB per CU.
// The activation fragments (MFMA A operand, rows = the <= 64 pixels) come from L2 the same way, with the tap gather, the
// concat split, the zero padding and a pending GroupNorm/AdaGN affine applied in registers.  The four partial 64x32
// tiles meet in LDS and leave as coalesced fp32 atomics into the zero-filled split-K scratch; splitk_epilogue_kernel
// finishes (bias, activation, residuals) exactly as for the tiled kernels.
// --------------------------------------------------------------------------------------------
#define SK_D 8   // weight fragments in flight per lane (x 16 B x 64 lanes = 8 KB per wave)

template <int TMB, bool BKM, bool AFF>
__global__ __launch_bounds__(256) void igemm_skinny_kernel(const IgemmLaunch L) {
  const sdm_igemm_desc& d = L.d;
  __shared__ long long s_aoff[SDM_MAX_TAPS * 64];     // source row of (tap, tile row) or -1 (padding / beyond M)
  __shared__ long long s_orow[64];
  __shared__ int s_img[64];
  __shared__ int s_tapw[SDM_MAX_TAPS];                // weight slice of a tap (read in the load path: LDS, not the
                                                      // kernel-argument segment, whose vector loads would drain the queue)
  __shared__ float s_red[4][TMB * 32][33];
  extern __shared__ __align__(16) float s_aff[];      // AFF: [2][images][K] scale | shift of the pending affine
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int tile_n = blockIdx.x / L.splitk, ks = blockIdx.x % L.splitk;
  const int phase = blockIdx.y;
  const int n0 = tile_n * 32;
  const int b0 = blockIdx.z / d.bdiv, b1 = blockIdx.z % d.bdiv;
  const float* a0 = d.a.p0 + b0 * d.a_bs0 + b1 * d.a_bs1;
  const float* a1 = d.a.p1;
  const float* bw = d.b + b0 * d.b_bs0 + b1 * d.b_bs1;
  const int HtWt = d.Ht * d.Wt;
  const int K = d.K, N = d.N, csplit = d.a.csplit;
  const int ntap = d.ntaps[phase];
  const int kch8 = K >> 3;                       // 8-deep reduction units per tap
  const int U = ntap * kch8;
  const int u0 = (int)((long long)U * ks / L.splitk), u1 = (int)((long long)U * (ks + 1) / L.splitk);
  const int ub = u0 + (int)((long long)(u1 - u0) * wave / 4), ue = u0 + (int)((long long)(u1 - u0) * (wave + 1) / 4);

  for (int e = tid; e < ntap * 64; e += 256) {
    const int tp = e >> 6, m = e & 63;
    long long off = -1;
    if (m < d.M) {
      const int n = m / HtWt, rem = m - n * HtWt;
      const int y = rem / d.Wt, x = rem - y * d.Wt;
      const int sy = y * d.sstride + d.tap_dy[phase][tp], sx = x * d.sstride + d.tap_dx[phase][tp];
      if ((unsigned)sy < (unsigned)d.a.H && (unsigned)sx < (unsigned)d.a.W) off = ((long long)n * d.a.H + sy) * d.a.W + sx;
    }
    s_aoff[e] = off;
  }
  if (tid < 64) {
    long long o = -1;
    int img = 0;
    if (tid < d.M) {
      const int n = tid / HtWt, rem = tid - n * HtWt;
      const int y = rem / d.Wt, x = rem - y * d.Wt;
      const int oy = y * d.ostride + d.ph_oy[phase], ox = x * d.ostride + d.ph_ox[phase];
      if (oy < d.Ho && ox < d.Wo) o = ((long long)n * d.Ho + oy) * d.Wo + ox;
      img = n;
    }
    s_orow[tid] = o;
    s_img[tid] = img;
  }
  if (tid < SDM_MAX_TAPS) s_tapw[tid] = d.tap_w[phase][tid];
  const int images = d.M / HtWt;
  if (AFF) {   // the tables are tiny (images x K): staged once, read at use time without touching the load queue
    for (int e = tid * 4; e < images * d.K; e += 1024) {
      const int n = e / d.K, c = e - n * d.K;
      *(f32x4*)&s_aff[e] = *(const f32x4*)(d.a.scale + (long long)n * d.a.aff_nstride + c);
      *(f32x4*)&s_aff[images * d.K + e] = *(const f32x4*)(d.a.shift + (long long)n * d.a.aff_nstride + c);
    }
  }
  __syncthreads();

  const bool nok = n0 + j < N;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // Branch-free load path (so that the compiler's counted s_waitcnt vmcnt(N) follows the static queue slots): the tap of
  // a unit, its weight slice and its source rows are re-derived per unit with arithmetic and two LDS reads.
  const int aimg0 = s_img[j], aimg1 = s_img[(TMB - 1) * 32 + j];
  struct Frag { f32x4 w; f32x4 a[TMB]; int c; int ok; };      // ok: bit i = row block i reads a real (non-padding) row
  auto load_unit = [&](Frag& f, int u) {      // issues the loads of reduction unit u (8 k-steps) for this lane
    const int t = u / kch8, q = u - t * kch8;
    const int c = q * 8 + 4 * h;
    const float* bt = bw + (long long)s_tapw[t] * d.b_tap_stride;
    if (!BKM) {
      f.w = *(const f32x4*)(bt + (long long)(nok ? n0 + j : 0) * d.ldb + c);
    } else {
      const float* p = bt + (long long)c * d.ldb + (nok ? n0 + j : 0);
      f.w.x = p[0];
      f.w.y = p[d.ldb];
      f.w.z = p[2 * d.ldb];
      f.w.w = p[3 * d.ldb];
    }
    f.ok = 0;
#pragma unroll
    for (int i = 0; i < TMB; ++i) {
      const long long ro = s_aoff[t * 64 + i * 32 + j];
      const long long r = ro < 0 ? 0 : ro;
      f.ok |= ro < 0 ? 0 : (1 << i);
      const float* p = c < csplit ? a0 + r * d.a.ld0 + c : a1 + r * d.a.ld1 + (c - csplit);
      f.a[i] = *(const f32x4*)p;
    }
    f.c = c;
  };
  f32x16 acc[TMB];
#pragma unroll
  for (int i = 0; i < TMB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  auto fma_unit = [&](const Frag& f) {
    const f32x4 wv = nok ? f.w : zero4;
#pragma unroll
    for (int i = 0; i < TMB; ++i) {
      f32x4 av = f.a[i];
      if (AFF) {
        const int img = i == 0 ? aimg0 : aimg1;
        av = av * *(const f32x4*)&s_aff[img * d.K + f.c] + *(const f32x4*)&s_aff[(images + img) * d.K + f.c];
      }
      av = (f.ok >> i) & 1 ? av : zero4;         // padding stays exactly zero AFTER the affine
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wv[s], acc[i], 0, 0, 0);
    }
  };

  // rotating queue: slot q holds unit (q + SK_D * round); a slot is refilled right after it has been consumed, so
  // SK_D - 1 younger fragments stay in flight behind every use.  Loads are unconditional (indices clamp to the wave's
  // last unit: a few redundant L2 hits at the tail); only the matrix instructions sit behind the uniform bound check.
  Frag fq[SK_D];
  const int cnt = ue - ub;
  if (cnt > 0) {
#pragma unroll
    for (int q = 0; q < SK_D; ++q) load_unit(fq[q], ub + (q < cnt ? q : cnt - 1));
    for (int base = 0; base < cnt; base += SK_D) {
#pragma unroll
      for (int q = 0; q < SK_D; ++q) {
        const int u = base + q;
        if (u < cnt) fma_unit(fq[q]);
        const int nu = u + SK_D;
        load_unit(fq[q], ub + (nu < cnt ? nu : cnt - 1));
      }
    }
  }

  // fold the four K-quarters and leave as coalesced atomics: lane = output channel, 2 rows per wave-instruction
#pragma unroll
  for (int i = 0; i < TMB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s_red[wave][i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h][j] = acc[i][r];
  __syncthreads();
  const int n = tid & 31;
  if (n0 + n < N) {
    for (int m = tid >> 5; m < TMB * 32; m += 8) {
      const long long orow = s_orow[m];
      if (orow < 0) continue;
      const float v = (s_red[0][m][n] + s_red[1][m][n]) + (s_red[2][m][n] + s_red[3][m][n]);
      atomicAdd(&d.ws[((long long)blockIdx.z * L.out_rows + orow) * N + n0 + n], v);
    }
  }
}

template <int TMB, bool BKM, bool AFF>
static int launch_skinny(const IgemmLaunch& L, hipStream_t st) {
  dim3 grid(L.ntn * L.splitk, L.d.nphase, L.d.batch);
  double taps = 0;
  for (int p = 0; p < L.d.nphase; ++p) taps += L.d.ntaps[p];
  {
    const double images = (double)L.d.M / ((double)L.d.Ht * L.d.Wt);
    const double in_b = images * L.d.a.H * L.d.a.W * L.d.K, w_b = (double)L.d.K * L.d.N * taps, out_b = images * L.d.Ho * L.d.Wo * L.d.N;
    sdm_prof_shape(L.d.M, L.d.N, L.d.K, (int)taps, L.splitk, 4.0 * L.d.batch * (in_b + w_b + out_b));
  }
  sdm_prof_pre(st);
  const size_t dyn = AFF ? (size_t)2 * (L.d.M / (L.d.Ht * L.d.Wt)) * L.d.K * sizeof(float) : 0;
  SDM_LAUNCH((igemm_skinny_kernel<TMB, BKM, AFF>), grid, dim3(256), dyn, st, L);
  sdm_prof_post(st, 6, 2.0 * L.d.M * L.d.N * L.d.K * taps * L.d.batch);
  SDM_CHECK_LAUNCH("igemm_skinny");
  return SDM_OK;
}

static int igemm_validate(const sdm_igemm_desc* d) {
  SDM_CHECK_ARG(d != nullptr, "igemm: null desc");
  SDM_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "igemm: bad M/N/K %d %d %d", d->M, d->N, d->K);
  SDM_CHECK_ARG(d->batch >= 1 && d->bdiv >= 1, "igemm: bad batch");
  SDM_CHECK_ARG(d->a.p0 && d->b && d->out, "igemm: null operand");
  SDM_CHECK_ARG(d->Ht > 0 && d->Wt > 0 && d->M % (d->Ht * d->Wt) == 0, "igemm: M %d not a multiple of Ht*Wt %d", d->M,
                d->Ht * d->Wt);
  SDM_CHECK_ARG(d->nphase >= 1 && d->nphase <= SDM_MAX_PHASE, "igemm: nphase");
  for (int p = 0; p < d->nphase; ++p) {
    SDM_CHECK_ARG(d->ntaps[p] >= 1 && d->ntaps[p] <= SDM_MAX_TAPS, "igemm: ntaps");
  }
  SDM_CHECK_ARG(d->a.p1 != nullptr || d->a.csplit >= d->K, "igemm: csplit %d < K %d without second source", d->a.csplit,
                d->K);
  SDM_CHECK_ARG(d->a.p1 == nullptr || d->a.csplit % 4 == 0, "igemm: concat split must be a multiple of 4");
  SDM_CHECK_ARG(d->a.scale == nullptr || (d->K % 4 == 0 && d->a.shift != nullptr && aligned16(d->a.scale) &&
                                          aligned16(d->a.shift) && d->a.aff_nstride % 4 == 0),
                "igemm: affine-on-load needs K %% 4 == 0 and aligned tables");
  SDM_CHECK_ARG(d->ostride >= 1 && d->sstride >= 1 && d->Ho > 0 && d->Wo > 0, "igemm: bad strides/output dims");
  SDM_CHECK_ARG(d->act >= 0 && d->act <= 2, "igemm: bad act");
  SDM_CHECK_ARG(!d->defer_epilogue || (d->batch == 1 && !d->res && !d->res2 && !d->dact_z),
                "igemm: a deferred epilogue carries only alpha / bias / activation (batch 1)");
  return SDM_OK;
}

// Skinny-M weight streaming applies to: one GEMM (no batch), M <= 64 tile-grid rows, 16-byte aligned operands, K a multiple
// of 8, at least 1 MB of weights to stream (below that the tiled kernels' latency is the same and they need no scratch).
#define SKINNY_AFF_LDS_MAX (48 * 1024)
static bool skinny_eligible(const sdm_igemm_desc* d) {
  // OPT-IN (SDM_IGEMM_SKINNY=1).  Measured on MI355X (profiles/r02_skinny_probe.txt): 35.0 us against the tiled kernel's
  // 20.6 us on M=64, N=K=1024, 9 taps (37.7 MB of weights) -- one 32-byte piece of 32 different weight rows per
  // wave-instruction reaches 1.1 TB/s where the LDS-DMA tiles (128 contiguous bytes of 8 rows per instruction) reach 1.8.
  static const char* on = getenv("SDM_IGEMM_SKINNY");
  if (!(on && on[0] == '1')) return false;
  if (d->batch != 1 || d->M > 64 || d->K % 8 != 0) return false;
  int taps = 0;
  for (int p = 0; p < d->nphase; ++p) taps += d->ntaps[p];
  if ((double)d->K * d->N * taps * 4.0 < 1048576.0) return false;
  const bool vecA = aligned16(d->a.p0) && d->a.ld0 % 4 == 0 && (d->a.p1 == nullptr || (aligned16(d->a.p1) && d->a.ld1 % 4 == 0));
  if (!vecA) return false;
  if (!d->b_kmajor && !(aligned16(d->b) && d->ldb % 4 == 0 && d->b_tap_stride % 4 == 0)) return false;
  if (d->a.scale != nullptr) {
    if (d->b_kmajor) return false;
    const long images = d->M / (d->Ht * d->Wt);
    if ((size_t)2 * images * d->K * sizeof(float) > SKINNY_AFF_LDS_MAX) return false;
  }
  return true;
}

static int skinny_splitk(const sdm_igemm_desc* d) {
  int maxtaps = 0;
  for (int p = 0; p < d->nphase; ++p) maxtaps = d->ntaps[p] > maxtaps ? d->ntaps[p] : maxtaps;
  const int U = maxtaps * (d->K / 8);                      // 8-deep reduction units per output tile
  const long wgs = (long)cdiv(d->N, 32) * d->nphase;
  int sk = (int)((512 + wgs - 1) / wgs);                  // ~2 workgroups per CU
  if (sk > U / 32) sk = U / 32;                            // >= 8 units per wave: the load queue must fill
  if (sk < 1) sk = 1;
  return sk;
}

struct TileCfg { int bm, bn; };
// index 4 is the weight-streaming kernel's marker (no tile of its own: entry unused); 5 = 64 x 128, the tile that spans all of N
// on the 128-channel layers (12 B/clk/CU of LDS-DMA feed at the matrix peak against the 64x64 tile's 16: DESIGN section 7 item 4)
static const TileCfg kCfgs[6] = {{128, 128}, {64, 64}, {32, 128}, {128, 64}, {64, 64}, {64, 128}};

// Tile / split-K choice by a small cost model (cycles per CU): a workgroup spends `mfma` matrix-pipe cycles plus
// `ovh` issue/barrier cycles per K-step; co-resident workgroups share the matrix pipe but hide each other's
// overhead.  Split-K adds an atomic pass over the output.
static void igemm_plan(const sdm_igemm_desc* d, int* cfg, int* splitk, size_t* ws_need) {
  if (skinny_eligible(d)) {     // cfg 4: weight streaming; always finishes through the zero-filled scratch
    *cfg = 4;
    *splitk = skinny_splitk(d);
    *ws_need = (size_t)(d->M / (d->Ht * d->Wt)) * d->Ho * d->Wo * d->batch * d->N * sizeof(float);
    return;
  }
  const long M = d->M, N = d->N;
  const long ph = (long)d->nphase * d->batch;
  int maxtaps = 0;
  for (int p = 0; p < d->nphase; ++p) maxtaps = d->ntaps[p] > maxtaps ? d->ntaps[p] : maxtaps;
  const int T = maxtaps * cdiv(d->K, BK);
  const long out_rows = (long)(d->M / (d->Ht * d->Wt)) * d->Ho * d->Wo;
  // Calibrated on MI355X (profiles/r01_igemm_tile_sweep.txt): a CU retires roughly one workgroup K-step per
  // (matrix cycles + staging/issue overhead) whatever the co-residency (co-resident workgroups overlap ~8 %),
  // so big tiles win whenever the grid still covers the 256 CUs evenly.
  static const double kMfma[6] = {4096, 1024, 1024, 2048, 0, 2048};
  static const double kOvh[6] = {3000, 900, 900, 1850, 0, 1850};   // 128-row tiles charged +600: they co-run worse with the
                                                                    // weight-gradient stream (A/B on the training step)
  static const int kOcc[6] = {2, 4, 3, 2, 0, 2};
  static const char* t5_s = getenv("SDM_IGEMM_TILE5");     // "1": the 64x128 tile takes part in the choice
  static const char* pen_s = getenv("SDM_IGEMM_BIGTILE_PENALTY");   // tuning aid: extra cycles charged to 128-row tiles
  const double pen = pen_s ? atof(pen_s) : 0.0;
  static const char* sc_s = getenv("SDM_IGEMM_SPLIT_COST");    // cycles charged to a split (finishing launch + atomic drain)
  const double split_cost = sc_s ? atof(sc_s) : 6000.0;
  double best = 1e300;
  int bc = 1, bs = 1;
  for (int c = 0; c < 6; ++c) {
    if (c == 4 || (c == 5 && !(t5_s && t5_s[0] == '1'))) continue;
    if (c == 5 && (M < 64 || N <= 64)) continue;
    if (c == 2 && M > 32) continue;
    if (c != 2 && M <= 32) continue;
    if ((c == 0 || c == 3) && M < 96) continue;
    if (c == 0 && N <= 64) continue;
    const long tiles = (long)cdiv(M, kCfgs[c].bm) * cdiv(N, kCfgs[c].bn) * ph;
    for (int sk = 1; sk <= 64; sk *= 2) {
      if (sk > 1 && sk > T / 8) break;
      const long nb = tiles * sk;
      const long per_cu = (nb + 255) / 256;
      const double overlap = (per_cu >= 2 && kOcc[c] >= 2) ? 0.92 : 1.0;
      double t = (double)per_cu * ((double)T / sk) * (kMfma[c] + kOvh[c] + ((c == 0 || c == 3) ? pen : 0.0)) * overlap + 3000.0;
      if (sk > 1) t += split_cost + (double)out_rows * d->batch * N * 4.0 * sk / 256.0 / 4.0;   // atomics + finishing launch
      if (t < best) { best = t; bc = c; bs = sk; }
    }
  }
  // tuning override (benchmarks only): SDM_IGEMM_FORCE="cfg,splitk"
  static const char* force = getenv("SDM_IGEMM_FORCE");
  if (force) {
    int fc = -1, fs = 1;
    if (sscanf(force, "%d,%d", &fc, &fs) >= 1 && fc >= 0 && fc < 6 && fc != 4) {
      bc = fc;
      bs = (fs >= 1 && fs <= T) ? fs : 1;
    }
  }
  *cfg = bc;
  *splitk = bs;
  *ws_need = bs > 1 ? (size_t)out_rows * d->batch * N * sizeof(float) : 0;
}

// (Split-K finished INSIDE the contraction kernel by each tile's last-arriving workgroup was built twice -- round 2 with returning
// exchanges, round 4 with agent-scope loads and arrival counters behind the sums (commit 3bcda65) -- and measured slower both
// times: round 4, 64x64 training step 7.35 -> 9.01 ms, N=1 sampling step 1.10 -> 1.51 ms although 47 / 20 launches per step were
// gone (profiles/r04_splitk_inkernel.txt).  Every slice must release its atomics at agent scope before it arrives, which on
// eight XCDs with one L2 each is an L2 write-back + invalidate per workgroup; the kernel boundary gives the separate finishing
// launch that ordering for free.  Even switched off, the extra code cost the 64x64 class 3 % (registers): removed.)
extern "C" size_t sdm_igemm_workspace_bytes(const sdm_igemm_desc* d) {
  if (igemm_validate(d) != SDM_OK) return 0;
  int c, sk;
  size_t need;
  igemm_plan(d, &c, &sk, &need);
  return need;
}

template <int BM, int BN, int WM, int WN>
static int dispatch_igemm(const IgemmLaunch& L, int ntm, bool kmajor, bool vec, bool aff, hipStream_t st) {
  static const char* noglds = getenv("SDM_IGEMM_GLDS");
  if (vec && !aff && !(noglds && noglds[0] == '0')) {
    constexpr int NBUF = (BM * BN >= 128 * 128) ? 2 : 3;
    // Small tiles, at most two workgroups per CU in the grid: 64-deep K-steps with a 2-slot ring (one barrier per 32
    // MFMAs instead of 16; the same 64 KB of LDS would not allow a third workgroup anyway).  Denser grids keep the
    // 32-deep / 3-slot form, whose 48 KB lets three workgroups share a CU.  SDM_IGEMM_BK64=0 turns it off.
    static const char* bk64 = getenv("SDM_IGEMM_BK64");
    const long nblocks = (long)ntm * L.ntn * L.d.nphase * L.splitk * L.d.batch;
    if (BM * BN <= 64 * 128 && L.d.K % 64 == 0 && nblocks <= 512 && !(bk64 && bk64[0] == '0')) {
      IgemmLaunch L2 = L;
      L2.kchunks = cdiv(L.d.K, 64);
      // At most one workgroup per CU in the grid (the N=1 sampling step, the 4x4 / 8x8 layers): the whole LDS is this
      // workgroup's, so the ring goes 3-4 deep (up to 128 KB: two or three 64-deep tiles in flight).  Such launches are pure latency
      // chains -- a K-step computes for ~0.45 us against ~1 us of weight-fetch latency -- and the deeper ring hides it.
      static const char* deep = getenv("SDM_IGEMM_DEEP");
      constexpr int DEEP = (BM + BN <= 128) ? 4 : 3;      // 64x64: 4 x 32 KB; 32x128: 3 x 40 KB (the CU has 160 KB)
      if (nblocks <= 256 && !(deep && deep[0] == '0'))
        return kmajor ? launch_igemm_glds<BM, BN, WM, WN, true, DEEP, 64>(L2, ntm, st)
                      : launch_igemm_glds<BM, BN, WM, WN, false, DEEP, 64>(L2, ntm, st);
      return kmajor ? launch_igemm_glds<BM, BN, WM, WN, true, 2, 64>(L2, ntm, st)
                    : launch_igemm_glds<BM, BN, WM, WN, false, 2, 64>(L2, ntm, st);
    }
    return kmajor ? launch_igemm_glds<BM, BN, WM, WN, true, NBUF, 32>(L, ntm, st)
                  : launch_igemm_glds<BM, BN, WM, WN, false, NBUF, 32>(L, ntm, st);
  }
  // the affine-on-load variant exists only for K-contiguous weights (forward convs); other combinations never carry it
  if (aff) {
    if (kmajor) return sdm_set_error(SDM_ERR_UNSUPPORTED, "igemm: affine-on-load with K-major weights");
    return vec ? launch_igemm<BM, BN, WM, WN, false, true, true>(L, ntm, st)
               : launch_igemm<BM, BN, WM, WN, false, false, true>(L, ntm, st);
  }
  if (kmajor)
    return vec ? launch_igemm<BM, BN, WM, WN, true, true, false>(L, ntm, st)
               : launch_igemm<BM, BN, WM, WN, true, false, false>(L, ntm, st);
  return vec ? launch_igemm<BM, BN, WM, WN, false, true, false>(L, ntm, st)
             : launch_igemm<BM, BN, WM, WN, false, false, false>(L, ntm, st);
}

extern "C" int sdm_igemm(const sdm_igemm_desc* d, sdm_stream_t stream) {
  int rc = igemm_validate(d);
  if (rc != SDM_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  IgemmLaunch L;
  L.d = *d;
  int cfg;
  size_t need;
  igemm_plan(d, &cfg, &L.splitk, &need);
  if (cfg == 4 && (d->ws == nullptr || d->ws_bytes < need)) {     // no scratch for the streaming kernel: tiled path
    static const TileCfg fallback = {64, 64};
    (void)fallback;
    cfg = d->M <= 32 ? 2 : 1;
    L.splitk = 1;
    need = 0;
  }
  if (L.splitk > 1 && (d->ws == nullptr || d->ws_bytes < need)) L.splitk = 1;  // no scratch: run unsplit
  L.kchunks = cdiv(d->K, BK);
  const bool vecA = aligned16(d->a.p0) && d->a.ld0 % 4 == 0 && d->a_bs0 % 4 == 0 && d->a_bs1 % 4 == 0 &&
                    (d->a.p1 == nullptr || (aligned16(d->a.p1) && d->a.ld1 % 4 == 0));
  const bool vecB = aligned16(d->b) && d->ldb % 4 == 0 && d->b_tap_stride % 4 == 0 && d->b_bs0 % 4 == 0 && d->b_bs1 % 4 == 0;
  const bool vec = vecA && vecB && d->K % 4 == 0 && (!d->b_kmajor || d->N % 4 == 0);
  const long out_rows = (long)(d->M / (d->Ht * d->Wt)) * d->Ho * d->Wo;
  L.out_rows = out_rows;
  igemm_geometry(L);
  const int tcfg = cfg == 4 ? 1 : cfg;              // (the streaming kernel has no tile table entry)
  const int ntm = cdiv(d->M, kCfgs[tcfg].bm);
  L.ntn = cdiv(d->N, kCfgs[tcfg].bn);
  if (cfg != 4) {
    // per-XCD L2 fill of the two tile orders: an XCD's run of c tiles spans (mt x nt) tiles of A rows / B columns
    static const char* xo = getenv("SDM_IGEMM_XCD_ORDER");   // tuning override: "n" / "m"
    const long c = cdiv((long)ntm * L.ntn, 8);
    int taps_all = 0;                                        // phases sit in gridDim.y: same XCD as their tile
    for (int ph = 0; ph < d->nphase; ++ph) taps_all += d->ntaps[ph];
    const double a_tile = (double)kCfgs[tcfg].bm * d->K, b_tile = (double)kCfgs[tcfg].bn * d->K * taps_all;
    const double nfast = (double)cdiv(c, L.ntn) * a_tile + (double)(c < L.ntn ? c : L.ntn) * b_tile;
    const double mfast = (double)(c < ntm ? c : ntm) * a_tile + (double)cdiv(c, ntm) * b_tile;
    L.mfast = mfast < nfast ? 1 : 0;
    if (xo && xo[0] == 'n') L.mfast = 0;
    if (xo && xo[0] == 'm') L.mfast = 1;
    if (xo && xo[0] == '0') L.mfast = 2;                     // round-robin order as dispatched (for A/B measurements)
  }
  const bool km = d->b_kmajor != 0;
  const bool aff = d->a.scale != nullptr;
  bool finish = L.splitk > 1;
  if (cfg == 4) {
    L.ntn = cdiv(d->N, 32);
    finish = true;
    if (d->M <= 32) rc = km ? launch_skinny<1, true, false>(L, st) : (aff ? launch_skinny<1, false, true>(L, st) : launch_skinny<1, false, false>(L, st));
    else rc = km ? launch_skinny<2, true, false>(L, st) : (aff ? launch_skinny<2, false, true>(L, st) : launch_skinny<2, false, false>(L, st));
  } else
  if (cfg == 0) rc = dispatch_igemm<128, 128, 2, 2>(L, ntm, km, vec, aff, st);
  else if (cfg == 1) rc = dispatch_igemm<64, 64, 2, 2>(L, ntm, km, vec, aff, st);
  else if (cfg == 2) rc = dispatch_igemm<32, 128, 1, 4>(L, ntm, km, vec, aff, st);
  else if (cfg == 5) rc = dispatch_igemm<64, 128, 2, 2>(L, ntm, km, vec, aff, st);
  else rc = dispatch_igemm<128, 64, 2, 2>(L, ntm, km, vec, aff, st);
  if (rc != SDM_OK) return rc;
  if (finish && d->defer_epilogue) return SDM_DEFERRED;     // raw sums stay in ws: the caller's next kernel finishes them
  if (finish) {
    const long long total = (long long)out_rows * d->batch * d->N;
    int grid = (int)((total + 1023) / 1024);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    const int evec = d->N % 4 == 0 && aligned16(d->ws) && aligned16(d->out) && d->ldo % 4 == 0 && d->o_bs0 % 4 == 0 &&
                     d->o_bs1 % 4 == 0 && d->r_bs0 % 4 == 0 && d->r_bs1 % 4 == 0 &&
                     (d->bias == nullptr || aligned16(d->bias)) && (d->out_preact == nullptr || aligned16(d->out_preact)) &&
                     (d->res == nullptr || (aligned16(d->res) && d->ldres % 4 == 0)) &&
                     (d->res2 == nullptr || (aligned16(d->res2) && d->ldres2 % 4 == 0)) &&
                     (d->dact_z == nullptr || (aligned16(d->dact_z) && d->ldz % 4 == 0 && d->dact_ncols % 4 == 0));
    if (evec) {
      grid = (int)((total / 4 + 255) / 256);
      if (grid > 4096) grid = 4096;
      if (grid < 1) grid = 1;
    }
    SDM_LAUNCH(splitk_epilogue_kernel, dim3(grid), dim3(256), 0, st, *d, (long long)out_rows, evec);
    SDM_CHECK_LAUNCH("splitk_epilogue");
  }
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// weight gradient: dw[tap][i][j] (+)= sum_p G[gpix(p,tap)][i] * X[xpix(p,tap)][j]
// Both operands are pixel rows with the reduction index on the row: tiles are stored [k][channel]
// and fragments read down the columns with ds_read_b32.
// --------------------------------------------------------------------------------------------
struct WgradLaunch {
  sdm_wgrad_desc d;
  int splitk, ntn, atomic_out;
};

template <int BT, int WM, int WN, bool VEC, bool AFF>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradLaunch L) {
  const sdm_wgrad_desc& d = L.d;
  constexpr int TM = BT / (WM * 32), TN = BT / (WN * 32);
  constexpr int IT = BT / 32;
  constexpr int TILE = BK * BT;
  constexpr int CPR = BT / 4, RPP = 256 / CPR;
  extern __shared__ __align__(16) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * TILE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, h = lane >> 5, l31 = lane & 31;
  // gridDim.x = pixel slices x tiles, slice-major behind the XCD remap: one XCD's L2 sees one contiguous range of
  // pixels of both operands (for all tiles and, via gridDim.y, all taps) instead of every XCD streaming all of them.
  // (Round 4 tried the taps of a slice as neighbours of ONE XCD run, as the bf16 kernels now have them: at the fp32 training
  // shapes, whose operands sit in L2 anyway, the 64x64 class got 2.7 % slower and the 128x128 class 1 % faster: not adopted.)
  const int ntiles_ = gridDim.x / L.splitk;
  const int wg_ = xcd_contiguous(blockIdx.x, gridDim.x);
  const int ks = wg_ / ntiles_, tile_ = wg_ - ks * ntiles_;
  const int tile_n = tile_ % L.ntn, tile_m = tile_ / L.ntn;
  const int tp = blockIdx.y;
  const int m0 = tile_m * BT, n0 = tile_n * BT;
  const int b0 = blockIdx.z / d.bdiv, b1 = blockIdx.z % d.bdiv;
  const float* g0 = d.g.p0 + b0 * d.g_bs0 + b1 * d.g_bs1;
  const float* x0 = d.x.p0 + b0 * d.x_bs0 + b1 * d.x_bs1;
  const float* x1 = d.x.p1;
  const int HW = d.Hk * d.Wk;
  const long long P = (long long)d.images * HW;
  const int T = (int)((P + BK - 1) / BK);
  const int it0 = (int)((long long)T * ks / L.splitk), it1 = (int)((long long)T * (ks + 1) / L.splitk);
  const int gdy = d.g_dy[tp], gdx = d.g_dx[tp], xdy = d.x_dy[tp], xdx = d.x_dx[tp];
  const int cm = m0 + 4 * (tid % CPR), cn = n0 + 4 * (tid % CPR);
  const bool gch = cm < d.M, xch = cn < d.N;
  const bool xsrc0 = cn < d.x.csplit;

  // pixel coordinates of this thread's IT rows, advanced by BK pixels per K-step without divisions
  int pn[IT], py[IT], px[IT];
  const int adv_x = BK % d.Wk, q32 = BK / d.Wk;
  const int adv_y = q32 % d.Hk, adv_n = q32 / d.Hk;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const long long p = (long long)it0 * BK + tid / CPR + RPP * i;
    const int n = (int)(p / HW), rem = (int)(p - (long long)n * HW);
    pn[i] = n;
    py[i] = rem / d.Wk;
    px[i] = rem - py[i] * d.Wk;
  }

  f32x4 ra[IT], rb[IT];
  f32x4 rsc[AFF ? IT : 1], rsh[AFF ? IT : 1];
  unsigned gmask = 0, xmask = 0;
  const bool do_bias = d.dbias != nullptr && tile_n == 0 && tp == 0;   // block-uniform
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto load = [&]() {
    gmask = 0;
    xmask = 0;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int n = pn[i], y = py[i], x = px[i];
      const bool in = n < d.images;
      const int gy = y * d.g_stride + gdy, gx = x * d.g_stride + gdx;
      const int xy = y * d.x_stride + xdy, xx = x * d.x_stride + xdx;
      const bool gu = in && gch && (unsigned)gy < (unsigned)d.g.H && (unsigned)gx < (unsigned)d.g.W;
      const bool xu = in && xch && (unsigned)xy < (unsigned)d.x.H && (unsigned)xx < (unsigned)d.x.W;
      gmask |= gu ? (1u << i) : 0u;
      xmask |= xu ? (1u << i) : 0u;
      const long long xrow = ((long long)n * d.x.H + xy) * d.x.W + xx;
      const float* gq = g0 + (((long long)n * d.g.H + gy) * d.g.W + gx) * d.g.ld0 + cm;
      const float* xq = xsrc0 ? x0 + xrow * d.x.ld0 + cn : x1 + xrow * d.x.ld1 + (cn - d.x.csplit);
      if (VEC) {
        gq = gu ? gq : d.g.p0;
        xq = xu ? xq : d.x.p0;
        ra[i] = *(const f32x4*)gq;
        rb[i] = *(const f32x4*)xq;
      } else {
        f32x4 va = zero4, vb = zero4;
        if (gu) {
          va.x = gq[0];
          if (cm + 1 < d.M) va.y = gq[1];
          if (cm + 2 < d.M) va.z = gq[2];
          if (cm + 3 < d.M) va.w = gq[3];
        }
        if (xu) {
          vb.x = xq[0];
          if (cn + 1 < d.N) vb.y = xq[1];
          if (cn + 2 < d.N) vb.z = xq[2];
          if (cn + 3 < d.N) vb.w = xq[3];
        }
        ra[i] = va;
        rb[i] = vb;
      }
      if (AFF) {
        const long long so = xu ? (long long)n * d.x.aff_nstride + cn : 0;
        rsc[i] = *(const f32x4*)(d.x.scale + so);
        rsh[i] = *(const f32x4*)(d.x.shift + so);
      }
      // advance to the next K-step
      int nx = x + adv_x, ny = y + adv_y, nn = n + adv_n;
      if (nx >= d.Wk) { nx -= d.Wk; ny += 1; }
      if (ny >= d.Hk) { ny -= d.Hk; nn += 1; }
      px[i] = nx; py[i] = ny; pn[i] = nn;
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int o = buf * TILE + (tid / CPR + RPP * i) * BT + 4 * (tid % CPR);
      f32x4 vb = rb[i];
      if (AFF) vb = vb * rsc[i] + rsh[i];
      const f32x4 va = (gmask >> i) & 1u ? ra[i] : zero4;
      if (do_bias) bsum += va;
      *(f32x4*)&As[o] = va;
      *(f32x4*)&Bs[o] = (xmask >> i) & 1u ? vb : zero4;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (it0 < it1) load();
  for (int it = it0; it < it1; ++it) {
    const int buf = (it - it0) & 1;
    store(buf);
    __syncthreads();
    if (it + 1 < it1) load();
    const float* Ab = As + buf * TILE + (h * 4) * BT + wm * TM * 32 + l31;
    const float* Bb = Bs + buf * TILE + (h * 4) * BT + wn * TN * 32 + l31;
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      float af[TM][4], bf[TN][4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i][s] = Ab[(kg * 8 + s) * BT + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j][s] = Bb[(kg * 8 + s) * BT + j * 32];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
  }

  if (do_bias) {  // column sums of the gradient tile rows this block walked: fold the RPP row phases through LDS
    __syncthreads();
    *(f32x4*)&As[(tid / CPR) * BT + 4 * (tid % CPR)] = bsum;
    __syncthreads();
    if (tid < BT && m0 + tid < d.M) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < RPP; ++r) a += As[r * BT + tid];
      atomicAdd(&d.dbias[m0 + tid], a);
    }
  }

  float* dw = d.dw + b0 * d.w_bs0 + b1 * d.w_bs1 + (long long)d.tap_w[tp] * d.w_tap_stride;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + l31;
      if (col >= d.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= d.M) continue;
        float* q = dw + (long long)row * d.ldw + col;
        const float v = acc[i][j][r] * d.alpha;
        if (L.atomic_out) atomicAdd(q, v);
        else *q = v;
      }
    }
}

// LDS-DMA variant of the weight-gradient kernel (aligned operands, no pending affine on x): both pixel-row tiles
// go global -> LDS with global_load_lds_dwordx4 into a ring of NBUF slots (lane-linear [k][channel] images, read
// down the columns, so no swizzle is needed); out-of-image taps read the zero word.
template <int BT, int WM, int WN, int NBUF>
__global__ __launch_bounds__(256) void wgrad_glds_kernel(const WgradLaunch L) {
  const sdm_wgrad_desc& d = L.d;
  constexpr int TM = BT / (WM * 32), TN = BT / (WN * 32);
  constexpr int IT = BT / 32;
  constexpr int TILE = BK * BT;
  constexpr int CPR = BT / 4, RPP = 256 / CPR;
  constexpr int G = 2 * IT;
  constexpr int PD = NBUF - 1;
  extern __shared__ __align__(16) float smem[];
  float* As = smem;
  float* Bs = smem + NBUF * TILE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, h = lane >> 5, l31 = lane & 31;
  // gridDim.x = pixel slices x tiles, slice-major behind the XCD remap: one XCD's L2 sees one contiguous range of
  // pixels of both operands (for all tiles and, via gridDim.y, all taps) instead of every XCD streaming all of them.
  // (Round 4 tried the taps of a slice as neighbours of ONE XCD run, as the bf16 kernels now have them: at the fp32 training
  // shapes, whose operands sit in L2 anyway, the 64x64 class got 2.7 % slower and the 128x128 class 1 % faster: not adopted.)
  const int ntiles_ = gridDim.x / L.splitk;
  const int wg_ = xcd_contiguous(blockIdx.x, gridDim.x);
  const int ks = wg_ / ntiles_, tile_ = wg_ - ks * ntiles_;
  const int tile_n = tile_ % L.ntn, tile_m = tile_ / L.ntn;
  const int tp = blockIdx.y;
  const int m0 = tile_m * BT, n0 = tile_n * BT;
  const int b0 = blockIdx.z / d.bdiv, b1 = blockIdx.z % d.bdiv;
  const float* g0 = d.g.p0 + b0 * d.g_bs0 + b1 * d.g_bs1;
  const float* x0 = d.x.p0 + b0 * d.x_bs0 + b1 * d.x_bs1;
  const float* x1 = d.x.p1;
  const float* zero = g_zero_chunk;
  const int HW = d.Hk * d.Wk;
  const long long P = (long long)d.images * HW;
  const int T = (int)((P + BK - 1) / BK);
  const int it0 = (int)((long long)T * ks / L.splitk), it1 = (int)((long long)T * (ks + 1) / L.splitk);
  const int gdy = d.g_dy[tp], gdx = d.g_dx[tp], xdy = d.x_dy[tp], xdx = d.x_dx[tp];
  const int cm = m0 + 4 * (tid % CPR), cn = n0 + 4 * (tid % CPR);
  const bool gch = cm < d.M, xch = cn < d.N;
  const bool xsrc0 = cn < d.x.csplit;
  const bool do_bias = d.dbias != nullptr && tile_n == 0 && tp == 0;   // block-uniform
  float bacc = 0.f;

  int pn[IT], py[IT], px[IT];
  const int adv_x = BK % d.Wk, q32 = BK / d.Wk;
  const int adv_y = q32 % d.Hk, adv_n = q32 / d.Hk;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const long long p = (long long)it0 * BK + tid / CPR + RPP * i;
    const int n = (int)(p / HW), rem = (int)(p - (long long)n * HW);
    pn[i] = n;
    py[i] = rem / d.Wk;
    px[i] = rem - py[i] * d.Wk;
  }

  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  int issued = it0;
  auto issue_next = [&]() {
    if (issued < it1) {
      const int buf = (issued - it0) % NBUF;
      float* Ad = As + buf * TILE + __builtin_amdgcn_readfirstlane(wave) * 256;
      float* Bd = Bs + buf * TILE + __builtin_amdgcn_readfirstlane(wave) * 256;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int n = pn[i], y = py[i], x = px[i];
        const bool in = n < d.images;
        const int gy = y * d.g_stride + gdy, gx = x * d.g_stride + gdx;
        const int xy = y * d.x_stride + xdy, xx = x * d.x_stride + xdx;
        const bool gu = in && gch && (unsigned)gy < (unsigned)d.g.H && (unsigned)gx < (unsigned)d.g.W;
        const bool xu = in && xch && (unsigned)xy < (unsigned)d.x.H && (unsigned)xx < (unsigned)d.x.W;
        const long long xrow = ((long long)n * d.x.H + xy) * d.x.W + xx;
        const float* gq = g0 + (((long long)n * d.g.H + gy) * d.g.W + gx) * d.g.ld0 + cm;
        const float* xq = xsrc0 ? x0 + xrow * d.x.ld0 + cn : x1 + xrow * d.x.ld1 + (cn - d.x.csplit);
        gq = gu ? gq : zero;
        xq = xu ? xq : zero;
        __builtin_amdgcn_global_load_lds((glb_ptr)gq, (lds_ptr)(Ad + i * RPP * BT), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr)xq, (lds_ptr)(Bd + i * RPP * BT), 16, 0, 0);
        int nx = x + adv_x, ny = y + adv_y, nn = n + adv_n;
        if (nx >= d.Wk) { nx -= d.Wk; ny += 1; }
        if (ny >= d.Hk) { ny -= d.Hk; nn += 1; }
        px[i] = nx; py[i] = ny; pn[i] = nn;
      }
    }
    ++issued;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int p = 0; p < PD; ++p) issue_next();
  for (int it = it0; it < it1; ++it) {
    const int buf = (it - it0) % NBUF;
    const int younger = (it1 - 1 - it) < (PD - 1) ? (it1 - 1 - it) : (PD - 1);
    if (younger >= 2) wait_vmcnt<2 * G>();
    else if (younger == 1) wait_vmcnt<G>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    issue_next();
    const float* Ab = As + buf * TILE + (h * 4) * BT + wm * TM * 32 + l31;
    const float* Bb = Bs + buf * TILE + (h * 4) * BT + wn * TN * 32 + l31;
    if (do_bias && tid < BT) {   // column sums of the gradient tile (bias gradient)
      const float* col = As + buf * TILE + tid;
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < BK; ++k) a += col[k * BT];
      bacc += a;
    }
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      float af[TM][4], bf[TN][4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i][s] = Ab[(kg * 8 + s) * BT + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j][s] = Bb[(kg * 8 + s) * BT + j * 32];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
  }
  if (do_bias && tid < BT && m0 + tid < d.M) atomicAdd(&d.dbias[m0 + tid], bacc);

  float* dw = d.dw + b0 * d.w_bs0 + b1 * d.w_bs1 + (long long)d.tap_w[tp] * d.w_tap_stride;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + l31;
      if (col >= d.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= d.M) continue;
        float* q = dw + (long long)row * d.ldw + col;
        const float v = acc[i][j][r] * d.alpha;
        if (L.atomic_out) atomicAdd(q, v);
        else *q = v;
      }
    }
}

template <int BT, int WM, int WN, int NBUF>
static int launch_wgrad_glds(const WgradLaunch& L, int ntm, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * NBUF * BK * BT * sizeof(float);
  dim3 grid(ntm * L.ntn * L.splitk, L.d.ntaps, L.d.batch);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)wgrad_glds_kernel<BT, WM, WN, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    attr_set = true;
  }
  {
    const double px = (double)L.d.images;
    const double g_b = px * L.d.g.H * L.d.g.W * L.d.M, x_b = px * L.d.x.H * L.d.x.W * L.d.N, w_b = (double)L.d.M * L.d.N * L.d.ntaps;
    sdm_prof_shape(L.d.M, L.d.N, (int)((long)L.d.images * L.d.Hk * L.d.Wk), L.d.ntaps, L.splitk, 4.0 * L.d.batch * (g_b + x_b + w_b));
  }
  sdm_prof_pre(st);
  SDM_LAUNCH((wgrad_glds_kernel<BT, WM, WN, NBUF>), grid, dim3(256), lds, st, L);
  sdm_prof_post(st, BT == 128 ? 3 : 4,
                2.0 * L.d.M * L.d.N * (double)L.d.images * L.d.Hk * L.d.Wk * L.d.ntaps * L.d.batch);
  SDM_CHECK_LAUNCH("wgrad_glds");
  return SDM_OK;
}

__global__ void zero_rows_kernel(float* p, long long rows, int cols, long long ld) {
  const long long total = rows * cols;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = e / cols;
    p[r * ld + (e - r * cols)] = 0.f;
  }
}

template <int BT, int WM, int WN, bool VEC, bool AFF>
static int launch_wgrad(const WgradLaunch& L, int ntm, hipStream_t st) {
  constexpr size_t lds = 4 * BK * BT * sizeof(float);
  dim3 grid(ntm * L.ntn * L.splitk, L.d.ntaps, L.d.batch);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<BT, WM, WN, VEC, AFF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  {
    const double px = (double)L.d.images;
    const double g_b = px * L.d.g.H * L.d.g.W * L.d.M, x_b = px * L.d.x.H * L.d.x.W * L.d.N, w_b = (double)L.d.M * L.d.N * L.d.ntaps;
    sdm_prof_shape(L.d.M, L.d.N, (int)((long)L.d.images * L.d.Hk * L.d.Wk), L.d.ntaps, L.splitk, 4.0 * L.d.batch * (g_b + x_b + w_b));
  }
  sdm_prof_pre(st);
  SDM_LAUNCH((wgrad_kernel<BT, WM, WN, VEC, AFF>), grid, dim3(256), lds, st, L);
  sdm_prof_post(st, BT == 128 ? 3 : 4,
                2.0 * L.d.M * L.d.N * (double)L.d.images * L.d.Hk * L.d.Wk * L.d.ntaps * L.d.batch);
  SDM_CHECK_LAUNCH("wgrad");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// Weight gradient of the two EDGE convolutions (in_layer.0: Cin <= 4 -> C, out_layers.1: C -> Cout <= 4; U_Net.py:55-60,121-128;
// autograd of their nn.Conv2d weights, train_diffusion.py:358).  One operand has <= 4 channels: on the 64x64 matrix-core tile the
// contraction is 95 % padding and split 40 ways over the pixels (53 + 56 us of the 7.3 ms headline step at 2 TFLOP/s).  Here:
//   dw[tap][..] += alpha * sum_q WIDE[q][c] * THIN[q + s * d_tap][t]          (s = +1: the activations are thin; -1: the gradient)
// a workgroup takes R image rows of the wide operand (lane = wide channel, the spare lanes of a row walk further pixels), keeps
// the thin operand's R + 2 rows in LDS as one 16-byte quad per pixel (rows are stored padded to 4 channels), holds the 9 x 3
// partial sums of its channel in registers, and ends in one fp32 atomic per (tap, channel pair) and workgroup.
// --------------------------------------------------------------------------------------------
struct ThinWgradLaunch {
  const float* wide;
  const float* thin;
  long long ld_wide;
  float* dw;
  long long w_tap_stride, stride_c, stride_t;
  float* dbias_wide;   // += column sums of the wide operand (the gradient rows of the stem conv), or NULL
  float* dbias_thin;   // += column sums of the thin operand (the gradient rows of the head conv), or NULL
  int Cw, TH, images, H, W, R, sgn, lw;     // lw: log2(W) or -1
  float alpha;
  int tapw[9];                              // weight slice of tap t = (dy + 1) * 3 + (dx + 1)
};

__global__ __launch_bounds__(256) void wgrad_thin_kernel(const ThinWgradLaunch L) {
  extern __shared__ __align__(16) float thin_smem[];
  f32x4* patch = (f32x4*)thin_smem;                        // [(R + 2) rows][W + 2] quads, zero outside the image
  __shared__ float red[256][29];
  const int tid = threadIdx.x, W = L.W, H = L.H, R = L.R, PW = W + 2;
  const long long r0 = (long long)blockIdx.x * R;          // first image row (n * H + y) of this workgroup; R divides H
  const int y0 = (int)(r0 % H);
  for (int i = tid; i < (R + 2) * PW; i += 256) {
    const int pr = i / PW, px = i - pr * PW;
    const int y = y0 - 1 + pr, x = px - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) v = *(const f32x4*)(L.thin + ((r0 - 1 + pr) * W + x) * 4);
    patch[i] = v;
  }
  __syncthreads();
  int cw = 1;
  while (cw < L.Cw) cw <<= 1;                              // lanes per pixel (Cw rounded up to a power of two <= 256)
  const int c = tid % cw, pl = tid / cw, NPL = 256 / cw;
  float acc[9][3];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t][0] = acc[t][1] = acc[t][2] = 0.f;
  float bsum = 0.f;
  const int npix = R * W, sgn = L.sgn;
  if (c < L.Cw) {
    const float* wp = L.wide + r0 * W * L.ld_wide + c;
    for (int i0 = pl; i0 < npix; i0 += 8 * NPL) {          // eight pixels per trip: their loads overlap
      float wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * NPL;
        wv[j] = i < npix ? wp[(long long)i * L.ld_wide] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * NPL;                        // (a pixel beyond the range carries wv = 0 and reads inside the patch)
        const int ii = i < npix ? i : 0;
        const int yy = L.lw >= 0 ? ii >> L.lw : ii / W, x = ii - yy * W;
        const f32x4* pc = patch + (yy + 1) * PW + x + 1;
        bsum += wv[j];
#pragma unroll
        for (int t = 0; t < 9; ++t) {                      // taps in kernel order: (dy, dx) = (t / 3 - 1, t % 3 - 1)
          const f32x4 th = pc[sgn * ((t / 3 - 1) * PW + (t % 3 - 1))];
          acc[t][0] = fmaf(wv[j], th[0], acc[t][0]);
          acc[t][1] = fmaf(wv[j], th[1], acc[t][1]);
          acc[t][2] = fmaf(wv[j], th[2], acc[t][2]);
        }
      }
    }
  }
  // pixel lanes of one channel meet in LDS; then one atomic per (tap, thin channel) and wide channel
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    red[tid][3 * t] = acc[t][0];
    red[tid][3 * t + 1] = acc[t][1];
    red[tid][3 * t + 2] = acc[t][2];
  }
  red[tid][27] = bsum;
  __syncthreads();
  {   // outputs in MEMORY order (lanes on consecutive words of a tap's slice: the atomics of a wave share cache lines)
    const int per_tap = L.Cw * L.TH;
    for (int e = tid; e < 9 * per_tap; e += 256) {
      const int t = e / per_tap, rem = e - t * per_tap;
      int cc, th;
      if (L.stride_t == 1) { cc = rem / L.TH; th = rem - cc * L.TH; }     // dw[tap][wide][thin]
      else { th = rem / L.Cw; cc = rem - th * L.Cw; }                     // dw[tap][thin][wide]
      float v = 0.f;
      for (int q = 0; q < NPL; ++q) v += red[q * cw + cc][3 * t + th];
      atomicAdd(&L.dw[(long long)L.tapw[t] * L.w_tap_stride + cc * L.stride_c + th * L.stride_t], L.alpha * v);
    }
    if (L.dbias_wide)
      for (int cc = tid; cc < L.Cw; cc += 256) {
        float v = 0.f;
        for (int q = 0; q < NPL; ++q) v += red[q * cw + cc][27];
        atomicAdd(&L.dbias_wide[cc], v);
      }
  }
  if (L.dbias_thin) {                                      // column sums of the thin operand over this workgroup's own pixels
    f32x4 part = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < npix; i += 256) {
      const int yy = L.lw >= 0 ? i >> L.lw : i / W, x = i - yy * W;
      part += patch[(yy + 1) * PW + x + 1];
    }
    __shared__ float bred[4][4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float v = wave_sum(part[k]);
      if ((tid & 63) == 0) bred[tid >> 6][k] = v;
    }
    __syncthreads();
    // ONE atomic per workgroup and channel: every workgroup of the launch adds to the same <= 3 words, and same-address atomics
    // serialise (four per workgroup made the 128 -> 3 gradient of the headline step 33 us slower)
    if (tid < L.TH) atomicAdd(&L.dbias_thin[tid], (bred[0][tid] + bred[1][tid]) + (bred[2][tid] + bred[3][tid]));
  }
}

static int wgrad_thin_launch(const ThinWgradLaunch& L, const sdm_wgrad_desc* d, size_t lds, hipStream_t st) {
  sdm_prof_pre(st);
  SDM_LAUNCH(wgrad_thin_kernel, dim3((unsigned)((long)d->images * d->Hk / L.R)), dim3(256), lds, st, L);
  sdm_prof_post(st, 4, 2.0 * d->M * d->N * (double)d->images * d->Hk * d->Wk * 9);
  SDM_CHECK_LAUNCH("wgrad_thin");
  return SDM_OK;
}

// the edge-conv form of a weight gradient: 3x3 taps on one pixel grid, one operand of <= 3 channels stored as 4-float rows
static bool wgrad_thin_try(const sdm_wgrad_desc* d, hipStream_t st, int* rc) {
  static const char* sw = getenv("SDM_WGRAD_THIN");        // "0": the matrix-core path for every shape (A/B)
  if (sw && sw[0] == '0') return false;
  if (d->batch != 1 || d->ntaps != 9 || d->g_stride != 1 || d->x_stride != 1 || d->accumulate == 0) return false;
  if (d->x.p1 != nullptr || d->x.scale != nullptr) return false;
  if (d->g.H != d->Hk || d->g.W != d->Wk || d->x.H != d->Hk || d->x.W != d->Wk) return false;
  const bool thin_x = d->N <= 3 && d->M > 4 && d->M <= 256, thin_g = d->M <= 3 && d->N > 4 && d->N <= 256;
  if (!thin_x && !thin_g) return false;
  for (int t = 0; t < 9; ++t)      // taps in kernel order (the table conv_wgrad hands in for a 3x3 stride-1 conv)
    if (d->g_dy[t] != 0 || d->g_dx[t] != 0 || d->x_dy[t] != t / 3 - 1 || d->x_dx[t] != t % 3 - 1) return false;
  const sdm_rows& thin = thin_x ? d->x : d->g;
  const sdm_rows& wide = thin_x ? d->g : d->x;
  if (thin.ld0 != 4 || !aligned16(thin.p0)) return false;
  ThinWgradLaunch L;
  L.wide = wide.p0;
  L.thin = thin.p0;
  L.ld_wide = wide.ld0;
  L.dw = d->dw;
  L.w_tap_stride = d->w_tap_stride;
  L.stride_c = thin_x ? d->ldw : 1;                        // dw[tap][i = gradient channel][j = activation channel]
  L.stride_t = thin_x ? 1 : d->ldw;
  L.dbias_wide = thin_x ? d->dbias : nullptr;
  L.dbias_thin = thin_x ? nullptr : d->dbias;
  L.Cw = thin_x ? d->M : d->N;
  L.TH = thin_x ? d->N : d->M;
  L.images = d->images;
  L.H = d->Hk;
  L.W = d->Wk;
  L.sgn = thin_x ? 1 : -1;   // sum_p G[p] X[p + d]: thin X sits at +d of the wide pixel; thin G at -d of the wide (X) pixel
  L.alpha = d->alpha;
  for (int t = 0; t < 9; ++t) L.tapw[t] = d->tap_w[t];
  L.lw = -1;
  for (int b = 0; b < 30; ++b)
    if ((1 << b) == d->Wk) L.lw = b;
  // R image rows per workgroup: a divisor of H, >= 512 workgroups where the map allows, LDS patch <= 48 KB
  static const char* mw_s = getenv("SDM_WGRAD_THIN_MINWG");      // tuning: fewest workgroups a larger R may leave
  const long minwg = mw_s ? atol(mw_s) : 512;
  int R = 1;
  for (int r = 2; r <= d->Hk; ++r) {
    if (d->Hk % r) continue;
    if ((long)d->images * d->Hk / r < minwg || (size_t)(r + 2) * (d->Wk + 2) * 16 > 32 * 1024) break;
    R = r;
  }
  if ((size_t)(R + 2) * (d->Wk + 2) * 16 > 32 * 1024) return false;
  L.R = R;
  const size_t lds = (size_t)(R + 2) * (d->Wk + 2) * 16;
  sdm_prof_shape(d->M, d->N, (long)d->images * d->Hk * d->Wk, 9, 1, 4.0 * ((double)d->images * d->Hk * d->Wk * (L.Cw + 4) + 9.0 * d->M * d->N));
  *rc = wgrad_thin_launch(L, d, lds, st);
  return true;
}

extern "C" int sdm_wgrad(const sdm_wgrad_desc* d, sdm_stream_t stream) {
  SDM_CHECK_ARG(d != nullptr, "wgrad: null desc");
  SDM_CHECK_ARG(d->M > 0 && d->N > 0 && d->images > 0 && d->Hk > 0 && d->Wk > 0, "wgrad: bad dims");
  SDM_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= SDM_MAX_TAPS, "wgrad: ntaps");
  SDM_CHECK_ARG(d->g.p0 && d->x.p0 && d->dw, "wgrad: null operand");
  SDM_CHECK_ARG(d->batch >= 1 && d->bdiv >= 1, "wgrad: bad batch");
  SDM_CHECK_ARG(d->g.p1 == nullptr && d->g.scale == nullptr, "wgrad: gradient rows cannot be concat/affine");
  SDM_CHECK_ARG(d->x.p1 != nullptr || d->x.csplit >= d->N, "wgrad: csplit < N without second source");
  SDM_CHECK_ARG(d->dbias == nullptr || (d->g_stride == 1 && d->g_dy[0] == 0 && d->g_dx[0] == 0 && d->batch == 1),
                "wgrad: fused bias gradient needs plain gradient rows");
  SDM_CHECK_ARG(d->x.p1 == nullptr || d->x.csplit % 4 == 0, "wgrad: concat split must be a multiple of 4");
  SDM_CHECK_ARG(d->x.scale == nullptr || (d->N % 4 == 0 && d->x.shift && aligned16(d->x.scale) && aligned16(d->x.shift) &&
                                          d->x.aff_nstride % 4 == 0),
                "wgrad: affine-on-load needs N %% 4 == 0 and aligned tables");
  hipStream_t st = (hipStream_t)stream;
  {
    int rc_thin = SDM_OK;
    if (wgrad_thin_try(d, st, &rc_thin)) return rc_thin;
  }
  WgradLaunch L;
  L.d = *d;
  const long P = (long)d->images * d->Hk * d->Wk;
  const int T = cdiv(P, BK);
  // 128x128 tiles when they keep the chip busy on their own, or when split-K over pixels can supply the parallelism
  // while each workgroup still walks >= 32 K-steps (amortises its 64 KB atomic epilogue); else 64x64 tiles
  const long blocks128 = (long)cdiv(d->M, 128) * cdiv(d->N, 128) * d->ntaps * d->batch;
  const bool big = d->M >= 128 && d->N >= 128 && (blocks128 >= 256 || (long)T * blocks128 >= 32L * 512);
  const int bt = big ? 128 : 64;
  const int ntm = cdiv(d->M, bt);
  L.ntn = cdiv(d->N, bt);
  const long blocks = (long)ntm * L.ntn * d->ntaps * d->batch;
  // Split over pixels by a small cost model (same reasoning as wgrad_bf16 in sdm_bf16_train.hip): tiles x taps x slices
  // is rarely a multiple of the resident-workgroup slots, and a 12 %-full trailing wave costs a whole wave.  Cost =
  // ceil(workgroups / slots) x (K-steps per slice x step time + fixed) + fp32 atomics of the extra slices at ~1.3 TB/s.
  int sk = 1;
  {
    static const char* legacy = getenv("SDM_WGRAD_PLAN");     // "0": the round-1 rule (A/B)
    if (legacy && legacy[0] == '0') {
      if (blocks < 512 && T >= 16) {
        sk = (int)((768 + blocks - 1) / blocks);
        if (sk > T / 8) sk = T / 8;
        if (sk > 128) sk = 128;
        if (sk < 1) sk = 1;
      }
    } else {
      const double slots = big ? 512.0 : 768.0;
      const double t_step = big ? 3.7e-6 : 1.3e-6;            // one 32-pixel K-step of a full wave of workgroups
      const double out_bytes = (double)d->M * d->N * d->ntaps * d->batch * 4.0;
      double best = 1e30;
      for (int c = 1; c <= 128 && c <= (T >= 8 ? T / 8 : 1); ++c) {
        const double waves = ceil((double)blocks * c / slots);
        const double steps = ceil((double)T / c);
        const double t = waves * (steps * t_step + 3.0e-6) + (c > 1 || d->accumulate == 1 ? out_bytes * c / 1.3e12 : out_bytes / 5e12);
        if (t < best * 0.97) { best = t; sk = c; }
      }
    }
  }
  L.splitk = sk;
  L.atomic_out = (sk > 1 || d->accumulate == 1) ? 1 : 0;
  const bool vecG = aligned16(d->g.p0) && d->g.ld0 % 4 == 0 && d->g_bs0 % 4 == 0 && d->g_bs1 % 4 == 0 && d->M % 4 == 0;
  const bool vecX = aligned16(d->x.p0) && d->x.ld0 % 4 == 0 && d->x_bs0 % 4 == 0 && d->x_bs1 % 4 == 0 && d->N % 4 == 0 &&
                    (d->x.p1 == nullptr || (aligned16(d->x.p1) && d->x.ld1 % 4 == 0));
  const bool vec = vecG && vecX;
  if (sk > 1 && d->accumulate == 0) {
    // one memset when the tap slices tile a contiguous block (conv / linear weights), else per slice
    bool contiguous = d->batch == 1 && d->ldw == d->N && (d->ntaps == 1 || d->w_tap_stride == (int64_t)d->M * d->N);
    if (contiguous) {
      int lo = 255, hi = 0;
      for (int t = 0; t < d->ntaps; ++t) { lo = d->tap_w[t] < lo ? d->tap_w[t] : lo; hi = d->tap_w[t] > hi ? d->tap_w[t] : hi; }
      contiguous = (hi - lo + 1) == d->ntaps;
      if (contiguous) SDM_MEMSET0(d->dw + (long long)lo * d->w_tap_stride, (size_t)d->ntaps * d->M * d->N * sizeof(float), st);
    }
    if (!contiguous) {
      for (int b = 0; b < d->batch; ++b)
        for (int t = 0; t < d->ntaps; ++t) {
          float* p = d->dw + (b / d->bdiv) * d->w_bs0 + (b % d->bdiv) * d->w_bs1 + (long long)d->tap_w[t] * d->w_tap_stride;
          SDM_LAUNCH(zero_rows_kernel, dim3(cdiv((long)d->M * d->N, 1024)), dim3(256), 0, st, p, (long long)d->M,
                             d->N, (long long)d->ldw);
          SDM_CHECK_LAUNCH("zero_rows");
        }
    }
  }
  const bool aff = d->x.scale != nullptr;
  static const char* noglds = getenv("SDM_IGEMM_GLDS");
  if (vec && !aff && !(noglds && noglds[0] == '0'))
    return big ? launch_wgrad_glds<128, 2, 2, 2>(L, ntm, st) : launch_wgrad_glds<64, 2, 2, 3>(L, ntm, st);
  if (big) {
    if (aff) return vec ? launch_wgrad<128, 2, 2, true, true>(L, ntm, st) : launch_wgrad<128, 2, 2, false, true>(L, ntm, st);
    return vec ? launch_wgrad<128, 2, 2, true, false>(L, ntm, st) : launch_wgrad<128, 2, 2, false, false>(L, ntm, st);
  }
  if (aff) return vec ? launch_wgrad<64, 2, 2, true, true>(L, ntm, st) : launch_wgrad<64, 2, 2, false, true>(L, ntm, st);
  return vec ? launch_wgrad<64, 2, 2, true, false>(L, ntm, st) : launch_wgrad<64, 2, 2, false, false>(L, ntm, st);
}
