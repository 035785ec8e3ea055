// bf16 TRAINING path (BASELINE configs 4-5; reference analogue: fp16 autocast, train_diffusion.py:333,358-364).
// fp32 master weights and fp32 weight gradients / optimizer; activations, activation gradients and the weight copies the
// contractions read are bf16; every sum is fp32 (matrix-core accumulators, GroupNorm statistics, softmax, bias sums).
//
//   forward, data gradient : sdm_igemm_bf16 (sdm_bf16.hip) -- the data gradient contracts against a TRANSPOSED bf16 weight
//                            copy [tap][c_in][c_out] (sdm_cast_bf16_t) so that both operands stay reduction-contiguous
//   weight gradient        : wgrad_bf16_kernel below -- the reduction runs over PIXELS, so both operands are K-major
//                            [pixel][channel] row tiles; they go global -> LDS by LDS-DMA exactly as they lie in HBM and the
//                            matrix-core fragments (8 consecutive k per lane) are read with ds_read_b64_tr_b16, the gfx950
//                            transposing LDS read (two per fragment), from an XOR-swizzled image that keeps them conflict-free
//   pointwise              : GroupNorm/AdaGN backward, column-softmax backward, SiLU backward, edge-conv gradients on bf16 rows
#include <math.h>
#include <stdlib.h>
#include <type_traits>

#define SDM_FAST_SIGMOID
#include "sdm_bf16_int.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ float g_zero_chunk_wg[4];

__device__ __forceinline__ int xcd_contiguous_wg(int bid, int total) {   // see sdm_igemm.hip
  const int q = total >> 3, r = total & 7, x = bid & 7, i = bid >> 3;
  return x < r ? x * (q + 1) + i : r * (q + 1) + (x - r) * q + i;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_wg() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The transposing LDS read as inline asm (base address + immediate byte offset).  Through the builtin the compiler sees an
// LDS load next to LDS-DMA stores it cannot disambiguate and puts s_waitcnt vmcnt(0) in front of the first read of every
// K-step: every stage in flight is drained and the counted vmcnt waits of these kernels meant nothing (rounds 2-3; found in
// round 4 from the ISA).  The compiler does not know when an asm read lands: every use sits behind an explicit
// s_waitcnt lgkmcnt(0) and an opaque touch of the registers.
template <int OFF>
__device__ __forceinline__ void tr8_wg(s16x4& v, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void cfor_wg(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    cfor_wg<I + 1, N>(f);
  }
}

struct WgradBfLaunch {
  sdm_wgrad_desc d;
  int splitk, ntn, atomic_out;
  int rows_per_step;   // FASTP: image rows covered by one 64-pixel K-step (64 / Wk when Wk < 64, else 1)
};

#define BKP 64   // pixels per K-step

// dw[tap][i][j] (+)= alpha * sum_p G[gpix(p,tap)][i] * X[xpix(p,tap)][j]      G, X: bf16 rows; dw: fp32
//
// LDS image of one operand stage: [BKP pixel rows][BT channels] bf16, row pitch 2*BT bytes, filled lane-linearly by
// global_load_lds_dwordx4 (one wave-instruction = 1 KiB = 4 rows of 256 B at BT=128, 8 rows of 128 B at BT=64).  The
// 16-byte chunk index is XOR-swizzled with f(k) on the per-lane SOURCE address and again on the read:
//   BT=128: f(k) = (k&3)<<2      BT=64: f(k) = ((k>>1)&1)<<2
// so that the 4 pixel rows x 64 bytes one 32-lane half of a ds_read_b64_tr_b16 touches fall into four different
// quarters of the 256-byte bank line (conflict-free; unswizzled they are 4-way / 2-way).
// Fragment of v_mfma_f32_32x32x16_bf16 (lane l: r=l&31, h=l>>5 holds k = 8h..8h+7 of channel r): two transposing reads;
// in each, the 16-lane group (l>>4) covers pixel rows kb..kb+3 (kb = 16*kk + 8h + 4*rd) x channels c0+16*((l>>4)&1)..+15,
// lane 4q+p of the group supplies the address of row kb+q, channels +4p..+4p+3, and receives channel (l&15)'s 4 rows.
// FASTP (the host sets it when a 64-pixel K-step is a whole number of image rows or an aligned piece of one row, i.e.
// Wk % 64 == 0, or 64 % Wk == 0 and Hk % (64 / Wk) == 0 -- every power-of-two feature map of >= 64 pixels): the position
// of a K-step is then WAVE-UNIFORM scalar state (image, first row, first column) advanced with scalar adds, and a lane's
// pixel inside the step is a lane CONSTANT; a DMA address is one 64-bit add and its padding test two compares.  The
// general walk (per-lane pixel coordinates with carries, two 64-bit multiplies per row) costs ~16 vector instructions per
// matrix instruction and made the kernel issue-bound (profiles/r02_wgrad_bf16_512_pmc.json, round 2 first cut: 45 % of
// wave time issuing, matrix pipe 20 % busy).
template <int BT, int WM, int WN, int NBUF, bool FASTP>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_bf16_kernel(const WgradBfLaunch L) {
  const sdm_wgrad_desc& d = L.d;
  constexpr int NT = 64 * WM * WN;              // WM x WN waves, each owning (BT/WM) x (BT/WN)
  constexpr int TM = BT / (32 * WM), TN = BT / (32 * WN);     // 32x32 MFMA tiles per wave
  constexpr int ROWB = 2 * BT;                  // bytes per LDS row
  constexpr int CPR = BT / 8;                   // 16-byte chunks per row
  constexpr int RPP = NT / CPR;                 // pixel rows per DMA pass of the NT threads
  static_assert(BKP % RPP == 0 && (RPP * ROWB) == (NT / 64) * 1024, "one DMA pass = one KiB per wave");
  constexpr int IT = BKP / RPP;                 // passes per operand per K-step
  constexpr int TILEB = BKP * ROWB;             // bytes per operand stage
  constexpr int G = 2 * IT;
  constexpr int PD = NBUF - 1;
  extern __shared__ __align__(16) char smem_wg[];
  char* As = smem_wg;
  char* Bs = smem_wg + NBUF * TILEB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, h = lane >> 5;
  // 1-D grid of (pixel slice x tile) x taps: the taps of one (slice, tile) read the same gradient rows and activation rows a
  // pixel / an image row apart, so they are neighbours of one XCD's contiguous run and meet in one L2 (as separate grid rows
  // every tap streamed its operands from HBM: 3.0x algorithmic, profiles/r02_wgrad_bf16_512_tile256_pmc.json)
  const int c_ = xcd_contiguous_wg(blockIdx.x, gridDim.x);
  const int tp = c_ % d.ntaps, wg_ = c_ / d.ntaps;
  const int ntiles_ = (gridDim.x / d.ntaps) / L.splitk;
  const int ks = wg_ / ntiles_, tile_ = wg_ - ks * ntiles_;
  const int tile_n = tile_ % L.ntn, tile_m = tile_ / L.ntn;
  const int m0 = tile_m * BT, n0 = tile_n * BT;
  const int b0 = blockIdx.z / d.bdiv, b1 = blockIdx.z % d.bdiv;
  const bf16_t* g0 = (const bf16_t*)d.g.p0 + b0 * d.g_bs0 + b1 * d.g_bs1;
  const bf16_t* x0 = (const bf16_t*)d.x.p0 + b0 * d.x_bs0 + b1 * d.x_bs1;
  const bf16_t* x1 = (const bf16_t*)d.x.p1;
  const bf16_t* zero = (const bf16_t*)g_zero_chunk_wg;
  const int HW = d.Hk * d.Wk;
  const long long P = (long long)d.images * HW;
  const int T = (int)((P + BKP - 1) / BKP);
  const int it0 = (int)((long long)T * ks / L.splitk), it1 = (int)((long long)T * (ks + 1) / L.splitk);
  const int gdy = d.g_dy[tp], gdx = d.g_dx[tp], xdy = d.x_dy[tp], xdx = d.x_dx[tp];
  // DMA role of this thread: pixel rows (tid / CPR) + RPP*i, physical chunk slot tid % CPR -> logical chunk slot ^ f(row)
  const int drow = tid / CPR;
  const int fk = BT >= 128 ? ((drow & 3) << 2) : (((drow >> 1) & 1) << 2);
  const int lchunk = (tid % CPR) ^ fk;
  const int cm = m0 + 8 * lchunk, cn = n0 + 8 * lchunk;
  const bool gch = cm < ((d.M + 7) & ~7), xch = cn < ((d.N + 7) & ~7);
  const bool xsrc0 = cn < d.x.csplit;
  const bool do_bias = d.dbias != nullptr && tile_n == 0 && tp == 0;   // block-uniform
  float bacc = 0.f;

  int pn[IT], py[IT], px[IT];
  const int adv_x = BKP % d.Wk, qw = BKP / d.Wk;
  const int adv_y = qw % d.Hk, adv_n = qw / d.Hk;
  // FASTP state: uniform (un, uy, ux) = image / first row / first column of the next K-step to issue; lane constants:
  // row and column of this lane's pixels inside a step, their tap-shifted source coordinates minus the uniform part,
  // and their element offsets minus the uniform part
  int un = 0, uy = 0, ux = 0;
  int lgy[IT], lgx[IT], lxy[IT], lxx[IT];
  long long lgo[IT], lxo[IT];
  if (FASTP) {
    const long long p0 = (long long)it0 * BKP;
    un = (int)(p0 / HW);
    const int rem = (int)(p0 - (long long)un * HW);
    uy = rem / d.Wk;
    ux = rem - uy * d.Wk;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int r = drow + RPP * i;
      const int ly = L.rows_per_step > 1 ? r / d.Wk : 0, lx = L.rows_per_step > 1 ? r - ly * d.Wk : r;
      lgy[i] = ly * d.g_stride + gdy;
      lgx[i] = lx * d.g_stride + gdx;
      lxy[i] = ly * d.x_stride + xdy;
      lxx[i] = lx * d.x_stride + xdx;
      lgo[i] = ((long long)lgy[i] * d.g.W + lgx[i]) * d.g.ld0 + cm;
      lxo[i] = xsrc0 ? ((long long)lxy[i] * d.x.W + lxx[i]) * d.x.ld0 + cn
                     : ((long long)lxy[i] * d.x.W + lxx[i]) * d.x.ld1 + (cn - d.x.csplit);
    }
  } else {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const long long p = (long long)it0 * BKP + drow + RPP * i;
      const int n = (int)(p / HW), rem = (int)(p - (long long)n * HW);
      pn[i] = n;
      py[i] = rem / d.Wk;
      px[i] = rem - py[i] * d.Wk;
    }
  }
  const bf16_t* xb = xsrc0 ? x0 : x1;
  const long long xld = xsrc0 ? d.x.ld0 : d.x.ld1;

  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  int issued = it0;
  auto issue_next = [&]() {
    if (issued < it1) {
      const int buf = (issued - it0) % NBUF;
      char* Ad = As + buf * TILEB + __builtin_amdgcn_readfirstlane(wave) * 1024;
      char* Bd = Bs + buf * TILEB + __builtin_amdgcn_readfirstlane(wave) * 1024;
      if (FASTP) {
        // uniform part of the source coordinates / offsets of this K-step (scalar registers)
        const int sgy = uy * d.g_stride, sgx = ux * d.g_stride, sxy = uy * d.x_stride, sxx = ux * d.x_stride;
        const long long sgo = (((long long)un * d.g.H + sgy) * d.g.W + sgx) * d.g.ld0;
        const long long sxo = (((long long)un * d.x.H + sxy) * d.x.W + sxx) * xld;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
          const bool gu = gch && (unsigned)(sgy + lgy[i]) < (unsigned)d.g.H && (unsigned)(sgx + lgx[i]) < (unsigned)d.g.W;
          const bool xu = xch && (unsigned)(sxy + lxy[i]) < (unsigned)d.x.H && (unsigned)(sxx + lxx[i]) < (unsigned)d.x.W;
          const bf16_t* gq = gu ? g0 + (sgo + lgo[i]) : zero;
          const bf16_t* xq = xu ? xb + (sxo + lxo[i]) : zero;
          __builtin_amdgcn_global_load_lds((glb_ptr)gq, (lds_ptr)(Ad + i * RPP * ROWB), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((glb_ptr)xq, (lds_ptr)(Bd + i * RPP * ROWB), 16, 0, 0);
        }
        if (L.rows_per_step > 1) {
          uy += L.rows_per_step;
        } else {
          ux += BKP;
          if (ux >= d.Wk) { ux = 0; uy += 1; }
        }
        if (uy >= d.Hk) { uy = 0; un += 1; }
        ++issued;
        return;
      }
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int n = pn[i], y = py[i], x = px[i];
        const bool in = n < d.images;
        const int gy = y * d.g_stride + gdy, gx = x * d.g_stride + gdx;
        const int xy = y * d.x_stride + xdy, xx = x * d.x_stride + xdx;
        const bool gu = in && gch && (unsigned)gy < (unsigned)d.g.H && (unsigned)gx < (unsigned)d.g.W;
        const bool xu = in && xch && (unsigned)xy < (unsigned)d.x.H && (unsigned)xx < (unsigned)d.x.W;
        const long long xrow = ((long long)n * d.x.H + xy) * d.x.W + xx;
        const bf16_t* gq = g0 + (((long long)n * d.g.H + gy) * d.g.W + gx) * d.g.ld0 + cm;
        const bf16_t* xq = xsrc0 ? x0 + xrow * d.x.ld0 + cn : x1 + xrow * d.x.ld1 + (cn - d.x.csplit);
        gq = gu ? gq : zero;
        xq = xu ? xq : zero;
        __builtin_amdgcn_global_load_lds((glb_ptr)gq, (lds_ptr)(Ad + i * RPP * ROWB), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr)xq, (lds_ptr)(Bd + i * RPP * ROWB), 16, 0, 0);
        int nx = x + adv_x, ny = y + adv_y, nn = n + adv_n;
        if (nx >= d.Wk) { nx -= d.Wk; ny += 1; }
        if (ny >= d.Hk) { ny -= d.Hk; nn += 1; }
        px[i] = nx; py[i] = ny; pn[i] = nn;
      }
    }
    ++issued;
  };

  // transposing-read role of this lane (constant over K-steps): row q of its group's 4-row block, channel quad p
  const int i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3, chalf = (lane >> 4) & 1;
  const int fq = BT >= 128 ? (q << 2) : ((q >> 1) << 2);
  int aoff[TM], boff[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int col = wm * TM * 32 + i * 32 + 16 * chalf + 4 * p4;
    aoff[i] = (8 * h + q) * ROWB + (((col >> 3) ^ fq) << 4) + ((col & 4) << 1);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = wn * TN * 32 + j * 32 + 16 * chalf + 4 * p4;
    boff[j] = (8 * h + q) * ROWB + (((col >> 3) ^ fq) << 4) + ((col & 4) << 1);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr)smem_wg;
#pragma unroll
  for (int pp = 0; pp < PD; ++pp) issue_next();
  for (int it = it0; it < it1; ++it) {
    const int buf = (it - it0) % NBUF;
    const int younger = (it1 - 1 - it) < (PD - 1) ? (it1 - 1 - it) : (PD - 1);
    if (younger >= 2) wait_vmcnt_wg<2 * G>();
    else if (younger == 1) wait_vmcnt_wg<G>();
    else wait_vmcnt_wg<0>();
    __builtin_amdgcn_s_barrier();
    issue_next();
    const char* Ab = As + buf * TILEB;
    if (do_bias && tid < BT) {   // column sums of the gradient tile (bias gradient); channel tid
      float a = 0.f;
#pragma unroll 8
      for (int k = 0; k < BKP; ++k) {
        const int f = BT >= 128 ? ((k & 3) << 2) : (((k >> 1) & 1) << 2);
        a += bf2f(*(const bf16_t*)(Ab + k * ROWB + (((tid >> 3) ^ f) << 4) + ((tid & 7) << 1)));
      }
      bacc += a;
    }
    const unsigned Aa = lds_base + buf * TILEB, Ba = lds_base + (NBUF + buf) * TILEB;
    cfor_wg<0, BKP / 16>([&](auto kc) {
      constexpr int kk = decltype(kc)::value;
      s16x4 al[TM], ah[TM], bl[TN], bh[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        tr8_wg<(kk * 16) * ROWB>(al[i], Aa + aoff[i]);
        tr8_wg<(kk * 16 + 4) * ROWB>(ah[i], Aa + aoff[i]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        tr8_wg<(kk * 16) * ROWB>(bl[j], Ba + boff[j]);
        tr8_wg<(kk * 16 + 4) * ROWB>(bh[j], Ba + boff[j]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      s16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        asm volatile("" : "+v"(al[i]), "+v"(ah[i]));
        af[i] = __builtin_shufflevector(al[i], ah[i], 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        asm volatile("" : "+v"(bl[j]), "+v"(bh[j]));
        bf[j] = __builtin_shufflevector(bl[j], bh[j], 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                              __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
    });
  }
  if (do_bias && tid < BT && m0 + tid < d.M) atomicAdd(&d.dbias[m0 + tid], bacc);

  float* dw = d.dw + b0 * d.w_bs0 + b1 * d.w_bs1 + (long long)d.tap_w[tp] * d.w_tap_stride;
  const int l31 = lane & 31;
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
        float* o = dw + (long long)row * d.ldw + col;
        const float v = acc[i][j][r] * d.alpha;
        if (L.atomic_out) atomicAdd(o, v);
        else *o = v;
      }
    }
}

template <int BT, int WM, int WN, int NBUF, bool FASTP>
static int launch_wgrad_bf16(const WgradBfLaunch& L, int ntm, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * NBUF * BKP * 2 * BT;
  dim3 grid(ntm * L.ntn * L.splitk * L.d.ntaps, 1, L.d.batch);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)wgrad_bf16_kernel<BT, WM, WN, NBUF, FASTP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    attr_set = true;
  }
  {
    const double px = (double)L.d.images;
    const double g_b = px * L.d.g.H * L.d.g.W * L.d.M, x_b = px * L.d.x.H * L.d.x.W * L.d.N, w_b = (double)L.d.M * L.d.N * L.d.ntaps;
    sdm_prof_shape(L.d.M, L.d.N, (int)((long)L.d.images * L.d.Hk * L.d.Wk), L.d.ntaps, L.splitk,
                   L.d.batch * (2.0 * (g_b + x_b) + 4.0 * w_b));
  }
  sdm_prof_pre(st);
  SDM_LAUNCH((wgrad_bf16_kernel<BT, WM, WN, NBUF, FASTP>), grid, dim3(64 * WM * WN), lds, st, L);
  sdm_prof_post(st, BT >= 128 ? 3 : 4, 2.0 * L.d.M * L.d.N * (double)L.d.images * L.d.Hk * L.d.Wk * L.d.ntaps * L.d.batch);
  SDM_CHECK_LAUNCH("wgrad_bf16");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// Three taps per workgroup for plain 3x3 stride-1 weight gradients on feature maps at least 64 pixels wide.
// wgrad_bf16_kernel gives every tap its own workgroups, each streaming the same gradient tile and an activation tile
// shifted by one pixel; what bounds these kernels is the volume moved L2 -> LDS (profiles/r02_bf16_tiles.txt).  Here a
// workgroup owns one kernel ROW (dy) of a 128x128 (c_out x c_in) tile: per 64-pixel K-step it loads the gradient rows once
// and the 66 activation pixels x0-1 .. x0+64 of image row y+dy once, and contracts the three column taps dx = -1, 0, +1
// from the same LDS image (fragment rows shifted by 1 + dx, swizzle taken from the shifted row): a third of the bytes per
// flop, 48 instead of 16 MFMAs per wave and barrier, 1.33 instead of 2 transposing reads per MFMA.
// --------------------------------------------------------------------------------------------
// (Wgrad3Launch: sdm_bf16_int.h)

#define W3_XROWS 68     // 66 activation pixels per K-step, padded to whole 4-row DMA instructions

// WSH: 6 = maps at least 64 pixels wide (a K-step is a piece of one image row); 5 / 4 = 32 / 16 pixels wide: a K-step is
// 2 / 4 whole rows, the activation "row" is the same 66 consecutive pixels of the (dy-shifted) linear pixel index, and the
// products whose gradient pixel has no left / right neighbour are dropped by zeroing fixed fragment elements (pixel k of a
// step sits in element k & 7 of k-group k >> 4, half (k >> 3) & 1: x == 0 <=> k % W == 0 is a compile-time position).
template <int WSH>
__global__ __launch_bounds__(256, 2) void wgrad3_bf16_kernel(const Wgrad3Launch L) {
  const sdm_wgrad_desc& d = L.d;
  constexpr int BT = 128, TM = 2, TN = 2, ROWB = 2 * BT, NBUF = 2;
  constexpr int GT = BKP * ROWB, XT = W3_XROWS * ROWB;          // bytes per stage: 16 KiB + 17 KiB
  extern __shared__ __align__(16) char smem_w3[];
  char* Gs = smem_w3;
  char* Xs = smem_w3 + NBUF * GT;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5;
  const int c_ = xcd_contiguous_wg(blockIdx.x, gridDim.x);    // the three kernel rows of a (slice, tile): neighbours on one XCD
  const int dyi = c_ % 3, wg_ = c_ / 3;
  const int ntiles_ = (gridDim.x / 3) / L.splitk;
  const int ks = wg_ / ntiles_, tile_ = wg_ - ks * ntiles_;
  const int tile_n = tile_ % L.ntn, tile_m = tile_ / L.ntn;
  const int dy = L.dy[dyi];
  const int m0 = tile_m * BT, n0 = tile_n * BT;
  const bf16_t* g0 = (const bf16_t*)d.g.p0;
  const bf16_t* x0 = (const bf16_t*)d.x.p0;
  const bf16_t* x1 = (const bf16_t*)d.x.p1;
  const bf16_t* zero = (const bf16_t*)g_zero_chunk_wg;
  const int H = d.Hk, W = d.Wk, HW = H * W;
  const long long P = (long long)d.images * HW;
  const int T = (int)(P / BKP);                                 // W % 64 == 0: K-steps are whole 64-pixel row pieces
  const int it0 = (int)((long long)T * ks / L.splitk), it1 = (int)((long long)T * (ks + 1) / L.splitk);
  // DMA roles: 16 chunks per 256-byte row, 16 rows per pass of the 256 threads
  const int drow = tid >> 4;
  const int cm = m0 + 8 * ((tid & 15) ^ ((drow & 3) << 2));      // gradient rows: swizzle by the pixel row (drow + 16 i: same & 3)
  const bool gch = cm < ((d.M + 7) & ~7);
  const bool do_bias = d.dbias != nullptr && tile_n == 0 && dyi == 0;
  float bacc = 0.f;
  // activation patch row r <-> pixel x0 - 1 + r; rows r = drow + 16 i (i = 0..3) and, wave 0 only, 64 + (tid >> 4)
  // (r & 3) == (drow & 3) for all of them, so one logical chunk column serves every pass
  const int cn = n0 + 8 * ((tid & 15) ^ ((drow & 3) << 2));
  const bool xch = cn < ((d.N + 7) & ~7);
  const bool xsrc0 = cn < d.x.csplit;
  const bf16_t* xb = xsrc0 ? x0 : x1;
  const long long xld = xsrc0 ? d.x.ld0 : d.x.ld1;
  const int xcol = xsrc0 ? cn : cn - d.x.csplit;

  int un, uy, ux;
  {
    const long long p0 = (long long)it0 * BKP;
    un = (int)(p0 / HW);
    const int rem = (int)(p0 - (long long)un * HW);
    uy = rem / W;
    ux = rem - uy * W;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  int issued = it0;
  auto issue_next = [&]() {
    if (issued < it1) {
      const int buf = (issued - it0) & 1;
      char* Gd = Gs + buf * GT + wave * 1024;
      char* Xd = Xs + buf * XT + wave * 1024;
      const long long grow0 = ((long long)un * H + uy) * W + ux;                 // gradient pixel of row 0 of the step
      const int xy = uy + dy;
      const bool yok = (unsigned)xy < (unsigned)H;
      const long long xrow0 = ((long long)un * H + xy) * W + ux - 1;              // activation pixel of patch row 0
      const int lin0 = xy * W + ux - 1;                                          // ... as a pixel index inside its image
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = drow + 16 * i;
        const bf16_t* gq = gch ? g0 + (grow0 + r) * d.g.ld0 + cm : zero;
        const int px = ux - 1 + r;
        const bool xu = WSH >= 6 ? (xch && yok && (unsigned)px < (unsigned)W) : (xch && (unsigned)(lin0 + r) < (unsigned)HW);
        const bf16_t* xq = xu ? xb + (xrow0 + r) * xld + xcol : zero;
        __builtin_amdgcn_global_load_lds((glb_ptr)gq, (lds_ptr)(Gd + i * 16 * ROWB), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr)xq, (lds_ptr)(Xd + i * 16 * ROWB), 16, 0, 0);
      }
      if (wave == 0) {                          // patch rows 64..67 (pixels x0+63 .. x0+66; 66, 67 are never read: zeros)
        const int r = 64 + drow;                // drow = 0..3 in wave 0
        const int px = ux - 1 + r;
        const bool xu = r < 66 && (WSH >= 6 ? (xch && yok && (unsigned)px < (unsigned)W) : (xch && (unsigned)(lin0 + r) < (unsigned)HW));
        const bf16_t* xq = xu ? xb + (xrow0 + r) * xld + xcol : zero;
        __builtin_amdgcn_global_load_lds((glb_ptr)xq, (lds_ptr)(Xs + buf * XT + 64 * ROWB), 16, 0, 0);
      }
      if (WSH >= 6) {
        ux += BKP;
        if (ux >= W) { ux = 0; uy += 1; }
      } else {
        uy += BKP >> WSH;
      }
      if (uy >= H) { uy = 0; un += 1; }
    }
    ++issued;
  };

  // transposing-read roles (see wgrad_bf16_kernel): 16-lane group row q, channel quad p4
  const int i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3, chalf = (lane >> 4) & 1;
  int aoff[TM], boff[3][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int col = wm * TM * 32 + i * 32 + 16 * chalf + 4 * p4;
    aoff[i] = (8 * h + q) * ROWB + (((col >> 3) ^ (q << 2)) << 4) + ((col & 4) << 1);
  }
#pragma unroll
  for (int sft = 0; sft < 3; ++sft)              // patch rows shifted by 1 + dx = 0, 1, 2
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = wn * TN * 32 + j * 32 + 16 * chalf + 4 * p4;
      boff[sft][j] = (8 * h + q + sft) * ROWB + (((col >> 3) ^ (((q + sft) & 3) << 2)) << 4) + ((col & 4) << 1);
    }

  f32x16 acc[3][TM][TN];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr)smem_w3;
  issue_next();
  for (int it = it0; it < it1; ++it) {
    const int buf = (it - it0) & 1;
    wait_vmcnt_wg<0>();
    __builtin_amdgcn_s_barrier();
    issue_next();
    const char* Ab = Gs + buf * GT;
    if (do_bias && tid < BT) {
      float a = 0.f;
#pragma unroll 8
      for (int k = 0; k < BKP; ++k) a += bf2f(*(const bf16_t*)(Ab + k * ROWB + (((tid >> 3) ^ ((k & 3) << 2)) << 4) + ((tid & 7) << 1)));
      bacc += a;
    }
    const unsigned Aa = lds_base + buf * GT, Ba = lds_base + NBUF * GT + buf * XT;
    cfor_wg<0, BKP / 16>([&](auto kc) {
      constexpr int kk = decltype(kc)::value;
      s16x4 al[TM], ah[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        tr8_wg<(kk * 16) * ROWB>(al[i], Aa + aoff[i]);
        tr8_wg<(kk * 16 + 4) * ROWB>(ah[i], Aa + aoff[i]);
      }
      s16x8 af[TM];
#pragma unroll
      for (int sft = 0; sft < 3; ++sft) {
        s16x4 bl[TN], bh[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          tr8_wg<(kk * 16) * ROWB>(bl[j], Ba + boff[sft][j]);
          tr8_wg<(kk * 16 + 4) * ROWB>(bh[j], Ba + boff[sft][j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (sft == 0) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            asm volatile("" : "+v"(al[i]), "+v"(ah[i]));
            af[i] = __builtin_shufflevector(al[i], ah[i], 0, 1, 2, 3, 4, 5, 6, 7);
          }
        }
        s16x8 bf[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          asm volatile("" : "+v"(bl[j]), "+v"(bh[j]));
          bf[j] = __builtin_shufflevector(bl[j], bh[j], 0, 1, 2, 3, 4, 5, 6, 7);
        }
        if (WSH < 6 && sft != 1) {          // narrow maps: gradient pixels at x == 0 (dx = -1) / x == W - 1 (dx = +1) have no neighbour
          const bool grp = WSH == 4 || (sft == 0 ? (kk & 1) == 0 : (kk & 1) == 1);
          if (grp && h == (sft == 0 ? 0 : 1)) {
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j][sft == 0 ? 0 : 7] = 0;
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[sft][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                     __builtin_bit_cast(bf16x8, bf[j]), acc[sft][i][j], 0, 0, 0);
      }
    });
  }
  if (do_bias && tid < BT && m0 + tid < d.M) atomicAdd(&d.dbias[m0 + tid], bacc);

  const int l31 = lane & 31;
#pragma unroll
  for (int sft = 0; sft < 3; ++sft) {
    float* dw = d.dw + (long long)L.tapw[dyi][sft] * d.w_tap_stride;
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
          float* o = dw + (long long)row * d.ldw + col;
          const float v = acc[sft][i][j][r] * d.alpha;
          if (L.atomic_out) atomicAdd(o, v);
          else *o = v;
        }
      }
  }
}

__global__ void zero_rows_f32_kernel(float* p, long long rows, int cols, long long ld) {
  const long long total = rows * cols;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = e / cols;
    p[r * ld + (e - r * cols)] = 0.f;
  }
}

extern "C" int sdm_wgrad_bf16(const sdm_wgrad_desc* d, sdm_stream_t stream) {
  SDM_CHECK_ARG(d != nullptr, "wgrad_bf16: null desc");
  SDM_CHECK_ARG(d->M > 0 && d->N > 0 && d->images > 0 && d->Hk > 0 && d->Wk > 0, "wgrad_bf16: bad dims");
  SDM_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= SDM_MAX_TAPS, "wgrad_bf16: ntaps");
  SDM_CHECK_ARG(d->g.p0 && d->x.p0 && d->dw, "wgrad_bf16: null operand");
  SDM_CHECK_ARG(d->batch >= 1 && d->bdiv >= 1, "wgrad_bf16: bad batch");
  SDM_CHECK_ARG(d->g.p1 == nullptr && d->g.scale == nullptr && d->x.scale == nullptr,
                "wgrad_bf16: gradient rows cannot be concat/affine, activation rows carry no pending affine");
  SDM_CHECK_ARG(d->x.p1 != nullptr || d->x.csplit >= d->N, "wgrad_bf16: csplit < N without second source");
  SDM_CHECK_ARG(d->x.p1 == nullptr || d->x.csplit % 8 == 0, "wgrad_bf16: concat split must be a multiple of 8");
  SDM_CHECK_ARG(d->dbias == nullptr || (d->g_stride == 1 && d->g_dy[0] == 0 && d->g_dx[0] == 0 && d->batch == 1),
                "wgrad_bf16: fused bias gradient needs plain gradient rows");
  // rows are read in whole 16-byte chunks of 8 channels: pitches / batch strides multiples of 8 elements, and each row
  // must physically hold its channel count rounded up to 8 (attention probabilities are padded like that)
  SDM_CHECK_ARG(aligned16(d->g.p0) && d->g.ld0 % 8 == 0 && d->g_bs0 % 8 == 0 && d->g_bs1 % 8 == 0 && d->g.ld0 >= ((d->M + 7) & ~7),
                "wgrad_bf16: gradient rows must be 16-byte aligned with pitch >= M rounded up to 8");
  SDM_CHECK_ARG(aligned16(d->x.p0) && d->x.ld0 % 8 == 0 && d->x_bs0 % 8 == 0 && d->x_bs1 % 8 == 0 &&
                    (d->x.p1 == nullptr || (aligned16(d->x.p1) && d->x.ld1 % 8 == 0)) &&
                    (d->x.p1 != nullptr || d->x.ld0 >= ((d->N + 7) & ~7)),
                "wgrad_bf16: activation rows must be 16-byte aligned with pitch >= N rounded up to 8");
  SDM_CHECK_ARG(d->x.p1 == nullptr || d->N % 8 == 0, "wgrad_bf16: concatenated rows need N %% 8 == 0");
  hipStream_t st = (hipStream_t)stream;
  WgradBfLaunch L;
  L.d = *d;
  const long P = (long)d->images * d->Hk * d->Wk;
  const int T = cdiv(P, BKP);
  // 128x128 tiles whenever the problem can feed them (the 64x64 form is LDS-read bound: 4 transposing reads per MFMA)
  const long blocks128 = (long)cdiv(d->M, 128) * cdiv(d->N, 128) * d->ntaps * d->batch;
  const bool big = d->M > 64 && d->N > 64 && (blocks128 >= 128 || (long)T * blocks128 >= 8L * 256);
  // Tile and split over pixels by a small cost model.  The grid is tiles x 9 taps x slices, which is rarely a multiple of
  // the chip's resident-workgroup slots (1 per CU at 256x256, 2 at 128x128, 3 at 64x64): 576 workgroups of a 1024x1024
  // layer run as one full wave plus a 12 %-full second one and took 1.78x the time of a balanced grid
  // (profiles/r02_c5_bf16_launches).  Cost of a choice = ceil(workgroups / slots) x K-steps per slice x step time + the
  // fp32 atomics of the extra slices (~1.3 TB/s chip-wide, MI355X_MICROARCH.md) + a per-workgroup prologue/epilogue;
  // step times calibrated on MI355X (a 256x256 / 8-wave workgroup does the work of four 128x128 ones per K-step).
  static const char* force = getenv("SDM_WGRAD_BF16_FORCE");   // "bt,splitk" tuning override
  int fbt = 0, fsk = 0;
  if (force && sscanf(force, "%d,%d", &fbt, &fsk) != 2) fbt = fsk = 0;
  static const char* t256s = getenv("SDM_WGRAD_BF16_T256");    // calibration override of the 256x256 step time [us]
  const double t256 = t256s ? atof(t256s) * 1e-6 : 2.3e-6;
  int sk = 1, bt = big ? 128 : 64;
  {
    const double out_bytes = (double)d->M * d->N * d->ntaps * d->batch * 4.0;
    double best = 1e30;
    for (int cand = 0; cand < 2; ++cand) {
      int cbt = cand == 0 ? (big ? 128 : 64) : 256;
      if (cand == 1 && !(big && d->M >= 192 && d->N >= 192)) break;
      if (fbt == 64 || fbt == 128 || fbt == 256) {
        if (cand == 1) break;
        cbt = fbt;
      }
      const double slots = cbt == 256 ? 256.0 : (cbt == 128 ? 512.0 : 768.0);
      const double t_step = cbt == 256 ? t256 : (cbt == 128 ? 1.45e-6 : 0.70e-6);   // one K-step of a full wave of workgroups
      const long blocks = (long)cdiv(d->M, cbt) * cdiv(d->N, cbt) * d->ntaps * d->batch;
      for (int c = 1; c <= 128 && c <= (T >= 4 ? T / 4 : 1); ++c) {
        const double waves = ceil((double)blocks * c / slots);
        const double steps = ceil((double)T / c);
        const double t = waves * (steps * t_step + 2.0e-6) + (c > 1 || d->accumulate == 1 ? out_bytes * c / 1.3e12 : out_bytes / 5e12);
        if (t < best * 0.97) { best = t; sk = c; bt = cbt; }     // prefer fewer slices / the earlier tile unless it pays >= 3 %
      }
    }
  }
  // three taps per workgroup (wgrad3_bf16_kernel): plain 3x3 stride-1 geometry on maps at least 64 pixels wide
  bool use3 = false, use8 = false;
  Wgrad3Launch L3;
  {
    static const char* w3 = getenv("SDM_WGRAD3");                // "0": off; "1": on wherever eligible (tests); default: cost model
    static const char* t3s = getenv("SDM_WGRAD3_T");             // calibration override of its step time [us]
    bool ok = !(w3 && w3[0] == '0') && d->ntaps == 9 && d->batch == 1 && d->g_stride == 1 && d->x_stride == 1 &&
              (d->Wk % BKP == 0 || ((d->Wk == 32 || d->Wk == 16) && ((long)d->Hk * d->Wk) % BKP == 0)) &&
              d->g.H == d->Hk && d->g.W == d->Wk && d->x.H == d->Hk && d->x.W == d->Wk &&
              (fbt == 0 || fbt == 3 || fbt == 8);
    int seen[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int t = 0; ok && t < 9; ++t) {
      const int ty = d->x_dy[t], tx = d->x_dx[t];
      if (d->g_dy[t] != 0 || d->g_dx[t] != 0 || ty < -1 || ty > 1 || tx < -1 || tx > 1 || seen[ty + 1][tx + 1]) ok = false;
      else { seen[ty + 1][tx + 1] = 1; L3.tapw[ty + 1][tx + 1] = d->tap_w[t]; }
    }
    if (ok) {
      const double t3 = t3s ? atof(t3s) * 1e-6 : 2.5e-6;          // one K-step of a full wave of 512 workgroups (3 taps each)
      const long blocks3 = (long)cdiv(d->M, 128) * cdiv(d->N, 128) * 3;
      const double out_bytes = (double)d->M * d->N * 9 * 4.0;
      double best3 = 1e30, best_other = 1e30;
      int sk3 = 1;
      for (int c = 1; c <= 128 && c <= (T >= 4 ? T / 4 : 1); ++c) {
        const double t = ceil((double)blocks3 * c / 512.0) * (ceil((double)T / c) * t3 + 2.0e-6) +
                         (c > 1 || d->accumulate == 1 ? out_bytes * c / 1.3e12 : out_bytes / 5e12);
        if (t < best3 * 0.97) { best3 = t; sk3 = c; }
      }
      {   // the cost of the choice made above (same formula)
        const double slots = bt == 256 ? 256.0 : (bt == 128 ? 512.0 : 768.0);
        const double t_step = bt == 256 ? t256 : (bt == 128 ? 1.45e-6 : 0.70e-6);
        const long blocks = (long)cdiv(d->M, bt) * cdiv(d->N, bt) * d->ntaps;
        best_other = ceil((double)blocks * sk / slots) * (ceil((double)T / sk) * t_step + 2.0e-6) +
                     (sk > 1 || d->accumulate == 1 ? out_bytes * sk / 1.3e12 : out_bytes / 5e12);
      }
      // the 3- / 6-channel edge convs (one side of the tile almost empty) on full-resolution maps: measured 223 -> 162 us at
      // 524288 pixels, which the per-tap model (calibrated on full tiles) does not see
      const bool edge = (d->M < 64 || d->N < 64) && T >= 1024;
      // the phase-interleaved form of the same tile (sdm_wgrad8.hip): one 8-wave workgroup per CU
      static const char* w8 = getenv("SDM_WGRAD8");               // "0": off; "1": wherever eligible (tests); default: cost model
      static const char* t8s = getenv("SDM_WGRAD8_T");            // calibration override of its step time [us]
      double best8 = 1e30;
      int sk8 = 1;
      if ((!(w8 && w8[0] == '0') && !(w3 && w3[0] == '1' && !w8) && fbt == 0) || fbt == 8) {
        const double t8 = t8s ? atof(t8s) * 1e-6 : 1.55e-6;       // one K-step of a full wave of 256 workgroups (3 taps each)
        for (int c = 1; c <= 256 && c <= (T >= 4 ? T / 4 : 1); ++c) {
          const double t = ceil((double)blocks3 * c / 256.0) * (ceil((double)T / c) * t8 + 3.0e-6) +
                           (c > 1 || d->accumulate == 1 ? out_bytes * c / 1.3e12 : out_bytes / 5e12);
          if (t < best8 * 0.97) { best8 = t; sk8 = c; }
        }
      }
      if ((w8 && w8[0] == '1' && best8 < 1e30) || fbt == 8 || (best8 < best3 && (best8 < best_other || edge))) {
        use3 = use8 = true;
        sk = (fbt == 8 && fsk >= 1 && fsk <= T) ? fsk : sk8;
        for (int i = 0; i < 3; ++i) L3.dy[i] = i - 1;
      } else if ((w3 && w3[0] == '1') || fbt == 3 || best3 < best_other || edge) {
        use3 = true;
        sk = (fbt == 3 && fsk >= 1 && fsk <= T) ? fsk : sk3;
        for (int i = 0; i < 3; ++i) L3.dy[i] = i - 1;
      }
    }
  }
  if (!use3 && fsk >= 1 && fsk <= T) sk = fsk;
  const int ntm = cdiv(d->M, bt);
  L.ntn = cdiv(d->N, bt);
  L.splitk = sk;
  L.atomic_out = (sk > 1 || d->accumulate == 1) ? 1 : 0;
  if (sk > 1 && d->accumulate == 0) {
    for (int b = 0; b < d->batch; ++b)
      for (int t = 0; t < d->ntaps; ++t) {
        float* p = d->dw + (b / d->bdiv) * d->w_bs0 + (b % d->bdiv) * d->w_bs1 + (long long)d->tap_w[t] * d->w_tap_stride;
        SDM_LAUNCH(zero_rows_f32_kernel, dim3(cdiv((long)d->M * d->N, 1024)), dim3(256), 0, st, p, (long long)d->M, d->N,
                           (long long)d->ldw);
        SDM_CHECK_LAUNCH("zero_rows");
      }
  }
  if (use3) {
    L3.d = *d;
    L3.splitk = sk;
    L3.atomic_out = L.atomic_out;
    L3.ntn = cdiv(d->N, 128);
    if (use8) return sdm_launch_wgrad8x3_bf16(L3, st);
    constexpr size_t lds3 = (size_t)2 * (BKP * 256 + W3_XROWS * 256);
    static bool attr3 = false;
    if (!attr3) {
      (void)hipFuncSetAttribute((const void*)wgrad3_bf16_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      (void)hipFuncSetAttribute((const void*)wgrad3_bf16_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      (void)hipFuncSetAttribute((const void*)wgrad3_bf16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      attr3 = true;
    }
    {
      const double px = (double)d->images;
      const double g_b = px * d->g.H * d->g.W * d->M, x_b = px * d->x.H * d->x.W * d->N, w_b = (double)d->M * d->N * 9;
      sdm_prof_shape(d->M, d->N, (int)P, 9, sk, 2.0 * (g_b + x_b) + 4.0 * w_b);
    }
    sdm_prof_pre(st);
    const dim3 grid3(cdiv(d->M, 128) * L3.ntn * sk * 3, 1, 1);
    if (d->Wk >= 64) SDM_LAUNCH(wgrad3_bf16_kernel<6>, grid3, dim3(256), lds3, st, L3);
    else if (d->Wk == 32) SDM_LAUNCH(wgrad3_bf16_kernel<5>, grid3, dim3(256), lds3, st, L3);
    else SDM_LAUNCH(wgrad3_bf16_kernel<4>, grid3, dim3(256), lds3, st, L3);
    sdm_prof_post(st, 3, 2.0 * d->M * d->N * (double)P * 9);
    SDM_CHECK_LAUNCH("wgrad3_bf16");
    return SDM_OK;
  }
  // structured pixel walk when a 64-pixel K-step is an aligned piece of one image row or a whole number of rows
  static const char* nofast = getenv("SDM_WGRAD_BF16_FASTP");
  bool fastp = !(nofast && nofast[0] == '0');
  if (d->Wk % BKP == 0) L.rows_per_step = 1;
  else if (BKP % d->Wk == 0 && d->Hk % (BKP / d->Wk) == 0) L.rows_per_step = BKP / d->Wk;
  else { L.rows_per_step = 0; fastp = false; }
  if (bt == 256) return fastp ? launch_wgrad_bf16<256, 2, 4, 2, true>(L, ntm, st) : launch_wgrad_bf16<256, 2, 4, 2, false>(L, ntm, st);
  if (bt == 128) return fastp ? launch_wgrad_bf16<128, 2, 2, 2, true>(L, ntm, st) : launch_wgrad_bf16<128, 2, 2, 2, false>(L, ntm, st);
  return fastp ? launch_wgrad_bf16<64, 2, 2, 3, true>(L, ntm, st) : launch_wgrad_bf16<64, 2, 2, 3, false>(L, ntm, st);
}

// --------------------------------------------------------------------------------------------
// dst[t][c][r] = bf16(src[t][r][c]): the transposed bf16 weight copy the data gradient contracts against
// (tap-major conv weights [tap][c_out][c_in] -> [tap][c_in][c_out]; linear weights [O][K] -> [K][O] with T = 1)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_bf16_t_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int R, int Cc,
                                                          int ldd) {
  __shared__ float tile[64][65];
  const float* s = src + (long long)blockIdx.z * R * Cc;
  bf16_t* o = dst + (long long)blockIdx.z * Cc * ldd;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? s[(long long)r * Cc + c] : 0.f;
  }
  __syncthreads();
  if ((ldd & 7) == 0 && ((((unsigned long long)dst) & 15) == 0)) {
    // 8 consecutive r per lane: one 16-byte store instead of eight 2-byte ones (rows of the copy start 16-byte aligned)
    for (int e = threadIdx.x; e < 64 * 8; e += 256) {
      const int i = e >> 3, r8 = (e & 7) * 8;
      const int c = c0 + i, r = r0 + r8;
      if (c >= Cc || r >= ldd) continue;
      u16x8 v;
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = f2bf(tile[r8 + k][i]);                // r in [R, ldd): zero padding (tile holds zeros)
      *(u16x8*)(o + (long long)c * ldd + r) = v;
    }
    return;
  }
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cc && r < ldd) o[(long long)c * ldd + r] = f2bf(tile[tx][i]);      // r in [R, ldd): zero padding
  }
}

extern "C" int sdm_cast_bf16_t(const float* src, uint16_t* dst, int T, int R, int Cc, int ldd, sdm_stream_t s) {
  SDM_CHECK_ARG(src && dst && T > 0 && R > 0 && Cc > 0 && ldd >= R, "cast_bf16_t: bad args");
  SDM_LAUNCH(cast_bf16_t_kernel, dim3(cdiv(Cc, 64), cdiv(ldd, 64), T), dim3(256), 0, (hipStream_t)s, src, dst, R, Cc, ldd);
  SDM_CHECK_LAUNCH("cast_bf16_t");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// AdaGN backward on bf16 rows (fp32 sums / coefficients; same ws / coef contract as sdm_gn_adagn_bwd).
// --------------------------------------------------------------------------------------------
// 8 channels per lane (C % 8 == 0, C <= 2048); partials meet in LDS, one atomic pair per channel per workgroup
__global__ __launch_bounds__(256) void gn_bwd_sums_bf16_kernel(const bf16_t* __restrict__ dout, long long lddo,
                                                               const bf16_t* __restrict__ y, long long ldy, int HW, int C,
                                                               int rows_per_block, float* __restrict__ ws) {
  __shared__ float part[2][256][8];
  const int n = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > HW) r1 = HW;
  const int q = C >> 3;                  // lanes per row
  const int rl = threadIdx.x / q, cl = threadIdx.x - rl * q;
  const int RL = 256 / q;                // rows in flight
  float s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
  if (rl < RL) {
    for (int r = r0 + rl; r < r1; r += RL) {
      const long long row = (long long)n * HW + r;
      const u16x8 dv = *(const u16x8*)(dout + row * lddo + cl * 8);
      const u16x8 yv = *(const u16x8*)(y + row * ldy + cl * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float dd = bf2f(dv[k]);
        s1[k] += dd;
        s2[k] = fmaf(dd, bf2f(yv[k]), s2[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    part[0][threadIdx.x][k] = s1[k];
    part[1][threadIdx.x][k] = s2[k];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c >> 3, k = c & 7;
    float a = 0.f, b = 0.f;
    for (int j = 0; j < RL; ++j) {
      a += part[0][j * q + g][k];
      b += part[1][j * q + g][k];
    }
    atomicAdd(&ws[((long long)n * C + c) * 2 + 0], a);
    atomicAdd(&ws[((long long)n * C + c) * 2 + 1], b);
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_bf16_kernel(const bf16_t* __restrict__ dout, long long lddo,
                                                                const bf16_t* __restrict__ y, long long ldy,
                                                                const bf16_t* __restrict__ z, long long ldz,
                                                                const float* __restrict__ coef, bf16_t* __restrict__ dz,
                                                                long long lddz, long long rows, int HW, int C) {
  const int C8 = C >> 3;
  const long long total = rows * C8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = sdm_idiv(e, C8);
    const int c = (int)(e - r * C8) * 8;
    const int n = (int)sdm_idiv(r, HW);
    const float* k = coef + (long long)n * 3 * C + c;
    const u16x8 dv = *(const u16x8*)(dout + r * lddo + c);
    const u16x8 yv = *(const u16x8*)(y + r * ldy + c);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = k[i] * bf2f(dv[i]) + k[C + i] * bf2f(yv[i]) + k[2 * C + i];
    if (z) {
      const u16x8 zv = *(const u16x8*)(z + r * ldz + c);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= sdm_dsilu(bf2f(zv[i]));
    }
    u16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
    *(u16x8*)(dz + r * lddz + c) = o;
  }
}

// coefficient pass: dtype-independent (works on the fp32 sums); kernel and launcher live in sdm_pointwise.hip
int sdm_launch_gn_bwd_coef(float* ws, int images, int HW, int C, int groups, const float* gamma, const float* beta,
                           const float* mod, int mod_ns, const float* mean, const float* rstd, float* coef, float* dmod,
                           float* dgamma, float* dbeta, hipStream_t st);

extern "C" int sdm_gn_adagn_bwd_bf16(const uint16_t* dout, int64_t lddo, const uint16_t* y, int64_t ldy, const uint16_t* z,
                                     int64_t ldz, int images, int HW, int C, int groups, const float* gamma,
                                     const float* beta, const float* mod, int mod_nstride, const float* mean,
                                     const float* rstd, uint16_t* dz, int64_t lddz, float* dmod, float* dgamma, float* dbeta,
                                     float* ws, float* coef, sdm_stream_t s) {
  SDM_CHECK_ARG(dout && y && gamma && beta && mean && rstd && dz && ws && coef, "gn_adagn_bwd_bf16: null pointer");
  SDM_CHECK_ARG(images > 0 && HW > 0 && groups > 0 && C % groups == 0, "gn_adagn_bwd_bf16: bad dims");
  SDM_CHECK_ARG(C % 8 == 0 && C <= 2048 && lddo % 8 == 0 && ldy % 8 == 0 && lddz % 8 == 0 && (z == nullptr || ldz % 8 == 0) &&
                    aligned16(dout) && aligned16(y) && aligned16(dz) && (z == nullptr || aligned16(z)) && aligned16(coef),
                "gn_adagn_bwd_bf16: rows must be 16-byte aligned, C a multiple of 8 (<= 2048)");
  hipStream_t st = (hipStream_t)s;
  int rpb = (HW + 63) / 64;
  if (rpb < 8) rpb = 8;
  SDM_LAUNCH(gn_bwd_sums_bf16_kernel, dim3(cdiv(HW, rpb), images), dim3(256), 0, st, dout, (long long)lddo, y,
                     (long long)ldy, HW, C, rpb, ws);
  SDM_CHECK_LAUNCH("gn_bwd_sums_bf16");
  int rc = sdm_launch_gn_bwd_coef(ws, images, HW, C, groups, gamma, beta, mod, mod_nstride, mean, rstd, coef, dmod, dgamma,
                                  dbeta, st);
  if (rc != SDM_OK) return rc;
  const long long rows = (long long)images * HW;
  const long long work = rows * (C / 8);
  int grid = (int)((work + 255) / 256);
  if (grid > 8192) grid = 8192;
  SDM_LAUNCH(gn_bwd_apply_bf16_kernel, dim3(grid), dim3(256), 0, st, dout, (long long)lddo, y, (long long)ldy, z,
                     (long long)ldz, coef, dz, (long long)lddz, rows, HW, C);
  SDM_CHECK_LAUNCH("gn_bwd_apply_bf16");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// column-softmax backward: dS[i][j] = scale * P[i][j] * (dP[i][j] - sum_i P[i][j] dP[i][j]);  P bf16, dP fp32 -> dS bf16;
// dS columns [S, ldds) are zero-filled (reduction padding of the contractions that follow).
// --------------------------------------------------------------------------------------------
#define CSB_W 16   // row phases (waves) per workgroup, lane = column (see col_softmax_bf16_kernel)
__global__ __launch_bounds__(64 * CSB_W) void col_softmax_bwd_bf16_kernel(const bf16_t* __restrict__ p, long long ldp,
                                                                          const float* __restrict__ dp, long long lddp,
                                                                          bf16_t* __restrict__ ds, long long ldds, int S,
                                                                          float scale) {
  __shared__ float sd[CSB_W][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const bf16_t* pb = p + (long long)blockIdx.y * S * ldp;
  const float* dpb = dp + (long long)blockIdx.y * S * lddp;
  bf16_t* dsb = ds + (long long)blockIdx.y * S * ldds;
  float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
  if (j < S) {
    int i = wave;
    for (; i + 3 * CSB_W < S; i += 4 * CSB_W) {       // four independent rows per trip
      d0 = fmaf(bf2f(pb[(long long)i * ldp + j]), dpb[(long long)i * lddp + j], d0);
      d1 = fmaf(bf2f(pb[(long long)(i + CSB_W) * ldp + j]), dpb[(long long)(i + CSB_W) * lddp + j], d1);
      d2 = fmaf(bf2f(pb[(long long)(i + 2 * CSB_W) * ldp + j]), dpb[(long long)(i + 2 * CSB_W) * lddp + j], d2);
      d3 = fmaf(bf2f(pb[(long long)(i + 3 * CSB_W) * ldp + j]), dpb[(long long)(i + 3 * CSB_W) * lddp + j], d3);
    }
    for (; i < S; i += CSB_W) d0 = fmaf(bf2f(pb[(long long)i * ldp + j]), dpb[(long long)i * lddp + j], d0);
  }
  sd[wave][lane] = (d0 + d1) + (d2 + d3);
  __syncthreads();
  float dot = 0.f;
#pragma unroll
  for (int w = 0; w < CSB_W; ++w) dot += sd[w][lane];
  if (j < S) {
#pragma unroll 4
    for (int i = wave; i < S; i += CSB_W)
      dsb[(long long)i * ldds + j] = f2bf(scale * bf2f(pb[(long long)i * ldp + j]) * (dpb[(long long)i * lddp + j] - dot));
  } else if (j < ldds) {
    for (int i = wave; i < S; i += CSB_W) dsb[(long long)i * ldds + j] = 0;
  }
}

extern "C" int sdm_col_softmax_bwd_bf16(const uint16_t* P, int64_t ldp, const float* dP, int64_t lddp, uint16_t* dS,
                                        int64_t ldds, int batch, int S, float scale, sdm_stream_t s) {
  SDM_CHECK_ARG(P && dP && dS && batch > 0 && S > 0 && ldp >= S && lddp >= S && ldds >= S, "col_softmax_bwd_bf16: bad args");
  SDM_LAUNCH(col_softmax_bwd_bf16_kernel, dim3(cdiv(ldds, 64), batch), dim3(64 * CSB_W), 0, (hipStream_t)s, P, (long long)ldp,
                     dP, (long long)lddp, dS, (long long)ldds, S, scale);
  SDM_CHECK_LAUNCH("col_softmax_bwd_bf16");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// elementwise / reductions on bf16 rows
// --------------------------------------------------------------------------------------------
__global__ void silu_bwd_bf16_kernel(const bf16_t* dy, const bf16_t* z, bf16_t* dx, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    dx[i] = f2bf(bf2f(dy[i]) * sdm_dsilu(bf2f(z[i])));
}
extern "C" int sdm_silu_bwd_bf16(const uint16_t* dy, const uint16_t* z, uint16_t* dx, size_t n, sdm_stream_t s) {
  SDM_CHECK_ARG(dy && z && dx && n > 0, "silu_bwd_bf16: bad args");
  size_t grid = (n + 255) / 256;
  if (grid > 8192) grid = 8192;
  SDM_LAUNCH(silu_bwd_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)s, dy, z, dx, n);
  SDM_CHECK_LAUNCH("silu_bwd_bf16");
  return SDM_OK;
}

// 8 channels (one 16-byte load) per lane: a workgroup's 256 lanes cover RL = 256 / (C8 chunks) rows per pass when the
// row is short (C <= 2048), fold their partial sums through LDS, and leave one fp32 atomic per column and workgroup.
// (The first cut read 2 bytes per lane and row with C of 256 lanes active: 0.5 TB/s on the 134 MB gradient of the
// 256x256 up-sampling conv, 0.7 ms of a 28 ms step.)
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, long long ldx, long long rows, int C,
                                                          int rows_per_block, float* __restrict__ out) {
  __shared__ float red[256][9];
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  const int C8 = (C + 7) >> 3;
  if (C8 <= 256 && (C & 7) == 0 && (ldx & 7) == 0 && (((unsigned long long)x) & 15) == 0) {
    int cl = 1;
    while (cl < C8) cl <<= 1;                // chunk lanes: C8 rounded up to a power of two (<= 256)
    const int RL = 256 / cl;                 // row lanes
    const int ch = threadIdx.x % cl, rl = threadIdx.x / cl;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ch < C8) {
      long long r = r0 + rl;
      for (; r + 7 * RL < r1; r += 8 * RL) {        // eight independent rows per trip: their loads overlap
        u16x8 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const u16x8*)(x + (r + j * RL) * ldx + 8 * ch);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          a[k] += ((bf2f(v[0][k]) + bf2f(v[1][k])) + (bf2f(v[2][k]) + bf2f(v[3][k]))) +
                  ((bf2f(v[4][k]) + bf2f(v[5][k])) + (bf2f(v[6][k]) + bf2f(v[7][k])));
      }
      for (; r < r1; r += RL) {
        const u16x8 v = *(const u16x8*)(x + r * ldx + 8 * ch);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += bf2f(v[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = a[k];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const int chunk = c >> 3, k = c & 7;
      float t = 0.f;
      for (int q = 0; q < RL; ++q) t += red[q * cl + chunk][k];
      atomicAdd(&out[c], t);
    }
    return;
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (long long r = r0; r < r1; ++r) a += bf2f(x[r * ldx + c]);
    atomicAdd(&out[c], a);
  }
}
extern "C" int sdm_colsum_bf16(const uint16_t* x, int64_t ldx, int rows, int C, float* out, sdm_stream_t s) {
  SDM_CHECK_ARG(x && out && rows > 0 && C > 0, "colsum_bf16: bad args");
  int rpb = (rows + 1023) / 1024;      // <= 1024 workgroups: every one ends in C atomics on the same C words
  if (rpb < 64) rpb = 64;
  SDM_LAUNCH(colsum_bf16_kernel, dim3(cdiv(rows, rpb)), dim3(256), 0, (hipStream_t)s, x, (long long)ldx, (long long)rows,
                     C, rpb, out);
  SDM_CHECK_LAUNCH("colsum_bf16");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// NCHW fp32 image / gradient -> NHWC bf16 rows padded to Cp channels (Cp a multiple of 8, pad = 0): puts the 3- or
// 6-channel side of the edge convolutions (in_layer.0 input, out_layers.1 output gradient) into the row format of the
// matrix-core kernels, so that their weight / data gradients run on sdm_wgrad_bf16 / sdm_igemm_bf16 like any other conv
// (the one-wave-per-pixel VALU forms took 9.4 ms + 1.4 ms + 1.0 ms per step at 256x256, batch 8).
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_rows_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int C,
                                                                int Cp, long long HW, long long P) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
    const long long n = sdm_idiv(p, HW), rem = p - n * HW;
    for (int c0 = 0; c0 < Cp; c0 += 8) {
      u16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (c0 + k < C) ? f2bf(x[(n * C + c0 + k) * HW + rem]) : (bf16_t)0;
      *(u16x8*)(out + p * Cp + c0) = o;
    }
  }
}

extern "C" int sdm_nchw_to_rows_bf16(const float* x_nchw, uint16_t* out_rows, int images, int C, int H, int W, int Cp,
                                     sdm_stream_t s) {
  SDM_CHECK_ARG(x_nchw && out_rows && images > 0 && C > 0 && H > 0 && W > 0 && Cp >= C && Cp % 8 == 0 && aligned16(out_rows),
                "nchw_to_rows_bf16: bad args (Cp must be a multiple of 8 and >= C)");
  const long long HW = (long long)H * W, P = HW * images;
  int grid = (int)((P + 255) / 256);
  if (grid > 8192) grid = 8192;
  SDM_LAUNCH(nchw_to_rows_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, x_nchw, out_rows, C, Cp, HW, P);
  SDM_CHECK_LAUNCH("nchw_to_rows_bf16");
  return SDM_OK;
}
