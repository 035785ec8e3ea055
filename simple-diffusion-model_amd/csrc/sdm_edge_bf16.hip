// Edge convolutions of the bf16 path on the matrix cores: the stem (in_layer.0: nn.Conv2d(in_channel, min_channel, 3, 1, 1)
// + SiLU on the fp32 NCHW image -> bf16 pixel rows) and the head (out_layers.1: nn.Conv2d(min_channel, out_channel, 3, 1, 1)
// (+ Tanh) on bf16 pixel rows -> the fp32 NCHW prediction) of the reference U-Net (models/U_Net.py:55-62,113-130 there).
//
// Why they exist (profiles/r04_c3_fwd_launches.csv, r04_c3_fwd_bf16_kernel_stats.csv): at 128 images of 128x128 the head,
// run as a 64-wide tile of sdm_igemm_bf16 with 3 live columns, took 720 us (+ 15 us for the NCHW pass) and the stem, one
// output channel per thread with the 27-value patch broadcast from LDS, 469 us: together 4.7 % of the forward, both 4-5x
// above what their HBM bytes cost.  Both are one side narrow (3 channels), so neither is a tile of the big contraction:
//
//   head  one wave owns a strip of 16 input pixels x `seg` rows of one image and walks DOWN the rows.  Every input row
//         is loaded ONCE straight into the matrix-core operand layout (lane = pixel, 16-byte k-chunk), its dx = -1 / +1
//         neighbours come from the 16-lane row shifts of the data-parallel-primitive moves (lanes 1..14 of the strip
//         are outputs: the strip pitch is 14), and the row feeds the accumulators of the THREE output rows it touches
//         (36 x v_mfma_f32_16x16x32_bf16 per row at 128 channels, weights as the row operand, zero-padded to 16 and held
//         in registers for the life of the wave).  Reads: 16/14 x (seg + 2)/seg of the activation, no LDS, the fp32 NCHW
//         result with bias (+ Tanh) is written from the accumulators.
//   stem  K = 9 * in_channel (27) is too short for the bf16 instructions and the image is fp32: v_mfma_f32_16x16x4_f32
//         with the fp32 weights as the row operand (held in registers, rows permuted so that a lane ends up with 8
//         CONSECUTIVE output channels of its pixel), 16 pixels per step gathered from the NCHW planes, bias + SiLU,
//         one 16-byte store per lane and 32 channels.  fp32 arithmetic like the kernel it replaces (the image and the
//         weights are NOT rounded to bf16).
#include <stdlib.h>
#include <type_traits>

#define SDM_FAST_SIGMOID
#include "sdm_common.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));

// row_shr:1 (lane i takes lane i - 1 of its 16-lane row) / row_shl:1 (lane i takes lane i + 1); lanes without a source get 0
__device__ __forceinline__ i32x4 dpp_from_left(i32x4 v) {
  i32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = __builtin_amdgcn_update_dpp(0, v[e], 0x111, 0xf, 0xf, true);
  return r;
}
__device__ __forceinline__ i32x4 dpp_from_right(i32x4 v) {
  i32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = __builtin_amdgcn_update_dpp(0, v[e], 0x101, 0xf, 0xf, true);
  return r;
}
__device__ __forceinline__ f32x4 mfma_bf(i32x4 a, i32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int N, typename F, int I = 0>
__device__ __forceinline__ void cfor_edge(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    cfor_edge<N, F, I + 1>(static_cast<F&&>(f));
  }
}

#define HEAD_PITCH 14      // output pixels of a 16-lane strip

// x: bf16 rows [images*H*W][ldx] (first KS*32 channels used); w16: bf16 [9][Cout][KS*32], tap = ky*3 + kx; out: fp32 NCHW.
template <int KS>
__global__ __launch_bounds__(256, 2) void head_conv_mfma_bf16_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                                     const bf16_t* __restrict__ w16,
                                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                                     int H, int W, int Cout, int act, int seg, int strips,
                                                                     int segs, int tasks) {
  constexpr int CIN = KS * 32;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, lq = lane >> 4;

  // weights as the row operand: row = output channel (zero beyond Cout), lane (row l15, k-chunk lq)
  i32x4 wf[9][KS];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      i32x4 v = {0, 0, 0, 0};
      if (l15 < Cout) v = *(const i32x4*)(w16 + ((long long)t * Cout + l15) * CIN + s * 32 + lq * 8);
      wf[t][s] = v;
    }
  for (int task = blockIdx.x * 4 + wave; task < tasks; task += gridDim.x * 4) {
    const int strip = task % strips, t2 = task / strips, sg = t2 % segs, n = t2 / segs;
    const int y0 = sg * seg, y1 = (y0 + seg < H) ? y0 + seg : H;
    const int px = strip * HEAD_PITCH - 1 + l15;                  // input pixel of this lane; lanes 1..14 are outputs
    const bool pin = (unsigned)px < (unsigned)W;
    const bool pout = l15 >= 1 && l15 <= HEAD_PITCH && px < W;
    const bf16_t* xp = x + ((long long)n * H * W + px) * ldx + lq * 8;
    float* op = out + (long long)n * Cout * H * W + px;

    auto load_row = [&](i32x4 (&b)[KS], int yin) {
      const bool ok = pin && (unsigned)yin < (unsigned)H;
      const bf16_t* r = xp + (long long)yin * W * ldx;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        i32x4 v = {0, 0, 0, 0};
        if (ok) v = *(const i32x4*)(r + s * 32);
        b[s] = v;
      }
    };
    auto store_row = [&](const f32x4& a, int yo) {
      if (pout) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (lq * 4 + e < Cout) op[((long long)(lq * 4 + e) * H + yo) * W] = sdm_act(a[e] + (bias ? bias[lq * 4 + e] : 0.f), act);
      }
    };

    // input row yin feeds output rows yin - 1 (kernel row 2), yin (kernel row 1), yin + 1 (kernel row 0).  Three row
    // buffers (the loads of rows yin + 1 and yin + 2 are in flight while row yin is consumed) and three accumulators
    // rotate with the same period: the walk is unrolled by three, no register moves.
    f32x4 acc[3];
    i32x4 buf[3][KS];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto step = [&](auto ic, int yin) {
      constexpr int I = decltype(ic)::value, TOP = I, MID = (I + 1) % 3, BOT = (I + 2) % 3;
      if (yin + 2 <= y1) load_row(buf[(I + 2) % 3], yin + 2);
      const bool do_top = yin - 1 >= y0, do_mid = yin >= y0 && yin < y1, do_bot = yin + 1 < y1;
      if ((unsigned)yin < (unsigned)H) {
        if (do_top && do_mid && do_bot) {
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const i32x4 c = buf[I][s], l = dpp_from_left(c), r = dpp_from_right(c);
            acc[TOP] = mfma_bf(wf[6][s], l, acc[TOP]);
            acc[MID] = mfma_bf(wf[3][s], l, acc[MID]);
            acc[BOT] = mfma_bf(wf[0][s], l, acc[BOT]);
            acc[TOP] = mfma_bf(wf[7][s], c, acc[TOP]);
            acc[MID] = mfma_bf(wf[4][s], c, acc[MID]);
            acc[BOT] = mfma_bf(wf[1][s], c, acc[BOT]);
            acc[TOP] = mfma_bf(wf[8][s], r, acc[TOP]);
            acc[MID] = mfma_bf(wf[5][s], r, acc[MID]);
            acc[BOT] = mfma_bf(wf[2][s], r, acc[BOT]);
          }
        } else {
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const i32x4 c = buf[I][s], l = dpp_from_left(c), r = dpp_from_right(c);
            if (do_top) {
              acc[TOP] = mfma_bf(wf[6][s], l, acc[TOP]);
              acc[TOP] = mfma_bf(wf[7][s], c, acc[TOP]);
              acc[TOP] = mfma_bf(wf[8][s], r, acc[TOP]);
            }
            if (do_mid) {
              acc[MID] = mfma_bf(wf[3][s], l, acc[MID]);
              acc[MID] = mfma_bf(wf[4][s], c, acc[MID]);
              acc[MID] = mfma_bf(wf[5][s], r, acc[MID]);
            }
            if (do_bot) {
              acc[BOT] = mfma_bf(wf[0][s], l, acc[BOT]);
              acc[BOT] = mfma_bf(wf[1][s], c, acc[BOT]);
              acc[BOT] = mfma_bf(wf[2][s], r, acc[BOT]);
            }
          }
        }
      }
      if (do_top) store_row(acc[TOP], yin - 1);
      acc[TOP] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    int yin = y0 - 1;
    load_row(buf[0], yin);
    load_row(buf[1], yin + 1);                     // y0 <= y1: always a row of the walk
    while (true) {
      step(std::integral_constant<int, 0>(), yin);
      if (++yin > y1) break;
      step(std::integral_constant<int, 1>(), yin);
      if (++yin > y1) break;
      step(std::integral_constant<int, 2>(), yin);
      if (++yin > y1) break;
    }
  }
}

// Cout <= 4 (RGB / RGBA): the three kernel ROWS share one matrix instruction.  Matrix row 4 q + c is (kernel row 2 - q, output
// channel c): an input row's fragment multiplied by that 12-row operand adds to the accumulators of output rows yin - 1
// (q = 0), yin (q = 1) and yin + 1 (q = 2) at once, which the 16x16 result layout puts in lane groups q = lane >> 4.  After
// the step group 0 is finished (stored), and the accumulator moves one lane group down (ds_bpermute: lane l takes lane
// l + 16, groups 2 and 3 restart at zero).  12 instead of 36 matrix instructions per input row and 48 instead of 144 weight
// registers: three waves per SIMD with NBUF - 1 row loads of 4 KB each in flight -- the kernel is a latency-bound stream,
// and bytes in flight are what it was short of (256 -> see profiles/r04_edge_probe.txt).
template <int KS, int NBUF, int WPS>
__global__ __launch_bounds__(256, WPS) void head_conv_mfma4_bf16_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                                      const bf16_t* __restrict__ w16,
                                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                                      int H, int W, int Cout, int act, int seg, int strips,
                                                                      int segs, int tasks) {
  constexpr int CIN = KS * 32;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, lq = lane >> 4;
  i32x4 wf[3][KS];                                   // [kx][k-step]: row l15 = 4 q + c -> tap (2 - q) * 3 + kx, channel c
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      i32x4 v = {0, 0, 0, 0};
      const int q = l15 >> 2, c = l15 & 3;
      if (q < 3 && c < Cout) v = *(const i32x4*)(w16 + ((long long)((2 - q) * 3 + kx) * Cout + c) * CIN + s * 32 + lq * 8);
      wf[kx][s] = v;
    }
  const int down = (lane + 16) * 4;                   // ds_bpermute address: the lane one group up

  for (int task = blockIdx.x * 4 + wave; task < tasks; task += gridDim.x * 4) {
    const int strip = task % strips, t2 = task / strips, sg = t2 % segs, n = t2 / segs;
    const int y0 = sg * seg, y1 = (y0 + seg < H) ? y0 + seg : H;
    const int px = strip * HEAD_PITCH - 1 + l15;
    const bool pin = (unsigned)px < (unsigned)W;
    const bool pout = lq == 0 && l15 >= 1 && l15 <= HEAD_PITCH && px < W;
    const bf16_t* xp = x + ((long long)n * H * W + px) * ldx + lq * 8;
    float* op = out + (long long)n * Cout * H * W + px;

    auto load_row = [&](i32x4 (&b)[KS], int yin) {
      const bool ok = pin && (unsigned)yin < (unsigned)H;
      const bf16_t* r = xp + (long long)yin * W * ldx;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        i32x4 v = {0, 0, 0, 0};
        if (ok) v = *(const i32x4*)(r + s * 32);
        b[s] = v;
      }
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    i32x4 buf[NBUF][KS];
    auto step = [&](auto ic, int yin) {
      constexpr int I = decltype(ic)::value;
      if (yin + NBUF - 1 <= y1) load_row(buf[(I + NBUF - 1) % NBUF], yin + NBUF - 1);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const i32x4 c = buf[I][s], l = dpp_from_left(c), r = dpp_from_right(c);
        acc = mfma_bf(wf[0][s], l, acc);
        acc = mfma_bf(wf[1][s], c, acc);
        acc = mfma_bf(wf[2][s], r, acc);
      }
      if (pout && yin - 1 >= y0) {                    // lane group 0: output row yin - 1 is complete
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < Cout) op[((long long)e * H + (yin - 1)) * W] = sdm_act(acc[e] + (bias ? bias[e] : 0.f), act);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = acc[e];
        asm volatile("" : "+v"(t));                    // (a bit_cast of the vector ELEMENT is miscompiled: it read element 0 each time)
        const int v = __builtin_amdgcn_ds_bpermute(down, __float_as_int(t));
        acc[e] = lane < 32 ? __int_as_float(v) : 0.f;
      }
    };
    int yin = y0 - 1;
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) load_row(buf[i], yin + i);   // rows past y1 are loaded (or zero) but never consumed
    bool more = true;
    while (more) {
      cfor_edge<NBUF>([&](auto ic) {
        if (more) {
          step(ic, yin);
          more = ++yin <= y1;
        }
      });
    }
  }
}

extern "C" int sdm_head_conv_mfma_bf16(const uint16_t* x, long long ldx, const uint16_t* w16, const float* bias, float* out,
                                       int images, int Cin, int H, int W, int Cout, int act, sdm_stream_t s) {
  SDM_CHECK_ARG(x && w16 && out && images > 0 && H > 0 && W > 0, "head_conv_mfma_bf16: bad args");
  SDM_CHECK_ARG(Cout >= 1 && Cout <= 16, "head_conv_mfma_bf16: Cout %d not in [1,16]", Cout);
  if (Cin != 64 && Cin != 128) return sdm_set_error(SDM_ERR_UNSUPPORTED, "head_conv_mfma_bf16: Cin %d (64 or 128)", Cin);
  SDM_CHECK_ARG(aligned16(x) && aligned16(w16) && ldx >= Cin && ldx % 8 == 0, "head_conv_mfma_bf16: 16-byte aligned rows");
  const int strips = cdiv(W, HEAD_PITCH);
  int seg = 32;                                   // rows per wave: shorter while the grid would leave SIMDs without a wave
  while (seg > 8 && (long long)images * cdiv(H, seg) * strips < 12288) seg >>= 1;
  const int segs = cdiv(H, seg);
  const long long tasks = (long long)images * segs * strips;
  SDM_CHECK_ARG(tasks < (1ll << 31) && (long long)images * H * W < (1ll << 31), "head_conv_mfma_bf16: too many pixels");
  int grid = cdiv(tasks, 4);
  if (grid > 512) grid = 512;                     // two workgroups per CU; a wave keeps its weights over its tasks
  const bf16_t* xb = (const bf16_t*)x;
  const bf16_t* wb = (const bf16_t*)w16;
  static const char* pk = getenv("SDM_HEAD_PACKED");                // tuning aid: 0 = the three-accumulator kernel for every Cout
  if (Cout <= 4 && !(pk && pk[0] == '0')) {
    static const char* vs = getenv("SDM_HEAD_VARIANT");            // tuning aid: "34" = 3 waves/SIMD x 4 row buffers (default), "43", "44"
    const int variant = vs ? atoi(vs) : 34;
    const int wps = variant / 10 == 4 ? 4 : 3;
    grid = cdiv(tasks, 4);
    if (grid > wps * 256) grid = wps * 256;
#define HEAD4(KS_, NB_, WPS_) SDM_LAUNCH((head_conv_mfma4_bf16_kernel<KS_, NB_, WPS_>), dim3(grid), dim3(256), 0, (hipStream_t)s, xb, ldx, wb, bias, out, H, W, Cout, act, seg, strips, segs, (int)tasks)
    if (Cin == 128) {
      if (variant == 43) HEAD4(4, 3, 4);
      else if (variant == 44) HEAD4(4, 4, 4);
      else HEAD4(4, 4, 3);
    } else {
      HEAD4(2, 4, 4);
    }
#undef HEAD4
    SDM_CHECK_LAUNCH("head_conv_mfma4_bf16");
    return SDM_OK;
  }
  if (Cin == 128)
    SDM_LAUNCH(head_conv_mfma_bf16_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)s, xb, ldx, wb, bias, out, H, W, Cout, act,
               seg, strips, segs, (int)tasks);
  else
    SDM_LAUNCH(head_conv_mfma_bf16_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)s, xb, ldx, wb, bias, out, H, W, Cout, act,
               seg, strips, segs, (int)tasks);
  SDM_CHECK_LAUNCH("head_conv_mfma_bf16");
  return SDM_OK;
}

// --------------------------------------------------------------------------------------------
// stem.  x: fp32 NCHW [images][CIN][H][W]; w: fp32 [9][Cout][CIN]; out / preact: bf16 rows [images*H*W][Cout].
// A workgroup column (blockIdx.y) owns 16 NB output channels = NB blocks of 16 matrix rows (NB = 4 or 2: 28 / 14 weight
// registers at 3 input channels, 3-4 waves per SIMD; with 8 blocks or a run-time block count the kernel had 2 waves or
// spilled).  Matrix row i of block b = 2g + h is channel 32g + 8 (i >> 2) + 4h + (i & 3), so that the accumulators of blocks
// 2g, 2g + 1 of a lane (pixel l15, rows 4 lq .. 4 lq + 3) are the 8 consecutive channels 32g + 8 lq .. + 7 of that pixel.
// The activation is a template parameter and the gathers are branch-free (an out-of-image entry reads a valid dummy address
// and is zeroed): straight-line code, so that the matrix instructions of one wave and the SiLU arithmetic of the others on
// its SIMD overlap.
// ACT == STEM_DGRAD turns the same kernel into the DATA GRADIENT of the head conv (the transposed conv 3 -> 128 channels):
// x = the fp32 NCHW gradient of the prediction, w = the head's taps [9][CIN][Cout] read transposed and flipped (tap 8 - t),
// no bias, and the result is multiplied by silu'(z) of the producer's pre-activation z (bf16 rows behind `preact`, an INPUT
// here; null: no factor): dx[p][c] = silu'(z[p][c]) * sum_tap sum_co dy[n][co][p - d(tap)] * w[tap][co][c].
// --------------------------------------------------------------------------------------------
#define STEM_DGRAD 3
template <int CIN, int NB, int ACT>
__global__ __launch_bounds__(256, (CIN <= 4 ? 3 : 2)) void stem_conv_mfma_bf16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                                     bf16_t* __restrict__ preact, int H, int W, int Cout,
                                                                     long long P, int groups) {
  constexpr int K = CIN * 9, KST = (K + 3) / 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, lq = lane >> 4;
  const int c0 = blockIdx.y * (16 * NB);

  float wa[NB][KST];
  int koff[KST], kdy[KST], kdx[KST];                               // the lane's patch entries k = 4 s + lq -> (plane, dy, dx)
#pragma unroll
  for (int s = 0; s < KST; ++s) {
    const int k = 4 * s + lq;
    const int ci = k / 9, tap = k - ci * 9;
    const bool live = k < K;
    kdx[s] = tap % 3 - 1;
    koff[s] = ci * H * W + (tap / 3 - 1) * W + kdx[s];
    kdy[s] = live ? tap / 3 - 1 : (1 << 20);                       // dead entries fail the bounds test
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int co = c0 + 32 * (b >> 1) + 8 * (l15 >> 2) + 4 * (b & 1) + (l15 & 3);
      if constexpr (ACT == STEM_DGRAD) wa[b][s] = live ? w[((long long)(8 - tap) * CIN + ci) * Cout + co] : 0.f;
      else wa[b][s] = live ? w[((long long)tap * Cout + co) * CIN + ci] : 0.f;
    }
  }
  f32x4 bv[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[b][e] = (ACT != STEM_DGRAD && bias) ? bias[c0 + 32 * (b >> 1) + 8 * lq + 4 * (b & 1) + e] : 0.f;

  const int HW = H * W;
  auto gather = [&](float (&pt)[KST], int grp) {                 // the lane's KST patch entries of pixel grp * 16 + l15
    const unsigned p = (unsigned)grp * 16u + (unsigned)l15;          // P < 2^31 (checked by the launcher)
    const bool pv = p < (unsigned)P;
    const int n = (int)(p / (unsigned)HW), rem = (int)(p - (unsigned)n * (unsigned)HW);
    const int y = (int)((unsigned)rem / (unsigned)W), xx = rem - y * W;
    const float* xb = x + (long long)n * CIN * HW + rem;
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      const bool ok = pv && (unsigned)(y + kdy[s]) < (unsigned)H && (unsigned)(xx + kdx[s]) < (unsigned)W;
      const float* q = ok ? xb + koff[s] : x;                      // always a valid address: the load is unconditional
      const float v = *q;
      pt[s] = ok ? v : 0.f;
    }
  };
  float pt[KST], pn[KST];
  const int gstep = gridDim.x * 4;
  int grp = blockIdx.x * 4 + wave;
  if (grp < groups) gather(pt, grp);
  for (; grp < groups; grp += gstep) {
    const long long p = (long long)grp * 16 + l15;
    const bool pv = p < P;
    const int nxt = grp + gstep < groups ? grp + gstep : grp;      // (the last iteration re-reads its own patch: no branch)
    gather(pn, nxt);                                               // in flight under this group's matrix work
    f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < KST; ++s)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[b][s], pt[s], acc[b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (pv) {
#pragma unroll
      for (int g = 0; g < NB / 2; ++g) {
        const long long o = (long long)p * Cout + c0 + 32 * g + 8 * lq;
        u16x8 zo, ao;
        if constexpr (ACT == STEM_DGRAD) {
          u16x8 zi = {0, 0, 0, 0, 0, 0, 0, 0};
          if (preact) zi = *(const u16x8*)(preact + o);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a = acc[2 * g + h][e];
              ao[4 * h + e] = f2bf(preact ? a * sdm_dsilu(bf2f(zi[4 * h + e])) : a);
            }
          *(u16x8*)(out + o) = ao;
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a = acc[2 * g + h][e] + bv[2 * g + h][e];
              zo[4 * h + e] = f2bf(a);
              ao[4 * h + e] = f2bf(ACT == SDM_ACT_SILU ? sdm_silu(a) : a);
            }
          if (preact) *(u16x8*)(preact + o) = zo;
          *(u16x8*)(out + o) = ao;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < KST; ++s) pt[s] = pn[s];
  }
}

// Called by sdm_stem_conv_bf16 (sdm_bf16.hip; act = SDM_ACT_NONE / SDM_ACT_SILU) and by sdm_head_conv_dgrad_bf16
// (sdm_pointwise.hip; act = STEM_DGRAD: x = dy, Cin = the head's output channels, Cout = its input channels, preact = z or
// null) when the shape qualifies; returns SDM_ERR_UNSUPPORTED otherwise (the caller falls back to its scalar kernel).
int sdm_launch_stem_mfma_bf16(const float* x, const float* w, const float* bias, bf16_t* out, bf16_t* preact, int images, int Cin,
                              int H, int W, int Cout, int act, hipStream_t st) {
  const long long P = (long long)images * H * W;
  if (Cout % 32 != 0 || !aligned16(out) || (preact && !aligned16(preact)) || P >= (1ll << 31) - 64 ||
      (long long)Cin * H * W >= (1ll << 31) || (act != SDM_ACT_SILU && act != SDM_ACT_NONE && act != STEM_DGRAD))
    return SDM_ERR_UNSUPPORTED;
  const int groups = (int)((P + 15) / 16);
  int gx = cdiv(groups, 4);
  if (gx > 1024) gx = 1024;
  const int nb = Cout % 64 == 0 ? 4 : 2;
  const dim3 grid(gx, Cout / (16 * nb));
#define STEM_MF(C, B, A) SDM_LAUNCH((stem_conv_mfma_bf16_kernel<C, B, A>), grid, dim3(256), 0, st, x, w, bias, out, preact, H, W, Cout, P, groups)
#define STEM_MF_C(C)                                                      \
  if (nb == 4 && act == SDM_ACT_SILU) STEM_MF(C, 4, SDM_ACT_SILU);        \
  else if (nb == 4 && act == STEM_DGRAD) STEM_MF(C, 4, STEM_DGRAD);       \
  else if (nb == 4) STEM_MF(C, 4, SDM_ACT_NONE);                          \
  else if (act == SDM_ACT_SILU) STEM_MF(C, 2, SDM_ACT_SILU);              \
  else if (act == STEM_DGRAD) STEM_MF(C, 2, STEM_DGRAD);                  \
  else STEM_MF(C, 2, SDM_ACT_NONE)
  switch (Cin) {
    case 1: STEM_MF_C(1); break;
    case 2: STEM_MF_C(2); break;
    case 3: STEM_MF_C(3); break;
    case 4: STEM_MF_C(4); break;
    case 5: STEM_MF_C(5); break;
    case 6: STEM_MF_C(6); break;
    case 7: STEM_MF_C(7); break;
    case 8: STEM_MF_C(8); break;
    default: return SDM_ERR_UNSUPPORTED;
  }
#undef STEM_MF_C
#undef STEM_MF
  return SDM_OK;
}
