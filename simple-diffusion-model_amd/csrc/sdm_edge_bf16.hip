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

#define HEAD_PITCH 14      // output pixels of a 16-lane strip

// x: bf16 rows [images*H*W][ldx] (first KS*32 channels used); w16: bf16 [9][Cout][KS*32], tap = ky*3 + kx; out: fp32 NCHW.
template <int KS>
__global__ __launch_bounds__(256, 2) void head_conv_mfma_bf16_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                                     const bf16_t* __restrict__ w16,
                                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                                     int H, int W, int Cout, int act, int seg, int strips,
                                                                     int segs, int tasks) {
  constexpr int CIN = KS * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;

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
  float bs[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bs[e] = (bias && lq * 4 + e < Cout) ? bias[lq * 4 + e] : 0.f;

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
          if (lq * 4 + e < Cout) op[((long long)(lq * 4 + e) * H + yo) * W] = sdm_act(a[e] + bs[e], act);
      }
    };

    // input row yin feeds output rows yin - 1 (kernel row 2), yin (kernel row 1), yin + 1 (kernel row 0)
    f32x4 a_top = {0.f, 0.f, 0.f, 0.f}, a_mid = a_top, a_bot = a_top;
    i32x4 cur[KS], nxt[KS];
    load_row(cur, y0 - 1);
    for (int yin = y0 - 1; yin <= y1; ++yin) {
      if (yin < y1) load_row(nxt, yin + 1);
      const bool do_top = yin - 1 >= y0, do_mid = yin >= y0 && yin < y1, do_bot = yin + 1 < y1;
      if ((unsigned)yin < (unsigned)H) {
        if (do_top && do_mid && do_bot) {
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const i32x4 c = cur[s], l = dpp_from_left(c), r = dpp_from_right(c);
            a_top = mfma_bf(wf[6][s], l, a_top);
            a_mid = mfma_bf(wf[3][s], l, a_mid);
            a_bot = mfma_bf(wf[0][s], l, a_bot);
            a_top = mfma_bf(wf[7][s], c, a_top);
            a_mid = mfma_bf(wf[4][s], c, a_mid);
            a_bot = mfma_bf(wf[1][s], c, a_bot);
            a_top = mfma_bf(wf[8][s], r, a_top);
            a_mid = mfma_bf(wf[5][s], r, a_mid);
            a_bot = mfma_bf(wf[2][s], r, a_bot);
          }
        } else {
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const i32x4 c = cur[s], l = dpp_from_left(c), r = dpp_from_right(c);
            if (do_top) {
              a_top = mfma_bf(wf[6][s], l, a_top);
              a_top = mfma_bf(wf[7][s], c, a_top);
              a_top = mfma_bf(wf[8][s], r, a_top);
            }
            if (do_mid) {
              a_mid = mfma_bf(wf[3][s], l, a_mid);
              a_mid = mfma_bf(wf[4][s], c, a_mid);
              a_mid = mfma_bf(wf[5][s], r, a_mid);
            }
            if (do_bot) {
              a_bot = mfma_bf(wf[0][s], l, a_bot);
              a_bot = mfma_bf(wf[1][s], c, a_bot);
              a_bot = mfma_bf(wf[2][s], r, a_bot);
            }
          }
        }
      }
      if (do_top) store_row(a_top, yin - 1);
      a_top = a_mid;
      a_mid = a_bot;
      a_bot = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
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
  while (seg > 8 && (long long)images * cdiv(H, seg) * strips < 8192) seg >>= 1;
  const int segs = cdiv(H, seg);
  const long long tasks = (long long)images * segs * strips;
  SDM_CHECK_ARG(tasks < (1ll << 31) && (long long)images * H * W < (1ll << 31), "head_conv_mfma_bf16: too many pixels");
  int grid = cdiv(tasks, 4);
  if (grid > 512) grid = 512;                     // two workgroups per CU; a wave keeps its weights over its tasks
  const bf16_t* xb = (const bf16_t*)x;
  const bf16_t* wb = (const bf16_t*)w16;
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
// A workgroup column (blockIdx.y) owns 16 NB output channels = NB blocks of 16 matrix rows (NB = 8, 4 or 2: the largest whose
// 16 NB divides Cout; a run-time block count made the compiler spill ~700 registers); matrix row i of block b = 2g + h
// is channel 32g + 8 (i >> 2) + 4h + (i & 3), so that the accumulators of blocks 2g, 2g + 1 of a lane (pixel l15, rows
// 4 lq .. 4 lq + 3) are the 8 consecutive channels 32g + 8 lq .. + 7 of that pixel.
// --------------------------------------------------------------------------------------------
template <int CIN, int NB>
__global__ __launch_bounds__(256, 2) void stem_conv_mfma_bf16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                                     bf16_t* __restrict__ preact, int H, int W, int Cout, int act,
                                                                     long long P, int groups) {
  constexpr int K = CIN * 9, KST = (K + 3) / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
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
      wa[b][s] = live ? w[((long long)tap * Cout + co) * CIN + ci] : 0.f;
    }
  }
  f32x4 bv[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[b][e] = bias ? bias[c0 + 32 * (b >> 1) + 8 * lq + 4 * (b & 1) + e] : 0.f;

  const int HW = H * W;
  for (int grp = blockIdx.x * 4 + wave; grp < groups; grp += gridDim.x * 4) {
    const long long p = (long long)grp * 16 + l15;
    const bool pv = p < P;
    const int n = (int)sdm_idiv(p, HW), rem = (int)(p - (long long)n * HW);
    const int y = rem / W, xx = rem - y * W;
    const float* xb = x + (long long)n * CIN * HW + rem;
    float pt[KST];
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      const bool ok = pv && (unsigned)(y + kdy[s]) < (unsigned)H && (unsigned)(xx + kdx[s]) < (unsigned)W;
      pt[s] = ok ? xb[koff[s]] : 0.f;
    }
    f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KST; ++s)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[b][s], pt[s], acc[b], 0, 0, 0);
    if (pv) {
#pragma unroll
      for (int g = 0; g < NB / 2; ++g) {
        u16x8 zo, ao;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = acc[2 * g + h][e] + bv[2 * g + h][e];
            zo[4 * h + e] = f2bf(a);
            ao[4 * h + e] = f2bf(sdm_act(a, act));
          }
        const long long o = p * Cout + c0 + 32 * g + 8 * lq;
        if (preact) *(u16x8*)(preact + o) = zo;
        *(u16x8*)(out + o) = ao;
      }
    }
  }
}

// Called by sdm_stem_conv_bf16 (sdm_bf16.hip) when the shape qualifies; returns SDM_ERR_UNSUPPORTED otherwise.
int sdm_launch_stem_mfma_bf16(const float* x, const float* w, const float* bias, bf16_t* out, bf16_t* preact, int images, int Cin,
                              int H, int W, int Cout, int act, hipStream_t st) {
  const long long P = (long long)images * H * W;
  if (Cout % 32 != 0 || !aligned16(out) || (preact && !aligned16(preact)) || P >= (1ll << 31) - 64 ||
      (long long)Cin * H * W >= (1ll << 31))
    return SDM_ERR_UNSUPPORTED;
  const int groups = (int)((P + 15) / 16);
  int gx = cdiv(groups, 4);
  if (gx > 512) gx = 512;
  const int nb = Cout % 128 == 0 ? 8 : (Cout % 64 == 0 ? 4 : 2);
  const dim3 grid(gx, Cout / (16 * nb));
#define STEM_MF(C, B) SDM_LAUNCH((stem_conv_mfma_bf16_kernel<C, B>), grid, dim3(256), 0, st, x, w, bias, out, preact, H, W, Cout, act, P, groups)
#define STEM_MF_C(C)            \
  if (nb == 8) STEM_MF(C, 8);   \
  else if (nb == 4) STEM_MF(C, 4); \
  else STEM_MF(C, 2)
  switch (Cin) {
    case 1: STEM_MF_C(1); break;
    case 2: STEM_MF_C(2); break;
    case 3: STEM_MF_C(3); break;
    case 4: STEM_MF_C(4); break;
    default: return SDM_ERR_UNSUPPORTED;
  }
#undef STEM_MF_C
#undef STEM_MF
  return SDM_OK;
}
