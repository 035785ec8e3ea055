// Internal to the bf16 contraction translation units (sdm_bf16.hip, sdm_igemm8.hip): the launch record the host planner
// fills for a kernel.
#pragma once
#include "sdm_common.h"

struct BfLaunch {
  sdm_igemm_desc d;
  int splitk, ntn, kchunks, mfast, out_f32;
  int vec_epi;   // 4-column vector epilogue allowed (N % 4 == 0, 8-byte aligned output / residual rows, 16-byte bias)
  int gn_stats;  // accumulate per-(image, channel) sum / sum of squares of the stored outputs into d.gn_sums
  long long out_rows;   // output rows of ONE batch entry (split-K scratch is [batch][out_rows][N])
  int pow2, lw, lhw;    // prologue shortcuts, as in sdm_igemm.hip: Wt == 1 << lw, Ht * Wt == 1 << lhw
  int plain_out;        // output row == tile-grid row
  int pp;               // 8-wave tiles: ping-pong schedule of the two 4-wave halves (see the kernel)
  int no_simple_epi;    // igemm8: SDM_IGEMM8_SIMPLE_EPI=0 keeps the general epilogue loop for every launch (A/B)
};

// sdm_igemm8.hip: the phase-interleaved 256x256 tile (FAST addressing, vector epilogue, no split-K)
int sdm_launch_igemm8_bf16(const BfLaunch& L, int ntm, hipStream_t st);

// Launch record of the three-taps-per-workgroup weight-gradient kernels (wgrad3_bf16_kernel in sdm_bf16_train.hip, the
// phase-interleaved wgrad8x3_bf16_kernel in sdm_wgrad8.hip)
struct Wgrad3Launch {
  sdm_wgrad_desc d;
  int splitk, ntn, atomic_out;
  int dy[3];           // kernel-row offsets handled by blockIdx.y = 0..2
  int tapw[3][3];      // weight tap index of (dy index, dx = -1, 0, +1)
};
int sdm_launch_wgrad8x3_bf16(const Wgrad3Launch& L3, hipStream_t st);

// sdm_edge_bf16.hip: the stem conv on v_mfma_f32_16x16x4_f32 (SDM_ERR_UNSUPPORTED when the shape does not qualify: the caller
// falls back to stem_conv_bf16_kernel)
int sdm_launch_stem_mfma_bf16(const float* x, const float* w, const float* bias, bf16_t* out, bf16_t* preact, int images, int Cin,
                              int H, int W, int Cout, int act, hipStream_t st);
