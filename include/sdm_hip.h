/*
 * sdm_hip.h — C ABI of libsdm_hip.so, the MI355X (gfx950) kernel library behind the
 * U-Net denoise hot path of Simple-Diffusion-Model.
 *
 * The reference has no FFI layer (it is pure PyTorch), so every entry
 * point below replaces one *class of torch op* that the reference's
 * models/custom_layers.py dispatches; the citation on each function is the reference
 * call site it stands in for.  The Python host (simple-diffusion-model_amd/sdm_hip/)
 * binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (tensor.data_ptr()); fp32 unless stated
 *   - activations are NHWC ("pixel rows x channels") inside the library; NCHW only at
 *     sdm_stem_conv (input) and sdm_head_conv (output)
 *   - conv weights are tap-major:  w[ky][kx][c_out][c_in]   (Conv2d and ConvTranspose2d
 *     alike; the PyTorch-shaped parameter is a strided *view* of that storage)
 *   - launch functions never allocate, synchronise or copy host<->device: they are
 *     safe inside hipGraph stream capture.  Scratch comes from the caller (ws).
 *   - return value: 0 = ok, <0 = error (sdm_last_error() gives the text, thread-local)
 */
#ifndef SDM_HIP_H
#define SDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sdm_stream_t; /* hipStream_t */

#define SDM_OK 0
#define SDM_ERR_ARG (-1)
#define SDM_ERR_LAUNCH (-2)
#define SDM_ERR_UNSUPPORTED (-3)
#define SDM_ERR_COMM (-5)        /* RCCL (or its absence) reported a failure; sdm_last_error() carries RCCL's own text */
#define SDM_ERR_DEVICE (-4)      /* a kernel reported a failure (a bounded device-side wait gave up); see sdm_device_error_poll */

#define SDM_ACT_NONE 0
#define SDM_ACT_SILU 1
#define SDM_ACT_TANH 2

#define SDM_MAX_PHASE 4
#define SDM_MAX_TAPS 16

int sdm_version(void);
const char* sdm_last_error(void);
/* out[0]=CU count, out[1]=LDS bytes per CU-workgroup limit, out[2]=wave size, out[3]=1 if arch is gfx950 */
int sdm_device_info(int* out4);
/* sizeof the ABI structs (0 igemm desc, 1 wgrad desc, 2 rows, 3 linear job, 4 adam job) so a binding can assert its layout */
size_t sdm_struct_size(int which);
/* Stream ordering without host work: everything enqueued on `from` so far happens-before what is enqueued on `to`
 * after this call (hipEventRecord on `from` + hipStreamWaitEvent on `to`, events from a library-owned ring; legal
 * under stream capture, where it makes `to` join the capture).  The weight-gradient side stream forks and joins
 * the main stream with it ~70 times per training step. */
int sdm_stream_order(sdm_stream_t from, sdm_stream_t to);
/* Number of GPU kernels the library has launched in this process so far (every launch function counts its own kernels,
 * including finishing passes; memsets and the caller's own torch kernels are not included).  bench.py reports the
 * difference over one step as launches_per_step. */
unsigned long long sdm_launch_count(void);
/* Device-side failure word.  The cooperative kernels (sdm_gn_apply_coop, sdm_gn_bwd_coop) bound every inter-workgroup
 * wait; a wait that gives up writes NaN outputs, leaves the exchange state zero and stores a code in a host-mapped word.
 * sdm_device_error_init allocates that word (idempotent; hipHostMalloc: call it once, outside stream capture, before the
 * first launch -- without it the kernels are still bounded and still write NaN, but nothing reports).  From then on EVERY
 * launch function of the library checks the word after enqueueing its kernels (a read of pinned host memory, no
 * synchronisation) and returns SDM_ERR_DEVICE with the failing kernel named in sdm_last_error(); sdm_device_error_poll
 * does the same check on its own (after a stream synchronisation it is exact).  The word is cleared by the report. */
int sdm_device_error_init(void);
int sdm_device_error_poll(void);

/* Launch plans: record the library's kernel launches once, replay them from C (SURVEY 8b "who calls the C ABI": the
 * samplers' captured step; VERDICT r2 "C-side launch plan").  Between sdm_plan_begin and sdm_plan_end every launch the
 * library issues -- eagerly, or into a stream capture -- is also appended to the plan with a copy of its argument VALUES
 * (device pointers included: the caller keeps those buffers alive and in place, e.g. by holding the hipGraph captured over
 * the same calls, whose memory pool owns them).  sdm_plan_end returns the handle (>= 1), or 0 when something ran that a
 * plan cannot replay (a memset inside the library): the recording is dropped.  sdm_plan_run re-issues the sequence
 * `repeat` times on `stream` with plain launches (no graph): a step whose only per-step state lives in device memory (the
 * samplers' cursor and tables) is replayed with ONE call.  Single-threaded, one recording at a time. */
int sdm_plan_begin(void);
int sdm_plan_end(void);
int sdm_plan_launches(int handle);
int sdm_plan_run(int handle, int repeat, sdm_stream_t stream);
int sdm_plan_free(int handle);

/* Per-kernel-class timing with HIP events on the launch stream (bench.py's roofline leg; not capturable, so
 * keep it off inside hipGraph capture).  Classes: 0 igemm 128x128, 1 igemm 64x64, 2 igemm 32x128,
 * 3 wgrad 128x128, 4 wgrad 64x64, 5 igemm 128x64, 6 skinny-M weight streaming.  sdm_prof_begin(cap) arms it for up to cap launches; sdm_prof_end
 * synchronises the recorded events and fills, per class c: ms[c], launches[c], flops[c] (algorithmic 2*M*N*K*taps), bytes[c] (algorithmic: every
 * input row, weight and output element once, 4 B each). */
#define SDM_PROF_CLASSES 7
int sdm_prof_begin(int capacity);
int sdm_prof_end(double* ms, long long* launches, double* flops, double* bytes);
/* Median elapsed time (ms) of an event pair around an EMPTY kernel, `reps` launches on the stream (which is synchronised): what
 * the per-launch timing above adds to a kernel's own duration, plus the empty kernel's own ~1-2 us.  bench.py reports it and a
 * kernel-only estimate next to the raw event figures (rocprofv3's kernel durations are the reference for that estimate). */
double sdm_prof_bracket_ms(int reps, sdm_stream_t s);

/* A logical matrix of "pixel rows x channels", optionally the channel-concat of two
 * tensors (U_Net.py:168 torch.cat) and optionally carrying a pending per-(image,channel)
 * affine v*scale+shift (the deferred GroupNorm+AdaGN of custom_layers.py:35-45) that is
 * applied when the rows are loaded; out-of-image taps stay exactly zero. */
typedef struct sdm_rows {
  const float* p0;     /* channels [0, csplit) */
  const float* p1;     /* channels [csplit, C); NULL when not a concat */
  int csplit;          /* = C when p1 == NULL */
  int64_t ld0, ld1;    /* row pitch in floats */
  const float* scale;  /* [n][C] or [1][C]; NULL = no affine */
  const float* shift;
  int aff_nstride;     /* C for per-image tables, 0 for a broadcast (batch-1 embedding) table */
  int H, W;            /* spatial dims of the rows: row = (n*H + y)*W + x ; plain matrix: H=rows, W=1 */
} sdm_rows;

/* Implicit-GEMM contraction on the fp32 matrix cores (v_mfma_f32_32x32x2_f32):
 *
 *   out[opix(m)][j] = epi( alpha * sum_{tap} sum_{c<K}  A[ipix(m,tap)][c] * Wt[tap][j][c] )
 *
 * m runs over the tile grid (n, y, x) of (Ht, Wt) per image; ipix = (n, y*sstride+dy, x*sstride+dx)
 * inside a's (H, W); opix = (n, y*ostride+oy, x*ostride+ox) inside (Ho, Wo).
 * Serves: Conv2d k3 s1/s2 (custom_layers.py:222-227,195-200), ConvTranspose2d k4 s2 p1 as 4
 * sub-pixel phases (:173-178), nn.Linear on [N,S,C] tokens (:138,:156), the attention
 * einsums (:144,:150), and every data-gradient of those (b_kmajor selects W used as [k][j]).
 *
 * epi(v): v += bias[j]; z=v (stored to out_preact if given); v = act(v);
 *         v += res[opix][j] + res2[opix][j]; v *= silu'(dact_z[opix][j]); store.
 */
typedef struct sdm_igemm_desc {
  int M;      /* images * Ht * Wt (rows of the tile grid, per batch entry) */
  int N;      /* output channels */
  int K;      /* reduction channels per tap */
  int batch;  /* independent GEMMs (attention: images*heads); 1 otherwise */
  int bdiv;   /* batch b -> (b / bdiv, b % bdiv) for the two-level strides below (>=1) */

  sdm_rows a;
  int64_t a_bs0, a_bs1; /* batch strides of a.p0 (floats) */
  int Ht, Wt;           /* tile grid per image (M = images*Ht*Wt) */
  int sstride;          /* source = tile*sstride + d */
  int nphase;           /* 1, or 4 for sub-pixel phases */
  int ntaps[SDM_MAX_PHASE];
  int8_t tap_dy[SDM_MAX_PHASE][SDM_MAX_TAPS];
  int8_t tap_dx[SDM_MAX_PHASE][SDM_MAX_TAPS];
  uint8_t tap_w[SDM_MAX_PHASE][SDM_MAX_TAPS]; /* weight slice index of the tap */

  const float* b;        /* weights */
  int64_t ldb;           /* pitch of one weight row */
  int64_t b_tap_stride;  /* floats between tap slices */
  int b_kmajor;          /* 0: Wt[tap][j][c] (c contiguous)   1: Wt[tap][c][j] (j contiguous) */
  int64_t b_bs0, b_bs1;

  int ostride, Ho, Wo;   /* output geometry */
  int ph_oy[SDM_MAX_PHASE], ph_ox[SDM_MAX_PHASE];
  float* out;
  int64_t ldo;
  int64_t o_bs0, o_bs1;

  float alpha;
  const float* bias;      /* [N] or NULL */
  int act;                /* SDM_ACT_* */
  float* out_preact;      /* NULL or same geometry as out (pitch ldo): pre-activation copy for backward */
  const float* res;       /* NULL or addend with pitch ldres at opix */
  int64_t ldres;
  int64_t r_bs0, r_bs1;
  const float* res2;      /* NULL or second addend (pitch ldres2) */
  int64_t ldres2;
  const float* dact_z;    /* NULL or pre-activation of the tensor whose gradient this is (pitch ldz) */
  int64_t ldz;
  int dact_ncols;         /* silu'(z) is applied to output columns [0, dact_ncols); 0 = all N */

  float* ws;              /* split-K scratch: batch*M_out_rows*N floats (sdm_igemm_workspace_bytes), used only when the
                             library splits K.
                             CONTRACT: the caller hands it over zero-filled once; every call leaves the part it
                             touched zero-filled again (the finishing pass clears what it reads) */
  size_t ws_bytes;
  float* gn_sums;         /* sdm_igemm_bf16 only; NULL or ZERO-FILLED fp32 [images][N][2]: the epilogue adds, per (image, output
                             channel), the sum and the sum of squares of the values it stores (after activation and bf16
                             rounding) -- the GroupNorm statistics of the conv's output without another pass over it
                             (SURVEY §7 step 5).  Done only for an unsplit plain 3x3 stride-1 conv whose tiles lie inside
                             one image; the call then returns SDM_STATS_DONE and sdm_gn_finalize turns the sums into the
                             affine tables; otherwise the sums stay untouched (zero) and the caller runs sdm_gn_affine_bf16. */
  /* ---- sdm_igemm_bf16 only: the two-pass column softmax of attention (custom_layers.py:144-150: softmax over the QUERY
   * axis, i.e. down the columns of scores = q k^T) without the S x S score matrix in HBM.  Same GEMM twice:
   *   pass 1, col_stats != NULL: nothing is stored to `out`; for every M-tile t of the grid and column j the epilogue
   *     writes the partial statistics of v = alpha * acc over the tile's rows,  col_stats[b][t][j] = {max_i v, sum_i
   *     exp(v - max)}  (fp32 pairs, batch stride col_stats_bs floats; sdm_attn_colstats_combine folds the tiles);
   *   pass 2, col_shift / col_mul != NULL: v = act(alpha * acc + col_shift[b][j]) * col_mul[b][j] with act =
   *     SDM_ACT_EXP, i.e. exp(s - m_j) / l_j written once, as bf16 probabilities.
   * Per-column vectors are fp32 [batch][N] with batch stride colvec_bs.  The product alpha * acc is ROUNDED to fp32 before
   * the shift in both passes (no fma contraction), so pass 2 sees exactly the values pass 1 took its maxima from. */
  float* col_stats;
  int64_t col_stats_bs;
  const float* col_shift;
  const float* col_mul;
  int64_t colvec_bs;
  int defer_epilogue;     /* 1: if the library splits K, it leaves the RAW sums in ws ([out row][N], alpha / bias / act not yet
                             applied) and returns SDM_DEFERRED instead of running its finishing pass: the caller's next
                             kernel finishes them (sdm_gn_apply_fused folds the finishing pass into the GroupNorm that
                             follows every U-Net conv block).  Only for batch == 1, no residual / dact_z (out_preact is
                             allowed and, like out, left unwritten when the call defers: sdm_gn_apply_fused_train).  When
                             the library does not split K the call behaves as usual and returns SDM_OK. */
} sdm_igemm_desc;

#define SDM_ACT_EXP 3       /* sdm_igemm_bf16 with col_shift only */
#define SDM_NOT_APPLICABLE 3  /* sdm_gn_apply_coop: shape outside its envelope, nothing launched */
#define SDM_DEFERRED 1
#define SDM_STATS_DONE 2
int sdm_igemm(const sdm_igemm_desc* d, sdm_stream_t stream);
/* scratch the library may want for this problem (0 if it will not split K) */
size_t sdm_igemm_workspace_bytes(const sdm_igemm_desc* d);

/* Weight-gradient contraction (autograd of Conv2d / ConvTranspose2d / Linear weight,
 * train_diffusion.py:358):  K runs over pixels.
 *
 *   dw[tap][i][j] (+)= sum_{p in (images,Hk,Wk)}  G[gpix(p,tap)][i] * X[xpix(p,tap)][j]
 */
typedef struct sdm_wgrad_desc {
  int M;          /* rows of dw per tap (output channels of the forward op) */
  int N;          /* cols of dw per tap (input channels of the forward op) */
  int images, Hk, Wk; /* reduction grid */
  int batch, bdiv;
  sdm_rows g;     /* gradient rows (channels = M) */
  int g_stride;   /* g source = p*g_stride + tap_gd */
  sdm_rows x;     /* activation rows (channels = N), may carry affine / concat */
  int x_stride;
  int64_t g_bs0, g_bs1, x_bs0, x_bs1;
  int ntaps;
  int8_t g_dy[SDM_MAX_TAPS], g_dx[SDM_MAX_TAPS];
  int8_t x_dy[SDM_MAX_TAPS], x_dx[SDM_MAX_TAPS];
  uint8_t tap_w[SDM_MAX_TAPS];
  float* dw;
  int64_t ldw, w_tap_stride;
  int64_t w_bs0, w_bs1;
  float* dbias;   /* optional [M]: += column sums of the gradient rows (bias gradient); needs plain g rows
                     (g_stride 1, zero tap offsets); NULL to skip */
  float alpha;
  int accumulate; /* 0: dw is overwritten (library zeroes it first if it splits K); 1: added into;
                     2: dw is known to be all zeros already (no memset; result = the contraction) */
} sdm_wgrad_desc;

int sdm_wgrad(const sdm_wgrad_desc* d, sdm_stream_t stream);

/* in_layer.0 (U_Net.py:55-60): Conv2d(Cin<=8 -> C, k3 p1) + SiLU, NCHW image in, NHWC rows out.
 * out_preact optional (training). */
int sdm_stem_conv(const float* x_nchw, const float* w_tap, const float* bias, float* out_nhwc,
                  float* out_preact, int images, int Cin, int H, int W, int Cout, int act, sdm_stream_t s);
/* out_layers.1 (+Tanh) (U_Net.py:121-128): Conv2d(C -> Cout<=8, k3 p1), NHWC rows in, NCHW out. */
int sdm_head_conv(const float* x_nhwc, const float* w_tap, const float* bias, float* out_nchw,
                  int images, int Cin, int H, int W, int Cout, int act, sdm_stream_t s);
/* data-gradient of the head conv: dy NCHW [images,Cout,H,W] -> dx NHWC [rows][Cin], times silu'(z). */
int sdm_head_conv_dgrad(const float* dy_nchw, const float* w_tap, float* dx_nhwc, const float* dact_z,
                        int images, int Cin, int H, int W, int Cout, sdm_stream_t s);
/* weight/bias gradient of stem (x NCHW, g NHWC) and head (x NHWC, g NCHW); dw tap-major.  Both ACCUMULATE into
 * dw / dbias (hand in zero-filled buffers) and need ws = sdm_edge_wgrad_workspace_bytes(small_channels, C) bytes:
 * per-workgroup partial slabs folded by a second pass (no atomics on the weight gradient). */
size_t sdm_edge_wgrad_workspace_bytes(int small_channels, int C);
int sdm_stem_conv_wgrad(const float* x_nchw, const float* g_nhwc, float* dw_tap, float* dbias, float* ws,
                        int images, int Cin, int H, int W, int Cout, sdm_stream_t s);
int sdm_head_conv_wgrad(const float* x_nhwc, const float* g_nchw, float* dw_tap, float* dbias, float* ws,
                        int images, int Cin, int H, int W, int Cout, sdm_stream_t s);
/* NCHW fp32 -> NHWC fp32 rows [images*H*W][Cp] (channels [C, Cp) zero): the image side of the edge convs as pixel rows, so
 * that their weight gradients run on sdm_wgrad (matrix cores) -- 3x / 1.5x faster than the two entry points above at 64x64
 * batch 4, which remain for callers that have no row buffer. */
int sdm_nchw_to_rows(const float* x_nchw, float* out_rows, int images, int C, int H, int W, int Cp, sdm_stream_t s);

/* GroupNorm(32) statistics of y [images][HW][C] folded with the AdaGN modulation into the
 * per-(image,channel) affine that consumers apply on load (custom_layers.py:35-45, quirk: shift==scale):
 *   scale[n][c] = rstd*gamma[c]*s ,  shift[n][c] = (beta[c] - mean*rstd*gamma[c])*s + s ,  s = mod[n or 0][c]
 * mean/rstd [images][groups] are also written (backward needs them). */
int sdm_gn_affine(const float* y, int64_t ldy, int images, int HW, int C, int groups, float eps,
                  const float* gamma, const float* beta, const float* mod, int mod_nstride,
                  float* scale, float* shift, float* mean, float* rstd, sdm_stream_t s);

/* UNet_ConvBlock tail in ONE pass per (image, group) (custom_layers.py:222-229 conv -> Swish -> AdaGN, :35-45, and the
 * ResidualBlock add :284-286), for no-grad forwards:
 *   y   = src                                   (src_raw == 0: the conv's finished output rows, pitch ld_src)
 *       = act(alpha * src + bias[c])            (src_raw == 1: split-K raw sums left by sdm_igemm(defer_epilogue), [rows][C];
 *                                                cleared to zero again, as the scratch contract demands)
 *   out = s * GroupNorm(y; gamma, beta) + s + cat(x0, x1)        s = mod[n or 0][c] (1 when mod == NULL; then no "+ s")
 * GroupNorm statistics with fp32 partials folded into fp64, like sdm_gn_affine.  Replaces three launches of the sampling
 * step (split-K finishing pass, sdm_gn_affine, sdm_affine_residual) with one; x0 NULL = no residual. */
int sdm_gn_apply_fused(const float* src, int64_t ld_src, int src_raw, float alpha, const float* bias, int act,
                       int images, int HW, int C, int groups, float eps, const float* gamma, const float* beta,
                       const float* mod, int mod_nstride, const float* x0, int64_t ldx0, const float* x1, int64_t ldx1,
                       int csplit, float* out, int64_t ldo, sdm_stream_t s);
/* sdm_gn_apply_fused for the layers where one workgroup per (image, group) is the wrong shape (a group = 16-32 bytes of
 * every 512-byte row at 64x64 / 32x32): a workgroup owns a SLAB of whole rows (coalesced, one sweep, values kept in
 * registers) and the slabs of an image exchange the per-group sums once through fp64 atomics and an arrival counter
 * (agent-scope atomics on both sides; placement-independent; at most 256 workgroups, all resident).  fp32 (bf16 = 0) or
 * bf16 rows / residual / output; src_raw as in sdm_gn_apply_fused(_bf16).  `state`: sdm_gn_apply_coop_state_bytes() of
 * device memory, ZERO on first use, owned by ONE stream at a time; every launch leaves it zero.  Returns
 * SDM_NOT_APPLICABLE (nothing launched) when the shape is outside the envelope -- fewer than two slabs, more than 256
 * workgroups, channel counts that are not powers of two: the caller then runs sdm_gn_apply_fused. */
size_t sdm_gn_apply_coop_state_bytes(void);
int sdm_gn_apply_coop(const void* src, int64_t ld_src, int src_raw, float alpha, const float* bias, int act, int images, int HW,
                      int C, int groups, float eps, const float* gamma, const float* beta, const float* mod, int mod_nstride,
                      const void* x0, int64_t ldx0, const void* x1, int64_t ldx1, int csplit, void* out, int64_t ldo, int bf16,
                      void* state, size_t state_bytes, sdm_stream_t s);

/* sdm_gn_adagn_bwd (or its bf16-rows form, bf16 = 1) in ONE launch for maps that fit 256 slabs of 4096 channel quads: the
 * slab's dout / y / z stay in registers, the per-group pair the apply needs crosses once (fp64 atomics + arrival counter,
 * the forward kernel's `state`), the per-channel sums go to `ws` ([images][C][2] fp32, zero on entry, zero on return) with
 * fire-and-forget atomics and the LAST slab of an image to leave turns them into dmod / dgamma / dbeta.  An image that
 * fits one slab uses no atomics at all.  Same results as sdm_gn_adagn_bwd up to the summation order of the group pair
 * (fp64 here).  Returns SDM_NOT_APPLICABLE (nothing launched) outside the envelope: C a power of two in 128..1024,
 * channels per group a power of two in 4..64, images * slabs <= 256. */
int sdm_gn_bwd_coop(const void* dout, int64_t lddo, const void* y, int64_t ldy, const void* z, int64_t ldz, int images, int HW,
                    int C, int groups, const float* gamma, const float* beta, const float* mod, int mod_nstride,
                    const float* mean, const float* rstd, void* dz, int64_t lddz, float* dmod, float* dgamma, float* dbeta,
                    float* ws, int bf16, void* state, size_t state_bytes, sdm_stream_t s);

/* The same pass inside a TRAINING step (small layers, where the step is a chain of dependent launches): additionally
 * leaves what the block's backward reads -- mean / rstd [images][groups] (both or neither) and, for raw sums, the conv's own
 * finished rows y_out (activated, pitch ldy) and z_out (pre-activation, optional; same pitch).  sdm_igemm(defer_epilogue)
 * may carry an out_preact pointer for that purpose: when it defers, it writes neither out nor out_preact. */
int sdm_gn_apply_fused_train(const float* src, int64_t ld_src, int src_raw, float alpha, const float* bias, int act,
                             int images, int HW, int C, int groups, float eps, const float* gamma, const float* beta,
                             const float* mod, int mod_nstride, const float* x0, int64_t ldx0, const float* x1,
                             int64_t ldx1, int csplit, float* out, int64_t ldo, float* z_out, float* y_out, int64_t ldy,
                             float* mean, float* rstd, sdm_stream_t s);

/* out[r][c] = scale[n][c]*y[r][c] + shift[n][c] + cat(x0,x1)[r][c]   (ResidualBlock tail, custom_layers.py:284-286);
 * x0 NULL = affine only (materialises a conv input whose GroupNorm+AdaGN affine is pending). */
int sdm_affine_residual(const float* y, int64_t ldy, const float* scale, const float* shift, int aff_nstride,
                        const float* x0, int64_t ldx0, const float* x1, int64_t ldx1, int csplit,
                        float* out, int64_t ldo, int images, int HW, int C, sdm_stream_t s);

/* Backward of y -> (scale*y+shift) with GroupNorm statistics and modulation (AdaGN backward):
 *   given dout (gradient w.r.t. the affine'd tensor), y, z (pre-SiLU of y, may be NULL):
 *   dz = silu'(z) * rstd*( dout*s*gamma - mean_g(dout*s*gamma) - xhat*mean_g(dout*s*gamma*xhat) )
 *   dmod[n][c]   = sum_hw dout*(xhat*gamma+beta+1)
 *   dgamma[c] += sum_{n,hw} dout*s*xhat ; dbeta[c] += sum_{n,hw} dout*s
 * ws: images*C*2 floats of per-channel sums, ZERO on entry and zero again when the call has run (zero-filled-scratch
 * contract, like the split-K workspace); coef: images*C*3 floats of scratch with no contract. */
int sdm_gn_adagn_bwd(const float* dout, int64_t lddo, const float* y, int64_t ldy, const float* z, int64_t ldz,
                     int images, int HW, int C, int groups, const float* gamma, const float* beta,
                     const float* mod, int mod_nstride, const float* mean, const float* rstd,
                     float* dz, int64_t lddz, float* dmod, float* dgamma, float* dbeta, float* ws, float* coef,
                     sdm_stream_t s);

/* Column softmax of attention scores (custom_layers.py:147: softmax(dim=1) = over QUERIES i for each key j).
 * scores [batch][S][lds] in place. */
int sdm_col_softmax(float* scores, int batch, int S, int64_t lds, sdm_stream_t s);
/* backward: ds = scale * p * (dp - sum_i p*dp) per column, in place on dp. */
int sdm_col_softmax_bwd(const float* p, float* dp, int batch, int S, int64_t lds, float scale, sdm_stream_t s);

/* Attention core for short sequences (S <= 256 tokens, head width a multiple of 16; custom_layers.py:144-150), two launches
 * on the exact-fp32 matrix cores for fp32 (bf16 = 0) or bf16 (bf16 = 1) rows.  qkv: [batch/heads * S][ld] rows with per-head
 * [q | k | v] chunks of 3*dk (custom_layers.py:138-141); batch = images * heads.
 *   sdm_attn_scores_small: P[b][i][j] = softmax over QUERIES i of (q_i . k_j) * scale -- one workgroup per 16 keys holds
 *     the scores of all queries on chip, so neither the scores nor a separate softmax pass touch HBM; P [batch][S][ldp]
 *     in the rows' type (bf16: columns [S, ldp) zero-filled, as sdm_col_softmax_bf16 leaves them);
 *   sdm_attn_pv_small: res[i][head*dk + c] = sum_j P[b][i][j] v_j[c]   (V read k-major: no transpose copy). */
int sdm_attn_scores_small(const void* qkv, int64_t ld, int batch, int S, int heads, int dk, float scale, void* P, int64_t ldp,
                          int bf16, sdm_stream_t s);
/* The same core when nothing but the result is kept (no-grad forwards: sampling), one call / two launches: one 16 x 16
 * score tile per workgroup (256 workgroups at S = 256) leaves raw scaled scores and per-tile column statistics in `ws`
 * (sdm_attn_small_nograd_workspace_bytes, no contract), the P.V launch folds the statistics and forms exp(s - m_j) / l_j
 * while loading its operand: the probabilities never exist in memory. */
size_t sdm_attn_small_nograd_workspace_bytes(int batch, int S);
int sdm_attn_small_nograd(const void* qkv, int64_t ld, int batch, int S, int heads, int dk, float scale, void* res, int64_t ldr,
                          int bf16, float* ws, size_t ws_bytes, sdm_stream_t s);
int sdm_attn_pv_small(const void* P, int64_t ldp, const void* qkv, int64_t ld, int batch, int S, int heads, int dk, void* res,
                      int64_t ldr, int bf16, sdm_stream_t s);

/* Folds the per-M-tile column statistics of sdm_igemm_bf16(col_stats) [batch][ntiles][N][2] into the vectors pass 2 takes:
 * neg_max[b][j] = -max_t m, inv_sum[b][j] = 1 / sum_t l_t exp(m_t - max). */
int sdm_attn_colstats_combine(const float* col_stats, int batch, int ntiles, int N, float* neg_max, float* inv_sum,
                              sdm_stream_t s);
/* M-tile height sdm_igemm_bf16 will use for this problem (the ntiles of col_stats = ceil(M / that)). */
int sdm_igemm_bf16_tile_m(const sdm_igemm_desc* d);

/* Sinusoidal timestep features (custom_layers.py:84-90): out[n][D] = [sin(t f_i) | cos(t f_i)]. t is int64;
 * freq[D/2] is the frequency table exp(-i ln(1e4)/(D/2-1)), computed once on the host exactly as the reference does. */
int sdm_sinusoid(const int64_t* t, const float* freq, int n, int D, float* out, sdm_stream_t s);
/* Small-batch Linear (+SiLU) for the embedding MLPs and the y_scale projections
 * (custom_layers.py:59-78, :38): out[m][o] = act(sum_k x[m][k] w[o][k] + b[o]) (+ add[m or 0][o]).
 * add (optional) has add_rows rows of pitch ldadd: either M rows, or ONE row with ldadd == 0 that is broadcast over the
 * batch (time embedding of a scalar t added to per-image label embeddings, custom_layers.py:97-98); anything else is
 * rejected. */
int sdm_linear_small(const float* x, int64_t ldx, const float* w, const float* b, const float* add, int64_t ldadd,
                     int add_rows, float* out, int64_t ldo, float* out_preact, int M, int K, int O, int act, sdm_stream_t s);
/* Batched form: njobs projections sharing x (the 16 y_scale GEMVs in one launch). jobs is a DEVICE array. */
typedef struct sdm_linear_job { const float* w; const float* b; float* out; int O; int pad; } sdm_linear_job;
int sdm_linear_multi(const float* x, int64_t ldx, int M, int K, const sdm_linear_job* jobs_dev, int njobs,
                     int maxO, sdm_stream_t s);

/* Data gradient of njobs projections that shared one input (the y_scale GEMVs): dx[m][k] += sum_j sum_o dy_j[m][o] w_j[o][k].
 * jobs[j] = {w_j, (unused), dy_j (pitch O_j), O_j}; chunk_job/chunk_o0 (DEVICE int arrays, nchunks long) map each
 * 64-wide output chunk to its job and first output column. */
int sdm_linear_multi_bwd_dx(const sdm_linear_job* jobs_dev, const int* chunk_job, const int* chunk_o0, int nchunks,
                            float* dx, int64_t lddx, int M, int K, sdm_stream_t s);

/* backward of a small-batch linear given dy (already multiplied by act'):
 *   dx[m][k] += sum_o dy[m][o] w[o][k] ; dw[o][k] += sum_m dy[m][o] x[m][k] ; db[o] += sum_m dy[m][o]
 * (all three ACCUMULATE: hand in zero-filled buffers for plain results).  dx / dw may be NULL to skip. */
int sdm_linear_small_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* w, float* dx,
                         int64_t lddx, int dx_accumulate, float* dw, float* db, int M, int K, int O, sdm_stream_t s);

/* Generic helpers used by backward. */
/* out[c] (+)= sum_r x[r][c]  (bias gradients) */
int sdm_colsum(const float* x, int64_t ldx, int rows, int C, float* out, int accumulate, sdm_stream_t s);
/* y = silu(x) elementwise (standalone Swish, custom_layers.py:18-20; inside the U-Net it is fused into epilogues) */
int sdm_silu(const float* x, float* y, size_t n, sdm_stream_t s);
/* dx = silu'(z) * dy  (elementwise, contiguous n) ; z == NULL -> copy */
int sdm_silu_bwd(const float* dy, const float* z, float* dx, size_t n, sdm_stream_t s);
/* y = a + b elementwise */
int sdm_add(const float* a, const float* b, float* y, size_t n, sdm_stream_t s);
/* MSE loss of NCHW tensors: loss_out[0] = mean((a-b)^2); grad (may be NULL) = 2(a-b)/n.  F.mse_loss, train_diffusion.py:350 */
int sdm_mse(const float* a, const float* b, float* loss_out, float* grad, size_t n, sdm_stream_t s);
/* d tanh: dx = dy * (1 - y^2) */
int sdm_tanh_bwd(const float* dy, const float* y, float* dx, size_t n, sdm_stream_t s);

/* Forward diffusion q(x_t|x_0) (degraders.py:51-59,96-104): xt = sqrt(ab[n]) x0 + sqrt(1-ab[n]) eps, NCHW. */
int sdm_q_sample(const float* x0, const float* eps, const float* alpha_bar, float* xt, int images, size_t per_image,
                 sdm_stream_t s);
/* DDPM reverse update (diffusion_sampling_algorithms.py:42-55). coef is a DEVICE array of 3 floats
 * {1/sqrt(alpha_t), (1-alpha_t)/sqrt(1-abar_t), sigma_t} (device-resident so a captured graph can be replayed
 * with new coefficients; sigma_t = 0 encodes the reference's z = 0 for noise_step <= 1). x is updated in place.
 * Noise: `z` (injected, trajectory tests) or, with z == NULL and rng_step != NULL, drawn IN the kernel -- the reference's
 * torch.randn_like(x_t) (:43) -- from Philox4x32-10 with counter {element/4 lo, hi, (uint32)*rng_step, 0} and key = seed,
 * 4 normals per call by Box-Muller (SURVEY 8b "sdm_sampler_update_ddpm (fused elementwise + Philox)"); rng_step is a
 * DEVICE word that differs from step to step and is stable during one (the sampler passes t_cur of sdm_step_begin);
 * z == NULL and rng_step == NULL: no noise.  `cursor` (optional, DEVICE): the step cursor of sdm_step_begin, advanced here
 * (saturating at nsteps) -- the step's last launch moves the captured graph on to the next table row. */
int sdm_ddpm_update(float* x, const float* eps, const float* z, const float* coef, size_t n, unsigned long long seed,
                    const int64_t* rng_step, int* cursor, int nsteps, sdm_stream_t s);
/* DDIM update (diffusion_sampling_algorithms.py:111-136, eta=0): coef (DEVICE, 5 floats) = {1/sqrt(abar_t),
 * sqrt(1-abar_t), sqrt(abar_tm1), sqrt(1-abar_tm1), update_flag}.  x0_out (optional) receives the x0 estimate;
 * x is advanced to x_{t-1} only when update_flag != 0 (the reference skips the update on the last step). */
int sdm_ddim_update(float* x, const float* eps, const float* coef, float* x0_out, size_t n, int* cursor, int nsteps,
                    sdm_stream_t s);   /* cursor: as in sdm_ddpm_update (optional) */

/* Classifier-free guidance (build-side extension, BASELINE config 3; not in the reference): the U-Net is evaluated once
 * on a 2N batch [conditional | unconditional(all-zero label)]; out[i] = e_u[i] + w*(e_c[i] - e_u[i]). */
int sdm_cfg_combine(const float* eps_2n, float* out, float w, size_t n_half, sdm_stream_t s);

/* Cold-diffusion update of the super-resolution sampler (diffusion_sampling_algorithms.py:187-209):
 *   x <- x - (c0*x0 + c1*noise) + (c2*x0 + c3*noise),  coef (DEVICE, 5 floats) = {sqrt(abar_t), sqrt(1-abar_t),
 *   sqrt(abar_tm1), sqrt(1-abar_tm1), update_flag}; skipped when update_flag == 0 (last step). */
int sdm_cold_update(float* x, const float* x0_hat, const float* noise, const float* coef, size_t n, float* x0_keep,
                    int* cursor, int nsteps, sdm_stream_t s);   /* x0_keep (optional): copy of x0_hat kept for the caller
                                                                  (the sampler's return value); cursor: as above */
/* F.interpolate(mode="area") on NCHW fp32 == adaptive average pooling (train_SR_diffusion.py:321-328): output pixel
 * (i,j) averages input rows [floor(i*Hin/Hout), ceil((i+1)*Hin/Hout)) x the same for columns; integer
 * down-scaling is block averaging, integer up-scaling is pixel replication. */
int sdm_area_resize(const float* x, float* out, int planes, int Hin, int Win, int Hout, int Wout, sdm_stream_t s);

/* Input pipeline on the device (replaces cv2.imread + DataLoader workers + per-image hflip, train_diffusion.py:148-152,
 * 312-314; img_dataset.py:26-36): the decoded dataset lives in HBM as uint8 [images][H][W][C] (BGR, as cv2 loads it);
 * out[b][c][y][x] = lut256[cache[idx[b]][y][flip[b] ? W-1-x : x][c]] -- gather + normalise + horizontal flip + HWC->CHW in
 * one pass.  lut256 is the host-computed table (v - 127.5) / 127.5 (float64 then rounded to float32, as the reference
 * does per pixel); flip may be NULL. */
int sdm_u8_gather_nchw(const uint8_t* cache_nhwc, const int64_t* idx, const uint8_t* flip, const float* lut256,
                       float* out_nchw, int batch, int C, int H, int W, sdm_stream_t s);

/* Head of a hipGraph-captured sampling step.  c = min(*cursor, nsteps-1) is the step being run (the cursor is READ
 * here and advanced by the step's update kernel, sdm_ddpm/ddim/cold_update): t_cur[0] = t_table[c];
 * coef_cur[0..ncoef) = coef_table[c][0..ncoef); and, with mod_time != NULL, the AdaGN modulation rows of the step
 *   mods_cur[i][0..width) = mod_time[c][0..width) (+ mod_cond[i][0..width)),  i < rows
 * from tables computed ONCE per sampling run: the time / label embedding MLPs and every y_scale projection
 * (custom_layers.py:82-98, :40-44) depend on t and the labels only, and y_scale is linear, so
 * W (te_t + ce_i) + b = (W te_t + b) + W ce_i.  With it the captured graph (begin -> U-Net convs -> update) is parameter-free
 * and is replayed once per denoise step (the python for-loops of diffusion_sampling_algorithms.py:21-63, :91-145). */
int sdm_step_begin(const int* cursor, const int64_t* t_table, const float* coef_table, int ncoef, int nsteps, int64_t* t_cur,
                   float* coef_cur, const float* mod_time, const float* mod_cond, float* mods_cur, int rows, int width,
                   sdm_stream_t s);
/* ++*cursor (saturating at nsteps) as a launch of its own: steps whose update does not go through the kernels above. */
int sdm_step_advance(int* cursor, int nsteps, sdm_stream_t s);

/* Fused multi-tensor Adam (torch.optim.Adam defaults, train_diffusion.py:214-217): jobs is a DEVICE array.
 * w16 / w16t (optional, bf16 training path): the kernel also writes the bf16 copy of the updated parameter (same dense
 * element order) and / or its transposed copy -- the parameter seen as [n / (R * Cc)][R][Cc], the copy as [..][Cc][ldd]
 * with ldd >= R (columns [R, ldd) are not touched: the first sdm_cast_bf16_t zero-filled them) -- so the next step's
 * contractions find their weight copies without the ~470 cast launches per step. */
typedef struct sdm_adam_job {
  float* p; const float* g; float* m; float* v; int64_t n;
  uint16_t* w16; uint16_t* w16t; int R, Cc, ldd, reserved;
} sdm_adam_job;
int sdm_adam_multi(const sdm_adam_job* jobs_dev, int njobs, int64_t max_n, float lr, float beta1, float beta2,
                   float eps, float bias_c1, float bias_c2, float grad_scale, const float* hyper_dev, sdm_stream_t s);
/* hyper_dev (optional, DEVICE, 2 floats {lr, step}): when given, lr and the bias corrections 1-beta^step are read /
 * computed on the device, so a captured hipGraph of the training step can be replayed while lr and the step advance. */

/* ---- bf16 storage path of the forward (sampling; BASELINE configs 3 and 5 name bf16) -------------------------------
 * Activations and contraction weights are bf16 bit patterns (uint16_t) in HBM; accumulators, GroupNorm statistics,
 * softmax, biases, modulation tables and the sampler state stay fp32.  The reference has no bf16 mode (its GPU path
 * is fp16 autocast, train_diffusion.py:333): the parity gate of this path is rel-L2 <= 1e-2 against the fp32 oracle. */

/* Same contraction as sdm_igemm with a.p0 / a.p1 / b / res / res2 / out_preact / dact_z (and out unless out_f32) pointing at
 * bf16 data; every pitch and stride counts ELEMENTS.  Restrictions: b K-contiguous (b_kmajor = 0: a data gradient
 * contracts against a transposed weight copy), K and csplit multiples of 8, rows 16-byte aligned, no affine-on-load.
 * out_f32 != 0 writes fp32 (attention scores ahead of the softmax; no residual / out_preact then).  Split-K scratch: fp32,
 * zero-filled, sdm_igemm_bf16_workspace_bytes(d). */
int sdm_igemm_bf16(const sdm_igemm_desc* d, int out_f32, sdm_stream_t s);
size_t sdm_igemm_bf16_workspace_bytes(const sdm_igemm_desc* d);
/* fp32 -> bf16, round to nearest even (weights, once per checkpoint). */
int sdm_cast_bf16(const float* src, uint16_t* dst, size_t n, sdm_stream_t s);
/* sdm_stem_conv / sdm_head_conv with the NHWC side in bf16 (weights, bias and the NCHW image stay fp32); out_preact
 * (optional, bf16) receives the pre-activation for the backward of the bf16 training path. */
int sdm_stem_conv_bf16(const float* x_nchw, const float* w_tap, const float* bias, uint16_t* out_nhwc, uint16_t* out_preact,
                       int images, int Cin, int H, int W, int Cout, int act, sdm_stream_t s);
int sdm_head_conv_bf16(const uint16_t* x_nhwc, const float* w_tap, const float* bias, float* out_nchw,
                       int images, int Cin, int H, int W, int Cout, int act, sdm_stream_t s);
/* The head conv on the matrix cores (csrc/sdm_edge_bf16.hip): bf16 rows x_nhwc (row pitch ldx elements, 16-byte aligned, the
 * first Cin channels used; Cin = 64 or 128, otherwise SDM_ERR_UNSUPPORTED) and bf16 taps w16_tap [9][Cout][Cin] (Cout <= 16),
 * fp32 bias, fp32 NCHW result with the activation applied.  Replaces nn.Conv2d(min_channel, out_channel, 3, 1, 1) (+ nn.Tanh)
 * at the end of U_Net.out_layers (reference models/U_Net.py:121-130). */
int sdm_head_conv_mfma_bf16(const uint16_t* x_nhwc, long long ldx, const uint16_t* w16_tap, const float* bias, float* out_nchw,
                            int images, int Cin, int H, int W, int Cout, int act, sdm_stream_t s);
/* sdm_gn_apply_fused on bf16 rows: src = the conv's finished bf16 rows (src_raw 0, pitch ld_src in elements) or the fp32
 * raw split-K sums left by sdm_igemm_bf16(defer_epilogue = 1) (src_raw 1; rounded to bf16 after bias / activation, as the
 * unfused path does, and cleared); x0 / x1 / out bf16. */
int sdm_gn_apply_fused_bf16(const void* src, int64_t ld_src, int src_raw, float alpha, const float* bias, int act,
                            int images, int HW, int C, int groups, float eps, const float* gamma, const float* beta,
                            const float* mod, int mod_nstride, const uint16_t* x0, int64_t ldx0, const uint16_t* x1,
                            int64_t ldx1, int csplit, uint16_t* out, int64_t ldo, sdm_stream_t s);
/* GroupNorm + AdaGN fold (the tables of sdm_gn_affine) from the per-(image, channel) {sum, sum of squares} a conv epilogue
 * emitted (sdm_igemm_desc.gn_sums): no pass over the activation.  The sums are cleared again (zero-filled contract).  mean /
 * rstd optional (training). */
int sdm_gn_finalize(float* sums, int images, int HW, int C, int groups, float eps, const float* gamma, const float* beta,
                    const float* mod, int mod_nstride, float* scale, float* shift, float* mean, float* rstd, sdm_stream_t s);
/* fp32 rows [images*H*W][ld] (first C columns) -> NCHW fp32: lays out the head conv's result as the reference returns it
 * when out_layers.1 runs on the matrix cores (sdm_igemm_bf16 with fp32 output). */
int sdm_rows_to_nchw(const float* rows, int64_t ld, float* out_nchw, int images, int C, int H, int W, sdm_stream_t s);
/* sdm_gn_affine on bf16 rows (fp32 tables out); mean / rstd [images][groups] are optional (both or neither): the training
 * path's backward needs them, sampling passes NULL. */
int sdm_gn_affine_bf16(const uint16_t* y, int64_t ldy, int images, int HW, int C, int groups, float eps,
                       const float* gamma, const float* beta, const float* mod, int mod_nstride,
                       float* scale, float* shift, float* mean, float* rstd, sdm_stream_t s);
/* out = scale*y + shift (+ cat(x0,x1)) on bf16 rows; x0 NULL = affine only (materialises the next conv's input). */
int sdm_affine_residual_bf16(const uint16_t* y, int64_t ldy, const float* scale, const float* shift, int aff_nstride,
                             const uint16_t* x0, int64_t ldx0, const uint16_t* x1, int64_t ldx1, int csplit,
                             uint16_t* out, int64_t ldo, int images, int HW, int C, sdm_stream_t s);
/* Column softmax (over queries i for every key j, custom_layers.py:147) of fp32 scores [batch][S][lds] into bf16
 * probabilities [batch][S][ldp]; columns [S, ldp) are zero-filled (reduction padding of P.V). */
int sdm_col_softmax_bf16(const float* scores, uint16_t* P, int batch, int S, int64_t lds, int64_t ldp, sdm_stream_t s);
/* dst[b][c][r] = src[b / bdiv * src_bs0 + b % bdiv * src_bs1][r][c]; dst columns [R, ld_dst) zero-filled.  Puts V with
 * the token index contiguous for P.V (the matrix cores take both operands contiguous along the reduction). */
int sdm_transpose_bf16(const uint16_t* src, int64_t ld_src, int64_t src_bs0, int64_t src_bs1, int bdiv,
                       uint16_t* dst, int64_t ld_dst, int64_t dst_bs, int batch, int R, int Cc, sdm_stream_t s);

/* ---- bf16 TRAINING path (BASELINE configs 4-5; reference analogue: fp16 autocast + GradScaler, train_diffusion.py:333,
 * 358-364).  fp32 master weights, fp32 weight gradients and optimizer state; activations, activation gradients and the
 * weight copies read by the contractions are bf16; all sums fp32.  Forward and data gradient run sdm_igemm_bf16 (the
 * data gradient against a transposed weight copy, sdm_cast_bf16_t, with out_preact / dact_z / res pointing at bf16 rows);
 * what follows is the rest of the backward.  Parity: the reference has no bf16 mode, gate = rel-L2 <= 2e-2 per gradient
 * tensor against the fp32 oracle ("parity unpinned" for the rounding itself). */

/* Weight gradient with bf16 operands and an fp32 result: same descriptor as sdm_wgrad, g.p0 / x.p0 / x.p1 point at bf16
 * rows (pitches and strides in ELEMENTS, multiples of 8; rows 16-byte aligned and physically >= their channel count
 * rounded up to 8; concat split a multiple of 8; no pending affine), dw / dbias fp32.  Both operands are K-major row
 * tiles staged by LDS-DMA and read with the transposing LDS read ds_read_b64_tr_b16. */
int sdm_wgrad_bf16(const sdm_wgrad_desc* d, sdm_stream_t stream);
/* dst[t][c][r] = bf16(src[t][r][c]) for t < T, dst rows of pitch ldd >= R with [R, ldd) zero-filled: transposed bf16 weight
 * copy (tap-major conv weights [tap][c_out][c_in] -> [tap][c_in][c_out]; Linear [O][K] -> [K][O] with T = 1; the head
 * conv's 3 output channels padded to ldd = 8, the reduction granule of the bf16 contraction). */
int sdm_cast_bf16_t(const float* src, uint16_t* dst, int T, int R, int Cc, int ldd, sdm_stream_t s);
/* sdm_gn_adagn_bwd with dout / y / z / dz as bf16 rows (C %% 8 == 0, C <= 2048); sums, coefficients, dmod / dgamma / dbeta
 * fp32; same zero-filled ws contract. */
int sdm_gn_adagn_bwd_bf16(const uint16_t* dout, int64_t lddo, const uint16_t* y, int64_t ldy, const uint16_t* z, int64_t ldz,
                          int images, int HW, int C, int groups, const float* gamma, const float* beta,
                          const float* mod, int mod_nstride, const float* mean, const float* rstd,
                          uint16_t* dz, int64_t lddz, float* dmod, float* dgamma, float* dbeta, float* ws, float* coef,
                          sdm_stream_t s);
/* Column-softmax backward: dS = scale * P * (dP - sum_i P*dP) per column; P bf16 [batch][S][ldp], dP fp32 [batch][S][lddp],
 * dS bf16 [batch][S][ldds] with columns [S, ldds) zero-filled. */
int sdm_col_softmax_bwd_bf16(const uint16_t* P, int64_t ldp, const float* dP, int64_t lddp, uint16_t* dS, int64_t ldds,
                             int batch, int S, float scale, sdm_stream_t s);
/* dx = silu'(z) * dy on contiguous bf16 data; out[c] += sum_r x[r][c] for bf16 rows (bias gradients, fp32 sums). */
int sdm_silu_bwd_bf16(const uint16_t* dy, const uint16_t* z, uint16_t* dx, size_t n, sdm_stream_t s);
int sdm_colsum_bf16(const uint16_t* x, int64_t ldx, int rows, int C, float* out, sdm_stream_t s);
/* Edge-conv gradients with the NHWC side in bf16 (NCHW side, weights and weight gradients fp32): see sdm_stem_conv_wgrad,
 * sdm_head_conv_dgrad, sdm_head_conv_wgrad. */
int sdm_stem_conv_wgrad_bf16(const float* x_nchw, const uint16_t* g_nhwc, float* dw_tap, float* dbias, float* ws,
                             int images, int Cin, int H, int W, int Cout, sdm_stream_t s);
int sdm_head_conv_dgrad_bf16(const float* dy_nchw, const float* w_tap, uint16_t* dx_nhwc, const uint16_t* dact_z,
                             int images, int Cin, int H, int W, int Cout, sdm_stream_t s);
int sdm_head_conv_wgrad_bf16(const uint16_t* x_nhwc, const float* g_nchw, float* dw_tap, float* dbias, float* ws,
                             int images, int Cin, int H, int W, int Cout, sdm_stream_t s);
/* NCHW fp32 [images][C][H][W] -> NHWC bf16 rows [images*H*W][Cp], channels [C, Cp) zero (Cp a multiple of 8): the 3- / 6-
 * channel side of the edge convs in the row format of the matrix-core kernels, whose weight / data gradients then run on
 * sdm_wgrad_bf16 / sdm_igemm_bf16 (U_Net.py:55-60,121-128 backward). */
int sdm_nchw_to_rows_bf16(const float* x_nchw, uint16_t* out_rows, int images, int C, int H, int W, int Cp, sdm_stream_t s);

/* ---- data-parallel gradient exchange, communicator owned by the library (north_star: "an RCCL all-reduce of
 * U-Net gradients over xGMI overlapped with the optimizer step").  The reference has no counterpart: it trains on one
 * device (train_diffusion.py:49).  RCCL is bound at run time (librccl.so.1, the copy the process already holds); without
 * it every call returns SDM_ERR_COMM.  One communicator per process and GPU; rank 0 makes the 128-byte id
 * (sdm_dp_unique_id) and the launcher hands it to every rank (torch.distributed broadcast, a file, the environment). */
typedef struct sdm_dp_comm* sdm_dp_comm_t;
#define SDM_DP_ID_BYTES 128
/* > 0: RCCL's version code (it can be loaded); < 0: SDM_ERR_COMM with the reason */
int sdm_dp_available(void);
int sdm_dp_unique_id(void* out_id, size_t bytes);
/* Collective over all `world` ranks, on the calling thread's current HIP device. */
int sdm_dp_init(const void* id, size_t bytes, int rank, int world, sdm_dp_comm_t* out);
int sdm_dp_rank(sdm_dp_comm_t c);
int sdm_dp_world(sdm_dp_comm_t c);
/* Sum (average != 0: mean) of the fp32 gradient slice buf[0, count) over all ranks, in place, enqueued on `stream`
 * (any 4-byte aligned start: bucket boundaries fall on arbitrary elements).  wire_bf16 == 0: one ncclAllReduce in fp32.  wire_bf16 != 0: the mesh exchange for the
 * switchless xGMI topology -- shards rounded to bf16, all-to-all over W - 1 point-to-point transfers, FP32 sum of the W
 * copies on the owning rank (rank order: deterministic), one rounding of the sum, bf16 all-gather; staging comes from the
 * communicator (sdm_dp_reserve sizes it ahead of time; otherwise the first bucket of a new maximum size allocates). */
int sdm_dp_allreduce_bucket(sdm_dp_comm_t c, float* buf, int64_t count, int wire_bf16, int average, sdm_stream_t stream);
int sdm_dp_reserve(sdm_dp_comm_t c, int64_t max_count);
/* 0 while the communicator is healthy; SDM_ERR_COMM once RCCL has recorded an asynchronous failure (a lost peer, a
 * network error): the trainer polls it once per step instead of hanging in the next collective. */
int sdm_dp_async_error(sdm_dp_comm_t c);
/* Collective; frees the staging and the communicator.  The handle is dead afterwards. */
int sdm_dp_finalize(sdm_dp_comm_t c);

#ifdef __cplusplus
}
#endif
#endif
