/*
 * sscg.h - C ABI of libsscg.so: the MI355X (gfx950) kernels behind the CycleGAN training step of
 * arnab39/Semi-supervised-segmentation-cycleGAN (model.py:370-552, `semisuper_cycleGAN.train`).
 *
 * The reference has no FFI of its own: every arithmetic op on its hot path is a stock torch.nn module
 * (SURVEY.md section 8(b)).  Each entry point below therefore names the torch call site it replaces
 * (reference file:line).  INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: device pointers + sizes, no torch types.  Tensors are channels-last (NHWC: [N][H][W][C]); conv weights
 *     [K][R][S][C] (= torch `[K,C,R,S]` in channels_last); labels int64.  Activations and conv weights are fp32
 *     (SSCG_F32, the reference's dtype and BASELINE config 2) or bfloat16 (SSCG_BF16, BASELINE configs 3/5): entry points
 *     that touch them take `void*` plus a dtype code.  Statistics, losses, biases, weight gradients, optimiser state: fp32.
 *   - `stream` is a hipStream_t passed as void*.  Calls are asynchronous and stream ordered, re-entrant,
 *     allocate nothing and keep no mutable state (tile-class / split overrides of tools and tests travel in
 *     sscg_conv_desc.tuning / wgrad_tuning); scratch memory is caller provided (`ws`) and sized by the matching
 *     *_workspace() query.
 *   - return value: 0 = ok, <0 = library error (SSCG_ERR_*), >0 = hipError_t.  Never throws/aborts.
 */
#ifndef SSCG_H
#define SSCG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSCG_ABI_VERSION 17

/* element types of activation / weight tensors */
#define SSCG_F32 0
#define SSCG_BF16 1
#define SSCG_BF16X3 2 /* conv WEIGHT operands only: an fp32 weight split into three bfloat16 planes h + m + l (sscg_split3), the operand
                       * of the fp32-accurate "split" contraction on the bf16 matrix cores */

#define SSCG_ERR_BAD_ARG (-1)
#define SSCG_ERR_UNSUPPORTED (-2)
#define SSCG_ERR_WORKSPACE (-3)

/* activation codes (conv epilogue, norm apply) */
#define SSCG_ACT_NONE 0
#define SSCG_ACT_RELU 1  /* nn.ReLU            arch/ops.py:50,57 */
#define SSCG_ACT_LRELU 2 /* nn.LeakyReLU(0.2)  arch/ops.py:44, arch/discriminators.py:46,71,74 */
#define SSCG_ACT_TANH 3  /* nn.Tanh            arch/generators.py:91 */

#define SSCG_PAD_ZEROS 0
#define SSCG_PAD_REFLECT 1 /* nn.ReflectionPad2d folded into the conv loader: arch/ops.py:62,67; arch/generators.py:73,84,89 */

int sscg_abi_version(void);
/* Measurement aid (no reference counterpart): on != 0 turns every kernel launch of the library into a no-op while the host side of
 * each entry point (checks, planning, workspace carving) still runs; returns the previous setting.  bench.py times the host's issue
 * cost of a training step with it (no back-pressure from the device).  Results are undefined while it is on. */
int sscg_set_dry_run(int on);

/* ------------------------------------------------------------------ convolution (K1, K2, K5) */
typedef struct sscg_conv_desc {
    int32_t N, H, W, C; /* input  [N][H][W][C] */
    int32_t K, R, S;    /* weight [K][R][S][C] */
    int32_t P, Q;       /* output [N][P][Q][K];  P = (H + 2*pad - dil*(R-1) - 1)/stride + 1 */
    int32_t stride, pad, dil;
    int32_t pad_mode;   /* SSCG_PAD_* (reflect: forward and wgrad only) */
    int32_t act;        /* fused epilogue activation of the forward */
    float slope;        /* LeakyReLU slope */
    int32_t x_dtype;    /* SSCG_F32 / SSCG_BF16 of the [N][H][W][C] tensor (forward input, dgrad output, wgrad x) */
    int32_t w_dtype;    /* ... of the weight operand handed to forward ([K][R][S][C]) / dgrad ([C][R][S][K]) */
    int32_t y_dtype;    /* ... of the [N][P][Q][K] tensor (forward output, dgrad / wgrad dy) */
    int32_t precision;  /* fp32 tensors only: 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32); 1 = operands rounded to bf16
                         * (RNE) between LDS and the matrix cores, v_mfma_f32_32x32x16_bf16, fp32 accumulation; 2 (weight
                         * gradient only) = both operands split into three bf16 pieces between LDS and the matrix cores, six
                         * exact piece products accumulated in fp32 (fp32-accurate; the two LDS-DMA tile classes, exact fp32
                         * elsewhere).  Forward / data gradient select the split contraction through w_dtype = SSCG_BF16X3. */
    int32_t tuning;     /* 0 = the library's own plan.  Tuning / test aid carried by the call (the library keeps no mutable state):
                         * bits 0..7 = 1 + forced tile class of the forward / data-gradient kernel family that serves the call,
                         * bits 8..15 = split-K (1 = never, n > 1 = every tile cut in n) */
    int64_t w_plane;    /* w_dtype == SSCG_BF16X3: elements between two planes of the weight operand; 0 = K*R*S*C (dense) */
    int32_t wgrad_tuning; /* 0 = the library's cost model.  bits 0..7 = 1 + forced weight-gradient tile class (0 = 128x128,
                         * 1 = 64x64) with bits 8..23 = pixel splits; precision 2 only: class 2 = never the pre-split-planes
                         * kernel (operands split on the fly), class 3 = the planes kernel also for 1x1 filters; bits 24..31 =
                         * kernel-variant switches (bf16 weight gradient: tools/conv16_bench.py; planes kernel: bit 24 = two
                         * copy stages) */
} sscg_conv_desc;
/* Supported dtype combinations.  forward: (x, w) both fp32 -> y fp32|bf16 (fp32 MFMA kernel: stems and few-channel
 * inputs); (x, w) both bf16 with C % 64 == 0 -> y fp32|bf16 (bf16 MFMA kernel, bf16 LDS tiles).  dgrad: the same with
 * (dy, wt) as the operands and dx as the result (K % 64 == 0 for bf16).  wgrad: x, dy each fp32|bf16, dw fp32. */

/* nn.Conv2d forward: arch/ops.py:43,49,68; arch/generators.py:85,90,325,331,336,373,388,415;
 * arch/discriminators.py:45,58,70-75.  y = act(conv(x, w) + bias); bias may be NULL. */
size_t sscg_conv2d_fwd_workspace(const sscg_conv_desc* d);   /* split-K scratch for few-channel heads; may be 0 */
int sscg_conv2d_fwd(const sscg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, void* ws,
                    size_t ws_bytes, void* stream);

/* Forward with the statistics of the normalisation layer that follows fused into the epilogue ("Conv + InstanceNorm /
 * BatchNorm" blocks, arch/ops.py:40-57, arch/generators.py:345-365): besides y, per-tile column sums of y and y^2 (fp64,
 * taken from the fp32 accumulators) go to `stats`; sscg_norm_stats_from_conv turns them into mean / rstd (and the
 * running-statistics update) without reading y again.  The output rows are viewed as G groups of L rows (G*L = N*P*Q).
 * sscg_conv2d_fwd_stats_bytes returns 0 when the fusion does not apply to this geometry (then use sscg_norm_stats). */
size_t sscg_conv2d_fwd_stats_bytes(const sscg_conv_desc* d, int G, int64_t L);
size_t sscg_conv2d_fwd_stats_workspace(const sscg_conv_desc* d);
int sscg_conv2d_fwd_stats(const sscg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int G, int64_t L,
                          void* stats, size_t stats_bytes, void* ws, size_t ws_bytes, void* stream);
int sscg_norm_stats_from_conv(const sscg_conv_desc* d, const void* stats, int G, int64_t L, float eps, float* mean, float* rstd,
                              float* running_mean, float* running_var, float momentum, void* stream);

/* PixelDiscriminator's front half as ONE launch (arch/discriminators.py:70-73: nn.Conv2d(input_nc, ndf, 1) -> nn.LeakyReLU(0.2) ->
 * nn.Conv2d(ndf, 2 ndf, 1) [-> the statistics of the InstanceNorm / BatchNorm layer at :73]).  `d` describes the SECOND conv (C = 64
 * source channels, 1x1, stride 1, no padding, fp32 tensors, w = its SSCG_BF16X3 split planes); `xf` [N*H*W][cin] fp32 is the FIRST
 * conv's input, (w1 [64][cin], b1 [64] or NULL, slope1) its parameters and LeakyReLU slope.  The 64-channel map between the two convs
 * is formed per workgroup in LDS (exact fp32 FMAs) and is written to `h1` ([N*H*W][64] fp32) only when the caller passes it (a
 * backward pass that wants it stored) - the launch itself never reads it back.  G > 0: `stats` receives the records of
 * sscg_conv2d_fwd_stats for `d` (same size, finalised by sscg_norm_stats_from_conv).  sscg_conv2d_front_applies: cin in {3, 4, 20, 21}
 * and a geometry whose tile class carries the fused prologue (else run the two convs separately). */
int sscg_conv2d_front_applies(const sscg_conv_desc* d, int cin);
int sscg_conv2d_front_fwd(const sscg_conv_desc* d, const void* xf, int cin, const float* w1, const float* b1, float slope1, void* h1,
                          const void* w, const float* bias, void* y, int G, int64_t L, void* stats, size_t stats_bytes, void* stream);

/* Data gradient of the same conv (autograd of model.py:472,539), and nn.ConvTranspose2d forward
 * (arch/ops.py:55-56): dx = act(dgrad(dy, wt) + bias).  `wt` = weight re-laid as [C][R][S][K]
 * by sscg_weight_krsc_to_crsk.  bias NULL / act NONE for a pure gradient. */
size_t sscg_conv2d_dgrad_workspace(const sscg_conv_desc* d);
int sscg_conv2d_dgrad(const sscg_conv_desc* d, const void* dy, const void* wt, const float* bias, void* dx,
                      int act, float slope, void* ws, size_t ws_bytes, void* stream);

/* Weight gradient: dw = beta*dw + wgrad(x, dy); dw is fp32 [K][R][S][C]. */
size_t sscg_conv2d_wgrad_workspace(const sscg_conv_desc* d);
int sscg_conv2d_wgrad(const sscg_conv_desc* d, const void* x, const void* dy, float* dw, float beta, void* ws,
                      size_t ws_bytes, void* stream);

/* [K][RS][C] -> [C][RS][K]; source and destination dtypes may differ (fp32 master weight -> bf16 operand copy) */
int sscg_weight_krsc_to_crsk(const void* w, int w_dtype, void* wt, int wt_dtype, int K, int RS, int C, void* stream);
/* The same re-layout of MANY weights in one launch (the operand copies of every conv that sees a backward pass are rebuilt after each
 * optimiser step: ~230 launches of a few microseconds each at the top of the step otherwise).  `jobs` is a table IN DEVICE MEMORY;
 * job i covers the 32x32 tiles [block0, block0 of job i+1) in the order (c-tile fastest, then k-tile, then tap) - block0 ascending,
 * job 0 at 0, `n_blocks` the total.  Same dtype pairs and bit-identical results as sscg_weight_krsc_to_crsk. */
typedef struct sscg_wt_job {
    const void* w;
    void* wt;
    int32_t w_dtype, wt_dtype;
    int32_t K, RS, C;
    int32_t block0;
} sscg_wt_job;
int sscg_weight_krsc_to_crsk_batch(const sscg_wt_job* jobs, int n_jobs, int n_blocks, void* stream);
/* The "split" contraction (fp32 accuracy on the bf16 matrix cores): dst = three bfloat16 planes h, m, l (each n elements,
 * `plane_stride` elements apart) with src[i] = h[i] + m[i] + l[i] to 2^-24 (round to nearest at every step).  A conv weight in
 * this form is passed with w_dtype = SSCG_BF16X3 (forward: planes of [K][R][S][C]; dgrad: planes of [C][R][S][K], which
 * sscg_weight_krsc_to_crsk produces directly with wt_dtype = SSCG_BF16X3). */
int sscg_split3(const float* src, void* dst, int64_t n, int64_t plane_stride, void* stream);
/* 1 when the split kernels serve this call (kind 0 = forward, 1 = data gradient; dtypes of `d` are ignored: fp32 tensors are
 * implied), 0 when the caller should use the exact-fp32 path (few-channel stems and heads, ragged channels).  (The weight
 * gradient takes precision = 2 for every geometry and falls back to exact fp32 by itself.) */
int sscg_conv2d_split_applies(const sscg_conv_desc* d, int kind);
/* dst[i] = (dst_dtype) src[i] */
int sscg_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);

/* out[c] = beta*out[c] + sum_r x[r][c]  (bias gradient).  ws: sscg_colsum_workspace bytes. */
size_t sscg_colsum_workspace(int64_t rows, int cols);
int sscg_colsum(const void* x, int dtype, float* out, int64_t rows, int cols, float beta, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ normalisation (K3, K4, K7)
 * x is viewed as [G][L][C]: InstanceNorm2d (arch/ops.py:11: affine=False, no running stats) has G = N,
 * L = H*W; BatchNorm2d (arch/generators.py:326-337,390,417; arch/ops.py:9) has G = 1, L = N*H*W.
 * eps 1e-5, biased variance for normalisation, unbiased for running_var (torch semantics). */
size_t sscg_norm_stats_workspace(int G, int64_t L, int C);
/* mean[G][C], rstd[G][C]; if running_mean != NULL:
 * running = (1-momentum)*running + momentum*batch (running_var from the unbiased batch variance), applied once per
 * group in the order g = 0..G-1.  G > 1 with running statistics is BatchNorm2d over G batches stacked along N in one
 * launch ("grouped"): bit-identical to G successive forwards of the layer, each on its own batch. */
int sscg_norm_stats(const void* x, int dtype, int G, int64_t L, int C, float eps, float* mean, float* rstd,
                    float* running_mean, float* running_var, float momentum, void* ws, size_t ws_bytes, void* stream);
/* y = act((x-mean)*rstd*gamma + beta + residual); gamma/beta/residual nullable (gamma,beta are [C]).
 * x, residual, y share `dtype`. */
int sscg_norm_apply(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                    const void* residual, void* y, int dtype, int G, int64_t L, int C, int act, float slope, void* stream);
/* eval-mode BatchNorm: mean = running_mean, rstd = 1/sqrt(running_var + eps) */
/* The reduction pass of a normalisation layer's backward, fused into the data gradient that produces its upstream gradient
 * ("Conv -> norm -> ReLU -> Conv" chains: arch/ops.py:40-57, the Bottleneck's conv1-bn1-relu-conv2-bn2-relu-conv3 and the
 * bn3 + residual -> relu -> next block's conv1 link, arch/generators.py:345-365; autograd of model.py:472,539).
 * sscg_conv2d_dgrad_bsums is sscg_conv2d_dgrad (no bias, no activation) of the CONSUMER convolution; its result dx
 * (+ addend, below) is the gradient at the output of act(norm(nx) [+ residual]) with nx [G * L][C] (C = d->C), and its epilogue
 * also leaves, per tile row and channel, the fp64 sums of gg and gg * xhat (gg = act'(...) dx, xhat = (nx - mean) * rstd) in `sums`.
 * The ReLU / LeakyReLU mask is the sign of gamma * xhat + beta for a unit without a residual (nz NULL), and the sign of the unit's
 * forward output nz ([G * L][C]; this conv's own input) for a unit a residual joined (fp32 tensors only).
 * addend (nullable, [N][H][W][C] like dx; fp32 tensors only): dx = dgrad(dy, wt) + addend - the gradient another consumer of the
 * same tensor left (a residual block's input feeds conv1 and the shortcut): the fan-in joins in the store phase, the sums see the
 * total.  sscg_norm_bwd_from_sums (same descriptor d) then finishes that layer's backward - finalize + apply, no pass over
 * (dx, nx) for the sums; y / dres (nullable): the unit's forward output and the residual's gradient, for a unit a residual joined.
 * sscg_conv2d_dgrad_bsums_bytes returns 0 when the fusion does not apply to the geometry (strided / few-channel data gradients,
 * groups shorter than a tile): use sscg_conv2d_dgrad + sscg_norm_bwd.
 * sscg_conv2d_dgrad_add: the addend alone (no sums), wherever sscg_conv2d_dgrad_add_applies (the split family, any stride). */
size_t sscg_conv2d_dgrad_bsums_bytes(const sscg_conv_desc* d, int G, int64_t L);
int sscg_conv2d_dgrad_bsums(const sscg_conv_desc* d, const void* dy, const void* wt, void* dx, const void* nx, const void* nz,
                            const void* addend, const float* mean, const float* rstd, const float* gamma, const float* beta, int G,
                            int64_t L, int act, float slope, void* sums, size_t sums_bytes, void* ws, size_t ws_bytes, void* stream);
int sscg_conv2d_dgrad_add_applies(const sscg_conv_desc* d);
int sscg_conv2d_dgrad_add(const sscg_conv_desc* d, const void* dy, const void* wt, const void* addend, void* dx, void* ws,
                          size_t ws_bytes, void* stream);
/* flags: bit 1 = dgamma / dbeta are written (else accumulated); ws: G * C * 2 floats */
int sscg_norm_bwd_from_sums(const sscg_conv_desc* d, const void* sums, const void* dy, const void* x, const void* y, const float* mean,
                            const float* rstd, const float* gamma, const float* beta, void* dx, void* dres, float* dgamma, float* dbeta,
                            int dtype, int G, int64_t L, int C, int act, float slope, int flags, void* ws, size_t ws_bytes, void* stream);

int sscg_rstd_from_var(const float* var, float* rstd, int n, float eps, void* stream);
/* backward of norm_apply (+ of the statistics): dx always; dres (= masked dy) if non-NULL;
 * dgamma/dbeta accumulate (+=) if non-NULL.  `y` (the forward output) supplies the activation mask; with ReLU / LeakyReLU
 * and NO residual in the forward, y may be NULL: the mask is then recomputed as gamma * xhat + beta > 0 (the forward's own
 * expression; beta is only read in that case) and the kernels read one tensor less.
 * stats_grad bit 0: 0 = the statistics are constants (eval-mode BN); bit 1 (value 2): dgamma / dbeta are WRITTEN instead of
 * accumulated (the caller then needs no zero-fill). */
size_t sscg_norm_bwd_workspace(int G, int64_t L, int C);
int sscg_norm_bwd(const void* dy, const void* x, const void* y, const float* mean, const float* rstd,
                  const float* gamma, const float* beta, void* dx, void* dres, float* dgamma, float* dbeta, int dtype, int G,
                  int64_t L, int C, int act, float slope, int stats_grad, void* ws, size_t ws_bytes, void* stream);   /* dy, x, y, dx, dres share `dtype` */

/* PixelDiscriminator tail, arch/discriminators.py:72-75: norm_layer(2 ndf) -> nn.LeakyReLU(0.2) -> nn.Conv2d(2 ndf, 1, 1x1) as ONE
 * pass over the C-channel map x (the conv output the statistics were taken from): out[r] = bias + sum_c w[c] * act(norm(x)[r][c]),
 * out fp32 [G * L].  The normalised map is never materialised, in either direction: the backward recomputes it from x, forms
 * dy[r][c] = dout[r] * w[c] in registers, and produces dx (gradient at x, dtype of x), dw[C] / dbias[1] (the head's weight and bias
 * gradient) and dgamma / dbeta.  flags: bit 0 = the statistics are functions of x (training-mode normalisation), bit 1 = dgamma /
 * dbeta are written (else accumulated), bit 2 = dw / dbias are written (else accumulated).  C: a power of two in [16, 256]
 * (sscg_norm_head_applies); act: none / ReLU / LeakyReLU. */
int sscg_norm_head_applies(int C);
int sscg_norm_head_fwd(const void* x, int dtype, const float* mean, const float* rstd, const float* gamma, const float* beta,
                       const float* w, const float* bias, float* out, int G, int64_t L, int C, int act, float slope, void* stream);
size_t sscg_norm_head_bwd_workspace(int G, int64_t L, int C);
int sscg_norm_head_bwd(const float* dout, const float* w, const void* x, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, void* dx, float* dw, float* dbias, float* dgamma, float* dbeta, int dtype, int G, int64_t L,
                       int C, int act, float slope, int flags, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ pointwise / pooling / resize */
/* standalone activation (nn.ReLU / nn.LeakyReLU / nn.Tanh not adjacent to a norm) */
int sscg_act_fwd(const void* x, void* y, int dtype, int64_t n, int act, float slope, void* stream);
int sscg_act_bwd(const void* dy, const void* y, void* dx, int dtype, int64_t n, int act, float slope, void* stream);
/* y = a + b */
int sscg_add(const void* a, const void* b, void* y, int dtype, int64_t n, void* stream);
/* dst[r][c] = (c < Cs) ? src[r][c] : 0 for c < Cd, r < rows (fp32): an NHWC tensor (or a [K][R][S][C] weight) with its channels
 * padded with zeros (Cd > Cs) or cut (Cd < Cs).  No reference counterpart: the 21-channel stems (arch/generators.py:73,373 on a
 * one-hot / softmax map) run as 32-channel convolutions on the split contraction - a zero channel meets a zero weight - and the data
 * gradient's extra channels are cut again. */
int sscg_resize_channels(const float* src, float* dst, int64_t rows, int Cs, int Cd, void* stream);
/* nn.Dropout(0.5) in training mode (arch/ops.py:66): y = x * keep / (1-p); keep is derived from a
 * counter-based hash of (seed, element index), so backward can regenerate it. */
int sscg_dropout(const void* x, void* y, int dtype, int64_t n, float p, uint64_t seed, void* stream);
/* utils.GaussianNoise (utils.py:116-140; call site model.py:486-488): y = x + sigma * x * n, n ~ N(0, 1) drawn from a
 * counter-based hash of (seed, element index) through Box-Muller. */
int sscg_gauss_noise(const float* x, float* y, int64_t n, float sigma, uint64_t seed, void* stream);
/* nn.MaxPool2d(2, 2) (floor mode: P = H / 2, Q = W / 2) of torchvision's VGG16 features (utils.Vgg16, utils.py:147-164);
 * idx = window position 0..3 of the first max */
int sscg_maxpool2x2_fwd(const void* x, void* y, uint8_t* idx, int dtype, int N, int H, int W, int C, void* stream);
int sscg_maxpool2x2_bwd(const void* dy, const uint8_t* idx, void* dx, int dtype, int N, int H, int W, int C, void* stream);
/* nn.MaxPool2d(3, 2, 1, ceil_mode=True) (arch/generators.py:394); idx = window position 0..8 of the first max */
int sscg_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int dtype, int N, int H, int W, int C, int P, int Q, void* stream);
int sscg_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int dtype, int N, int H, int W, int C, int P, int Q, void* stream);
/* nn.Upsample(size, mode='bilinear', align_corners=True) (model.py:268, calls :390-392,:413-415) */
int sscg_upsample_bilinear_fwd(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, void* stream);
int sscg_upsample_bilinear_bwd(const float* dy, float* dx, int N, int H, int W, int C, int OH, int OW, void* stream);
/* nn.ReflectionPad2d as a materialised copy (only for callers that cannot fold it) */
int sscg_reflect_pad(const void* x, void* y, int dtype, int N, int H, int W, int C, int pad, void* stream);
int sscg_reflect_pad_bwd(const void* dy, void* dx, int dtype, int N, int H, int W, int C, int pad, void* stream);
/* layout plumbing at the NCHW boundary of the reference's module interface */
int sscg_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, void* stream);
int sscg_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------ class-axis ops (K10, K12) : x is [rows][C] */
/* nn.Softmax2d (model.py:273, calls :401-402,:420-421) */
int sscg_softmax_fwd(const float* x, float* y, int64_t rows, int C, void* stream);
int sscg_softmax_bwd(const float* dy, const float* y, float* dx, int64_t rows, int C, void* stream);
/* fake_gt.data.max(1)[1] -> make_one_hot (model.py:435-437,:509-511; utils.py:344-348): first max wins */
int sscg_argmax_onehot(const float* x, float* onehot, int64_t* index, int64_t rows, int C, void* stream);
/* make_one_hot(labels) (utils.py:314-350) */
int sscg_label_onehot(const int64_t* labels, float* onehot, int64_t rows, int C, void* stream);
/* runningScore._fast_hist (utils.py:363-369) of the per-epoch evaluation (model.py:555-574):
 * hist[C*t + p] += 1 for every pixel with 0 <= t < C (others, e.g. the 255 "void" label, are ignored).
 * `hist` is int64 [C][C] on the device and is accumulated into; C <= 64. */
int sscg_confusion_hist(const int64_t* label_true, const int64_t* label_pred, int64_t n, int C, int64_t* hist, void* stream);

/* ------------------------------------------------------------------ input pipeline
 * The tail of the reference's per-sample transforms, batched on the device: images travel to HBM as the uint8
 * pixels PIL decoded / resized / cropped, labels as their uint8 ids.
 * ToTensor + Normalize (data_utils/__init__.py:126-150): y = ((float)u / 255 - mean[c]) / std[c], the same two fp32
 * operations in the same order => bit-exact.  src uint8 [rows][C] (HWC pixels of a batch), dst fp32 NHWC. */
int sscg_image_u8_to_f32(const uint8_t* src, float* dst, int64_t rows, int C, const float* mean, const float* stdev, void* stream);
/* ToLabel + Relabel(255, 0) (data_utils/__init__.py:34-58) / CityscapesDataset.encode_segmap (dataloader.py:260-267) as
 * one 256-entry table: dst[i] = lut[src[i]] (int64 out, the dtype of `.long()`). */
int sscg_label_lut(const uint8_t* src, int64_t* dst, int64_t n, const int64_t* lut256, void* stream);

/* ------------------------------------------------------------------ losses (K10, K11), mean reduction
 * Each forward writes one fp32 scalar to `loss` (device).  Each backward takes the upstream gradient as
 * a device scalar `gscale` (NULL = 1) times the host factor `w`. */
size_t sscg_loss_workspace(int64_t n);
/* nn.CrossEntropyLoss (model.py:272; calls :398,:455): logits [rows][C], labels [rows].  Pixels whose label is outside
 * [0, C) are ignored (no read past the row, excluded from the mean, zero gradient) - torch's ignore_index behaviour for
 * every out-of-range id.  `valid` (nullable) receives the number of counted pixels; the backward divides by it
 * (NULL = rows).  No counted pixel at all: the loss is NaN (torch's 0 / 0), the gradient zero. */
int sscg_ce_fwd(const float* logits, const int64_t* labels, int64_t rows, int C, float* loss, float* valid, void* ws,
                size_t ws_bytes, void* stream);
int sscg_ce_bwd(const float* logits, const int64_t* labels, int64_t rows, int C, const float* gscale, float w,
                const float* valid, float* dx, void* stream);
/* The head of the segmentation generator without the resized logits in memory (ABI v12): x = low-resolution logits [N][H][W][C]
 * (C <= 64), resized to [OH][OW] by bilinear interpolation with align_corners=True (model.py:390-392, the arithmetic of
 * sscg_upsample_bilinear_fwd), then
 *   labels != NULL ([N][OH][OW]): nn.CrossEntropyLoss of the resized logits (model.py:398, :455; sscg_ce_fwd's label rules) into
 *     `loss` / `valid`, and `dlogits` [N][H][W][C] = sum over the counted pixels of d(their loss term) / dx - the gradient with
 *     respect to x up to the factor g / valid, which sscg_upsample_head_bwd applies;
 *   y_soft != NULL ([N][OH][OW][C]): softmax over C of the resized logits (model.py:401-402).
 * sscg_upsample_head_bwd: dx = adjoint of the resize applied to softmax_bwd(dy_soft, y_soft) (dy_soft NULL: that branch is unused)
 * + dlogits * g_ce / valid (dlogits NULL: no cross-entropy branch; g_ce NULL = 1).  Gather form, fixed summation order.
 * ws: sscg_upsample_head_workspace bytes (cross-entropy branch only). */
size_t sscg_upsample_head_workspace(int N, int H, int W);
int sscg_upsample_head_fwd(const float* x, const int64_t* labels, float* y_soft, float* loss, float* valid, float* dlogits, int N, int H,
                           int W, int C, int OH, int OW, void* ws, size_t ws_bytes, void* stream);
int sscg_upsample_head_bwd(const float* x, const float* dy_soft, const float* dlogits, const float* g_ce, const float* valid, float* dx,
                           int N, int H, int W, int C, int OH, int OW, void* stream);
/* nn.MSELoss against a constant target map of ones/zeros (LSGAN; model.py:445-446,452,521-528) */
int sscg_mse_const_fwd(const float* x, int64_t n, float target, float* loss, void* ws, size_t ws_bytes, void* stream);
int sscg_mse_const_bwd(const float* x, int64_t n, float target, const float* gscale, float w, float* dx, void* stream);
/* nn.MSELoss between two tensors (utils.perceptual_loss, utils.py:205-206): gradient to a, and to b when db != NULL */
int sscg_mse_fwd(const float* a, const float* b, int64_t n, float* loss, void* ws, size_t ws_bytes, void* stream);
int sscg_mse_bwd(const float* a, const float* b, int64_t n, const float* gscale, float w, float* da, float* db, void* stream);
/* nn.L1Loss (model.py:271; call :461) */
int sscg_l1_fwd(const float* a, const float* b, int64_t n, float* loss, void* ws, size_t ws_bytes, void* stream);
int sscg_l1_bwd(const float* a, const float* b, int64_t n, const float* gscale, float w, float* da, void* stream);
/* out = sum_i w[i] * (*terms[i]) for up to 16 device scalars (gen_loss / discriminator_loss, model.py:464-468,538) */
int sscg_weighted_sum(const float* const* terms, const float* w, int n, float* out, void* stream);

/* ------------------------------------------------------------------ optimiser (K14)
 * torch.optim.Adam (model.py:286-287; steps :474,:542): eps 1e-8, no weight decay, no amsgrad.
 * One launch over a flat arena; grad is multiplied by grad_scale first (1/world_size under data parallel).
 * `shadow` (nullable): an operand copy of the parameter arena rewritten in the same pass - the copy the convolutions read (the
 * fp32 arena stays the master copy): shadow_dtype SSCG_BF16 = a bfloat16 arena of n elements; SSCG_BF16X3 = the three planes
 * of the split contraction (sscg_split3), n elements apart. */
int sscg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow, int shadow_dtype, int64_t n,
                   double lr, double beta1, double beta2, double eps, int step, float grad_scale, void* stream);
int sscg_fill(float* x, int64_t n, float v, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSCG_H */
