/*
 * hsp.h -- C ABI of libhsp.so, the MI355X (gfx950) kernel library behind the
 * HierSpeech++ waveform-generation hot path.
 *
 * The reference (liuhuang31/Megatts2_HierSpeechpp) has no FFI/plugin layer: its hot
 * path is a chain of torch ops called from Python nn.Modules.  The
 * entry points below are therefore "what the reference's FFI for this path would
 * bind": one launcher per fused op, plain pointers + sizes + a hipStream_t, no torch
 * types.  Each one names the reference code it replaces.  The Python host side
 * (megatts2_hierspeechpp_amd/_lib.py) binds them with ctypes; INTEGRATION.md shows the
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *   - all tensors fp32, device pointers, layout (B, C, T) with T fastest unless stated;
 *     strides are in ELEMENTS
 *   - launchers never allocate, never synchronise, never throw; they return 0 on
 *     success, HSP_EINVAL (-1) on bad arguments, or the positive hipError_t of the launch
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream)
 *   - thread-compatible: the only global state is a per-device "dynamic LDS limit raised" flag per kernel
 *     (atomic, idempotent); the first launch of a kernel on a device must not happen inside a stream capture
 *
 * Limits a caller can reach (each is checked: the entry point returns HSP_EINVAL, it never computes garbage)
 *   - first launch outside a capture: run one eager pass of a model before capturing it into a hipGraph (the
 *     dynamic-LDS limit of a kernel is raised with hipFuncSetAttribute on its first launch per device)
 *   - hsp_conv1d_mfma_f32: stride 1; halo (K - 1) * dil + 3 <= 125 columns (plain and gated rows); one utterance of
 *     a tensor and the packed weight each below 2^31 elements
 *   - hsp_mha_f32: no sequence-length ceiling (score rows that do not fit LDS are walked in key blocks); head dim
 *     <= 128 without a relative-position window, <= 256 with one
 *   - hsp_lstm_bidir_f32: hidden size H <= 256 (one workgroup of 4 H <= 1024 threads holds the gate rows)
 */
#ifndef HSP_H_
#define HSP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libhsp.so is built with -fvisibility=hidden: exactly the functions declared between this push and the pop at the end
 * of the header are exported (`nm -D --defined-only libhsp.so` lists these names and nothing else; tests/test_host_logic.py). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define HSP_VERSION 102 /* 0.1.2: round 6 -- hsp_dftseg_pair_f32 takes the pass-through form (inv->y / inv->res); 0.1.1: round 5
                           * -- hsp_dftseg_args grew a field (prod3); hsp_cprod3_f32, hsp_cprod3_supported,
                           * hsp_dftseg_weight_spectrum_f32, hsp_dftseg_supported are new */
#define HSP_EINVAL (-1)

int hsp_version(void);
/* gfx arch string the library was compiled for ("gfx950") */
const char* hsp_arch(void);

/* ---------------------------------------------------------------- conv1d (fused) */

/* prologue applied to the conv INPUT while it is staged in LDS */
enum {
  HSP_PRO_NONE = 0,
  HSP_PRO_LRELU = 1, /* F.leaky_relu(x, slope): DBlock, hierspeechpp_speechsynthesizer.py:336 */
  HSP_PRO_ACT1D = 2, /* Activation1d(SnakeBeta): alias_free_torch/act.py:23-28 + activations.py:107-119 */
  HSP_PRO_SILU = 3   /* nn.SiLU before a Linear: modules.py:402-405 (direct kernel only) */
};

/* pointwise function applied to (acc + bias) in the epilogue */
enum {
  HSP_ACT_NONE = 0,
  HSP_ACT_TANH = 1,      /* Generator tail, hierspeechpp_speechsynthesizer.py:450 */
  HSP_ACT_GELU_TANH = 2, /* FFN_Conv, modules.py:384,400 */
  HSP_ACT_RELU = 3,      /* attentions.FFN, attentions.py:292 */
  HSP_ACT_MISH = 4,      /* StyleEncoder, styleencoder.py:9-10 */
  HSP_ACT_SILU = 5,      /* cond_block, hierspeechpp_speechsynthesizer.py:72 */
  HSP_ACT_SOFTPLUS = 6,
  HSP_ACT_GELU_ERF = 7   /* exact GELU x Phi(x) of the wav2vec2 producer (HF ACT2FN["gelu"]) */
};

/* how packed weight rows map to output channels */
enum {
  HSP_ROWS_PLAIN = 0,    /* row m -> channel m */
  HSP_ROWS_GATE_WN = 1,  /* rows come in 32-blocks (a, b): out = tanh(a)*sigmoid(b); commons.py:107-114 */
  HSP_ROWS_GATE_GLU = 2, /* out = a*sigmoid(b); styleencoder.py:26-31 */
  HSP_ROWS_SHUFFLE = 3   /* ConvTranspose1d as polyphase conv: row m -> (co = m / up, phase = m % up) */
};

enum { HSP_MASK_NONE = 0, HSP_MASK_PRE = 1, HSP_MASK_POST = 2, HSP_MASK_BOTH = 3 };

/*
 * y = epilogue( conv1d( prologue(x) ) )
 *
 * Replaces torch Conv1d / ConvTranspose1d / Linear call sites together with the
 * elementwise work around them: hierspeechpp_speechsynthesizer.py:195-200 (Posterior
 * SF encoder), :292-307 (SourceNetwork), :331-339 (DBlock), :380-384 (AMPBlock1),
 * :430-450 (Generator); modules.py:156-175 (WN), :383-386 (FFN_Conv), :409-410 (DiT
 * residuals), :461,473,486 (coupling); styleencoder.py:26-31,66-78.
 *
 * Weights are pre-folded (weight-norm) and packed by the host as w[K][Cin][w_ld] (w_ld =
 * packed row count, a multiple of 4; a launch computes rows [0, M) of it) in the row
 * order `rows` names.
 *
 * conv:     acc[m, t] = sum_{j<K} sum_{ci<Cin} w[j][ci][m] * xin[ci, t*stride + j*dil - pad]
 *           xin = prologue(x) inside [0, Lin), 0 outside; x is read at
 *           x + b*x_bs + ci*x_cs + i*x_ts
 * epilogue: v = acc + bias[row] + cbias[b*cbias_bs + row]          (row = channel index
 *               of the un-packed layer; GATE modes add both halves before gating)
 *           v = act(v)                       | gate(v_a, v_b) in GATE modes
 *           v *= mask[b*mask_bs + t]         if mask_mode & PRE
 *           v *= cscale[b*cscale_bs + co]    if cscale
 *           v *= scale
 *           v += res[b*res_bs + co*res_cs + t]   if res
 *           v *= mask[...]                   if mask_mode & POST
 *           v += y[...]                      if accumulate
 *           y[b*y_bs + co*y_cs + t] = v * post_scale
 *           (SHUFFLE: co = m / up, t -> up*t + m % up - shuf_pad, bounds-checked to Lout)
 */
typedef struct hsp_conv1d_args {
  /* input */
  const float* x;
  int64_t x_bs, x_cs, x_ts;
  int32_t B, Cin, Lin;
  /* packed weights */
  const float* w;
  int32_t K, M, dil, pad, stride;
  const float* zeros; /* >= 16 B of zeros, 16-B aligned (MFMA path: source of out-of-range DMA lanes) */
  int32_t w_ld; /* row count of the packed matrix w points into (>= M): lets a launch
                   use a row sub-range [r0, r0+M) by passing w + r0 (r0 % 4 == 0) */
  /* output */
  float* y;
  int64_t y_bs, y_cs;
  int32_t Cout, Lout; /* logical output tensor [B, Cout, Lout] */
  int32_t ncols;      /* conv columns to compute (== Lout except SHUFFLE) */
  /* prologue */
  int32_t prologue;
  float slope;
  const float* alpha_exp; /* [Cin] exp(alpha)              (ACT1D) */
  const float* beta_inv;  /* [Cin] 1/(exp(beta)+1e-9)      (ACT1D) */
  const float* filt;      /* [24]  up filter then down filter (ACT1D) */
  /* epilogue */
  int32_t rows, gate_half, up, shuf_pad;
  const float* bias;
  const float* cbias;
  int64_t cbias_bs;
  int32_t act;
  const float* mask;
  int64_t mask_bs;
  int32_t mask_mode;
  const float* cscale;
  int64_t cscale_bs;
  float scale;
  const float* res;
  int64_t res_bs, res_cs;
  int32_t accumulate;
  float post_scale;
  int32_t debug; /* must be 0: libhsp.so returns HSP_EINVAL otherwise.  The kernel tuning switches this word
                    selects (skip staging / MFMAs / epilogue, force a tile shape) are compiled only into the
                    separate libhsp_tune.so (make tune, -DHSP_TUNING), which tools/ load through HSP_LIB. */
  /* Fused input LayerNorm (1x1 token GEMMs only; any other shape is refused with HSP_EINVAL):
   * y = W LN(x) + b with LN over the Cin channels of every column.  The caller packs
   * w = W diag(gamma), bias = W beta + b and ln_c1[row] = sum_ci w[ci][row]; the kernel takes the
   * column statistics from the staged input tile and applies
   *   v = rstd[t] * (acc[m, t] - mean[t] * ln_c1[row]) + bias[row]      before the activation.
   * nn.LayerNorm -> nn.Linear pairs of ttv_v1/transformer_mega.py:125-131.  NULL = off. */
  const float* ln_c1;
  float ln_eps;
  /* Second output of a 1x1 token GEMM (token-GEMM path only; any other shape is refused with HSP_EINVAL).
   * With split_row > 0 (a multiple of 64) the rows [split_row, Cout) are written to
   *   y2[b][m - split_row][t] = (accumulate2 ? y2 : 0) + (acc[m, t] + bias[m]) [* mask per mask_mode2]
   * while the rows [0, split_row) keep y / res / accumulate / mask_mode (indexed by m).  One GEMM then serves the
   * two halves of modules.WN's res_skip_layers (modules.py:166-174): x = (x + rs[:H]) * mask and out += rs[H:]. */
  int32_t split_row;
  int32_t accumulate2;
  int32_t mask_mode2;
  float* y2;
  int64_t y2_bs, y2_cs;
  /* Column stride of `res` in elements (0 or 1: unit).  Only the register-path token GEMM reads a residual at another
   * stride -- the last layer of the PLM loop adds the last position of every utterance, columns T-1, 2T-1, ... of the
   * layer input, to a [D, B] output without a gather launch in between (ttv_v1/transformer_mega.py:118-131 on the
   * last position only); any other kernel / shape refuses res_ts > 1 with HSP_EINVAL. */
  int64_t res_ts;
  /* Weight stride between the B "utterances" of the batch (elements; 0 = one weight matrix for all of them, as every
   * layer of the reference has).  Non-zero only for the frequency-domain form of a long dilated conv (round 4,
   * hsp_dftseg_*_f32 below), where the batch index is the frequency BIN and every bin has its own [Cin][M] matrix.
   * The implicit-GEMM conv kernel only (the token GEMMs refuse it). */
  int64_t w_bs;
  /* Modulated input LayerNorm (round 6; with ln_c1, on the block token GEMM only -- any other launch that carries one of
   * these fields is refused with HSP_EINVAL): the adaLN form of a DiT block's first half,
   *   qkv = W ((LN(x) * mask) * (1 + scale_b) + shift_b) + bias        (modules.py:346-347,406-409: norm1 has no affine,
   *                                                                     scale / shift differ per UTTERANCE b)
   * as ONE launch on the un-normalised x:  v = rstd[t] mask[b, t] (acc[m, t] - mean[t] c1_b[m]) + bias_b[m]  with
   *   acc = W diag(1 + scale_b) x          ln_scale[b * ln_scale_bs + ci] = scale_b[ci]: the kernel multiplies the staged
   *                                        input fragments by (1 + scale) on their way into the MFMA
   *   c1_b[m] = sum_ci W[m][ci] (1 + scale_b[ci])      = ln_c1[b * ln_c1_bs + m]        (ln_c1_bs = 0: one vector for all b)
   *   bias_b[m] = sum_ci W[m][ci] shift_b[ci] + bias[m] = cbias[b * cbias_bs + m]       (bias NULL)
   *   ln_mask[b * ln_mask_bs + t]: the 0 / 1 column mask applied to the normalised input (NULL = none).
   * c1_b and bias_b are linear in the conditioning vector: the caller gets them as extra rows of the one GEMM that
   * produces scale_b / shift_b (hip_layers / modules.DiTConVBlock).  Cin <= 1024.  NULL ln_scale = plain ln_c1 form. */
  const float* ln_scale;
  int64_t ln_scale_bs;
  int64_t ln_c1_bs;
  const float* ln_mask;
  int64_t ln_mask_bs;
} hsp_conv1d_args;

/* MFMA (v_mfma_f32_32x32x2_f32, exact fp32) path; stride must be 1, M % 4 == 0.  Behind this entry point: the
 * implicit-GEMM conv kernel (any K / dilation / prologue / row mode) and, for 1x1 convs over token columns (or with
 * ln_c1 / split_row), three token GEMMs -- the register-path and the LDS-DMA kernel (latency-oriented, up to a few
 * hundred thousand outputs) and the block token GEMM (64 x 64 / 128 x 128 tiles, from ~96 tiles of 64 x 64 upward);
 * hsp_conv1d_mfma_plan tells which one a launch takes. */
int hsp_conv1d_mfma_f32(const hsp_conv1d_args* a, void* stream);
/* VALU path: any shape (Cin = 1, Cout = 1, stride > 1, L = 1 "Linear" cases);
 * rows must be PLAIN; prologue NONE/LRELU/SILU. */
int hsp_conv1d_direct_f32(const hsp_conv1d_args* a, void* stream);
/* which kernel / tile configuration hsp_conv1d_mfma_f32 would pick: writes BM, BN, KC, LDS bytes
 * (KC > 0: the conv kernel's chunk depth; KC = 0: the LDS-DMA token GEMM; KC = -1: the register-path token GEMM;
 * KC = -2: the block token GEMM of hsp_bgemm.hip, 64 x 64 or 128 x 128 tiles) */
int hsp_conv1d_mfma_plan(const hsp_conv1d_args* a, int32_t out4[4]);

/* --------------------------------------- feature producer of inference_vc.py */
/* y[i] = act(x[i]) (HSP_ACT_*): the GELU that follows the channel LayerNorm of every wav2vec2 feature-encoder
 * layer (HF Wav2Vec2LayerNormConvLayer.forward: conv -> LayerNorm -> GELU). */
int hsp_act_f32(const float* x, float* y, int64_t n, int32_t act, void* stream);
/* y[b, t] = x[b, reflect(t - pad)], t < L + 2 pad : F.pad(audio, (40, 40), "reflect") at inference_vc.py:85.
 * x rows of stride x_bs, y contiguous [B, L + 2 pad]; pad < L. */
int hsp_reflect_pad_f32(const float* x, int64_t x_bs, float* y, int32_t B, int32_t L, int32_t pad, void* stream);
/* F0 conversion of inference_vc.py:80-81,104-105: with V = {f0_src != 0}, W = {f0_trg != 0} (population statistics,
 * numpy's default ddof = 0): out = log(max((f0_src - mean_V) / std_V * std_W + mean_W, 0) + 1) on V and log(1) = 0
 * elsewhere.  One utterance per call (the statistics are per utterance): f0_src [n_src], f0_trg [n_trg]. */
int hsp_f0_convert_f32(const float* f0_src, int32_t n_src, const float* f0_trg, int32_t n_trg, float* out, void* stream);

/* ----------------------------------------------- prompt denoiser (MP-SENet) */
/* The parts of denoiser/ that are neither convolutions nor GEMMs (those run on hsp_conv1d_mfma_f32 /
 * hsp_conv1d_direct_f32 / hsp_mha_f32 / hsp_layernorm_mod_f32).  One utterance per call, as the reference
 * (denoiser/infer.py:3-10 takes a 1-D prompt). */
/* out[0] = sum_i x[i]^2 (double accumulation): the norm factor sqrt(len / sum) of denoiser/infer.py:4. */
int hsp_sum_sq_f32(const float* x, int64_t n, float* out, void* stream);
/* mag[f][t] = |z|^compress, pha[f][t] = angle(z) for z = spec[f][t] + i spec[n_freqs + f][t] (row pitch s_ld):
 * mag_pha_stft of denoiser/infer.py:12-24 after the DFT product; the imaginary part of the DC and Nyquist rows is
 * taken as +0, as a real FFT returns it.  mag, pha contiguous [n_freqs, T]. */
int hsp_mag_pha_f32(const float* spec, int64_t s_ld, float* mag, float* pha, int32_t n_freqs, int32_t T, float compress,
                    void* stream);
/* In place: nn.InstanceNorm2d(C, affine=True) then nn.PReLU(C) over the N contiguous values of each of the C
 * channel planes at x + c * x_cs (biased variance, double accumulation as torch's CPU path):
 * denoiser/generator.py:22-23,40-41,47-48,65-66,83-84. */
int hsp_instnorm_prelu_f32(float* x, int64_t x_cs, int32_t C, int64_t N, const float* gamma, const float* beta,
                           const float* slope, float eps, void* stream);
/* y = SiLU(BatchNorm1d_eval(depthwise Conv1d(x))) over contiguous [B, C, N]; w [C, K] (K odd, padding K / 2),
 * batch-norm as y = x alpha + (bias - mean alpha), alpha = weight / sqrt(var + eps): denoiser/conformer.py:34-37. */
int hsp_dwconv_bn_silu_f32(const float* x, const float* w, const float* bias, const float* bn_weight,
                           const float* bn_bias, const float* bn_mean, const float* bn_var, float bn_eps, float* y,
                           int32_t B, int32_t C, int32_t N, int32_t K, void* stream);
/* out[t][f] = mag[t][f] * beta * sigmoid(slope[f] * m[t][f]): LearnableSigmoid_2d (denoiser/utils.py:44-53) and the
 * mask product of generator.py:140; all [T, F] contiguous. */
int hsp_lsigmoid_mul_f32(const float* m, const float* slope, float beta, const float* mag, float* out, int32_t T,
                         int32_t F, void* stream);
/* out[i] = atan2(y[i], x[i]): PhaseDecoder.forward (denoiser/generator.py:96). */
int hsp_atan2_f32(const float* y, const float* x, float* out, int64_t n, void* stream);
/* re[f][t] = mag[f][t]^power cos(pha[f][t]), im likewise with sin (row pitches re_ld / im_ld): denoised_com of
 * generator.py:142-143 (power 1) and the decompression of mag_pha_istft (infer.py:26-29, power 1 / compress). */
int hsp_polar_f32(const float* mag, const float* pha, float power, float* re, int64_t re_ld, float* im, int64_t im_ld,
                  int32_t F, int32_t T, void* stream);
/* torch.istft(center=True) after the inverse DFT: out[n] = scale * sum_t frames[k][t] w[k] / sum_t w[k]^2 with
 * k = n + n_fft / 2 - t hop in [0, n_fft), n < hop (T - 1); frames [n_fft, f_ld] (denoiser/infer.py:30-31 and the
 * division by the norm factor at :9). */
int hsp_istft_ola_f32(const float* frames, int64_t f_ld, const float* window, float* out, int32_t n_fft, int32_t hop,
                      int32_t T, float scale, void* stream);

/* ------------------------------------------------------- anti-aliased activation */
/* y = DownSample2x(SnakeBeta(UpSample2x(x))): alias_free_torch/act.py:23-28,
 * resample.py:25-33,47-49, filter.py:86-95, activations.py:107-119.  x, y contiguous
 * [B, C, L]; filt = 12 up taps then 12 down taps. */
int hsp_act1d_snakebeta_f32(const float* x, float* y, int32_t B, int32_t C, int32_t L,
                            const float* alpha_exp, const float* beta_inv, const float* filt,
                            void* stream);
/* per-channel constants of SnakeBeta(alpha_logscale=True): activations.py:113-117 */
int hsp_snake_consts_f32(const float* alpha_log, const float* beta_log, float* alpha_exp,
                         float* beta_inv, int32_t C, void* stream);

/* ------------------------------------------------------------------ weight prep */
/* w[r, :] = g[r] * v[r, :] / ||v[r, :]||_2 : torch.nn.utils.weight_norm (dim 0), recomputed
 * on every forward by the reference; folded once here. */
int hsp_fold_weight_norm_f32(const float* v, const float* g, float* w, int32_t rows, int32_t cols,
                             void* stream);
/* dst[i] = map[i] >= 0 ? src[map[i]] : 0   (weight re-layout with a host-built index map) */
int hsp_gather_f32(const float* src, const int32_t* map, float* dst, int64_t n, void* stream);

/* ------------------------------------------------------------------- small ops */
/* mask[b, t] = t < length[b] : commons.sequence_mask commons.py:128-132 (as float) */
int hsp_sequence_mask_f32(const int64_t* length, float* mask, int32_t B, int32_t T, void* stream);
/* y[b, c, t] = x[b, C-1-c, t] : modules.Flip modules.py:270-277 */
int hsp_flip_channels_f32(const float* x, float* y, int32_t B, int32_t C, int32_t T, void* stream);
/* z = (m + noise * exp(logs) * noise_scale) * mask; stats = [B, 2C, T] (m then logs):
 * hierspeechpp_speechsynthesizer.py:201-202, 687 */
int hsp_sample_prior_f32(const float* stats, const float* noise, const float* mask, float* z,
                         int32_t B, int32_t C, int32_t T, float noise_scale, void* stream);
/* LayerNorm over C (no affine) then optional mask, then x*(1+scale)+shift with per-(b,c)
 * scale/shift: modules.py:346-347, 396, 398, 409-410.  gamma/beta (per-channel affine,
 * modules.py:19-31) optional. */
int hsp_layernorm_mod_f32(const float* x, float* y, int32_t B, int32_t C, int32_t T, float eps,
                          const float* mask, const float* shift, const float* scale, int64_t mod_bs,
                          const float* gamma, const float* beta, void* stream);
/* softmax(q^T k * qk_scale [masked]) v per (b, head); q,k,v,o are channel-major
 * [B, H*D, T] views (strides in elements).  mask, if given, is the key/query validity
 * [B, T]: scores where mask_q*mask_k == 0 are set to -1e4 (attentions.py:174-175).
 * Without mask: timm 0.6.13 Attention (modules.py:409).  Optional relative-position
 * window (emb_rel_k/v [2w+1, D]): attentions.py:165-170,183-186.
 * No length ceiling: score rows that do not fit the CU's LDS are walked in key blocks with an online softmax
 * (head dim <= 128 without a window, <= 256 with one). */
typedef struct hsp_mha_args {
  const float *q, *k, *v;
  float* o;
  int64_t q_bs, k_bs, v_bs, o_bs; /* batch strides; channel stride is Tq / Tk (contiguous rows) */
  int32_t B, H, D, Tq, Tk;
  float qk_scale;
  const float* mask_q;
  const float* mask_k; /* [B, Tq], [B, Tk] or NULL */
  const float* rel_k;
  const float* rel_v;
  int32_t window;                 /* w > 0 with rel_k / rel_v.  Negative = test hook: -(w + 1) forces the key-streaming
                                     kernels (which otherwise serve only the lengths the whole-row kernels cannot hold) */
  int64_t q_cs, k_cs, v_cs, o_cs; /* channel strides; 0 = contiguous rows (Tq / Tk).  A larger stride lets the
                                     utterances of a batch sit side by side on the column axis of one
                                     [C][B*T] matrix (batch stride T): the layout of the PLM loop */
  const float* mask_dense;        /* optional [B][Tq][Tk] (batch stride mask_dense_bs >= Tq * Tk, shared by the heads):
                                     the reference's general ``attn_mask`` (attentions.py:147-155,174-175): scores where
                                     it is 0 become -1e4.  NULL = none.  May be combined with mask_q / mask_k. */
  int64_t mask_dense_bs;
} hsp_mha_args;
int hsp_mha_f32(const hsp_mha_args* a, void* stream);
/* Self-attention over ALL heads + the output projection + its epilogue in ONE launch (csrc/hsp_mhaproj.hip):
 *   o[b, h D + d, i] = sum_j softmax_j(qk_scale q[b, h D + :, i] . k[b, h D + :, j]) v[b, h D + d, j]
 *   y[b, m, i] = ((sum_c wt[m, c] o[b, c, i] + bias[m]) * mask[b, i]) * cscale[b, m] + res[b, m, i]
 * = scaled_dot_product_attention + out_proj + the residual add of the Mega-TTS2 PLM layer
 * (ttv_v1/transformer_mega.py:63-87,121-123: 4 heads x 69) and timm Attention's softmax(q k^T) v + proj followed by
 * `x + gate_msa * attn(.)` of a DiT block (modules.py:397,409: 2 heads x 96).  Any Tk >= 4 (up to 2^20): rows of up to 256
 * keys take the one-pass form, longer ones stream the keys in groups with an online softmax (tested to 1 000 keys).  NO
 * masks inside the softmax -- neither key / query masks nor a dense or causal attn_mask: a caller that needs one uses
 * hsp_mha_f32 (mask_q / mask_k / mask_dense) and the projection as two launches.
 * q / k / v: element (b, c, t) at base + b * bs + c * cs + t (channel-major; a batch may sit side by side on the columns
 * of one [C][B * T] matrix: bs = T, cs = row pitch).  wt = the nn.Linear / 1x1-conv weight AS STORED, [M][H D] row-major
 * with row pitch wt_ld (M == H D).  y / res: element (b, m, i) at base + b * bs + m * cs + i * ts, so that the
 * "last position of every utterance" form of the PLM's final layer (Tq = 1, q pointing at column T - 1) writes a
 * [M][B] matrix and reads its residual in place.  mask [B][Tq], cscale [B][M], bias [M], res: optional (NULL).
 * hsp_mha_proj_supported: 1 when a kernel exists for (H, D, M, Tk) -- otherwise the caller issues hsp_mha_f32 and
 * the projection as two launches (hsp_mha_proj_f32 returns HSP_EINVAL). */
typedef struct hsp_mha_proj_args {
  const float *q, *k, *v;
  int64_t q_bs, q_cs, k_bs, k_cs, v_bs, v_cs;
  int32_t B, H, D, Tq, Tk;
  float qk_scale;
  const float* wt;
  int32_t M, wt_ld;
  const float* bias;
  const float* mask;
  int64_t mask_bs;
  const float* cscale;
  int64_t cscale_bs;
  const float* res;
  int64_t res_bs, res_cs, res_ts;
  float* y;
  int64_t y_bs, y_cs, y_ts;
  int32_t debug; /* must be 0 (tuning build only, as hsp_conv1d_args.debug) */
} hsp_mha_proj_args;
int hsp_mha_proj_f32(const hsp_mha_proj_args* a, void* stream);
int hsp_mha_proj_supported(int32_t H, int32_t D, int32_t M, int32_t Tk);
/* out[b, c] = sum_t x[b, c, t] / sum_t mask[b, t] : styleencoder.py:83-91 */
int hsp_masked_mean_f32(const float* x, const float* mask, float* out, int32_t B, int32_t C,
                        int32_t T, void* stream);
/* y[b, c, t] = x[b, c, t] * mask[b, t] : the `x * x_mask` steps (modules.py:407) */
int hsp_mask_mul_f32(const float* x, const float* mask, float* y, int32_t B, int32_t C, int32_t T, void* stream);
/* y = F.interpolate(x, Lout, mode='linear') along T (align_corners=False), fp32 index arithmetic
 * bit-compatible with torch-CPU: speechsr48k/speechsr.py:96 */
int hsp_linear_interp_f32(const float* x, float* y, int32_t B, int32_t C, int32_t Lin, int32_t Lout, void* stream);
/* y = a*x + b*z elementwise (style interpolation, hierspeechpp_speechsynthesizer.py:682) */
int hsp_axpby_f32(const float* x, const float* z, float* y, float a, float b, int64_t n, void* stream);


/* -------------------------------------------------- Mega-TTS2 PLM loop */
/* x[b, c, j] = cat(tc[b, :, j], emb[codes[b, j]])[c] + alpha[0] * pe_t[c, j] for j < n :
 * the body of Megatts2PLM1.infer up to pos_emb (ttv_v1/t2w2v_transformer.py:710-713) with
 * SinePositionalEmbedding.forward (:510-514).  tc is channel-major with strides (tc_bs, tc_cs, 1);
 * codes is int64 [B, >= n] with row stride codes_bs and holds the go token at column 0;
 * pe_t is the sinusoid table transposed to [Dtc + Demb][P]; x has strides (x_bs, x_cs, 1):
 * (n, B*n) lays the batch side by side on the columns of one [D][B*n] matrix; with x_bs == n and
 * x_cs > B*n the columns [B*n, x_cs) of every row are zeroed (row padding to a multiple of 4). */
int hsp_plm_embed_f32(const float* tc, int64_t tc_bs, int64_t tc_cs, int32_t Dtc, const int64_t* codes,
                      int64_t codes_bs, const float* emb, int32_t Demb, int32_t n_emb, const float* pe_t,
                      int32_t P, const float* alpha, float* x, int64_t x_bs, int64_t x_cs, int32_t B, int32_t n,
                      void* stream);
/* The same for step n >= 2 of the greedy loop with the choice of the PREVIOUS step folded in (one launch per step less):
 * first codes[b, n - 1] = argmax_c logits[b * l_bs + c * l_cs] (first maximal index on ties, as hsp_argmax_f32; the
 * value is also stored to `codes`), then x as above.  `logits` = the n_logits scores of step n - 1
 * (ttv_v1/t2w2v_transformer.py:716-717 followed by :710-713 of the next iteration).
 * n == 1 (round 6) is the ONE-POSITION form of the loop's layer-0 cache: every per-position operand (tc, codes, pe_t, x)
 * points at position t and only that position is chosen and embedded -- a caller that passes n == 1 with un-shifted
 * pointers overwrites its go token. */
int hsp_plm_embed_step_f32(const float* tc, int64_t tc_bs, int64_t tc_cs, int32_t Dtc, int64_t* codes,
                           int64_t codes_bs, const float* emb, int32_t Demb, int32_t n_emb, const float* pe_t,
                           int32_t P, const float* alpha, float* x, int64_t x_bs, int64_t x_cs, int32_t B, int32_t n,
                           const float* logits, int64_t l_bs, int64_t l_cs, int32_t n_logits, void* stream);
/* out[b * out_bs] = argmax_c logits[b * l_bs + c * l_cs], first maximal index on ties :
 * logits.argmax(dim=-1) of the greedy loop (ttv_v1/t2w2v_transformer.py:716-717) */
int hsp_argmax_f32(const float* logits, int64_t l_bs, int64_t l_cs, int32_t B, int32_t N, int64_t* out,
                   int64_t out_bs, void* stream);
/* y[b, c, t] (contiguous) = x[b * s_bs + c * s_cs + t * s_ts] : strided gather, e.g. the last
 * position of every utterance (`[:, -1:, :]`, ttv_v1/t2w2v_transformer.py:716) */
int hsp_copy_strided_f32(const float* x, int64_t s_bs, int64_t s_cs, int64_t s_ts, float* y, int32_t B,
                         int32_t C, int32_t T, void* stream);


/* ---------------------------------------------- t2w2v front-end */
/* out[b, c, t] = sum_k tab_k[id_k[b, t], c] * scale over up to three tables (tab1 / tab2 may be
 * NULL), channel-major output with strides (o_bs, o_cs, 1) : TextEncoder.forward
 * (ttv_v1/t2w2v_transformer.py:127-131); one table with scale 1 = the RVQ codebook lookup of
 * quantizer.decode (ttv_v1/core_vq.py:188-190,380-386). */
int hsp_embedding_sum_f32(const int64_t* id0, const int64_t* id1, const int64_t* id2, const float* tab0,
                          const float* tab1, const float* tab2, int32_t n0, int32_t n1, int32_t n2, float scale,
                          float* out, int64_t o_bs, int64_t o_cs, int32_t B, int32_t C, int32_t T, void* stream);
/* One bidirectional torch.nn.LSTM layer (gate order i, f, g, o; zero initial state), the recurrence
 * only: xproj[b][dir][4H][N] = x W_ih^T + b_ih (a 1x1 GEMM), whh_t[dir][H][4H] = W_hh^T,
 * bhh[dir][4H]; utterance b runs lengths[b] steps (the reverse direction from its own last step, as a
 * packed sequence does) and writes zeros after; out[b, dir*H + j, t] with strides (o_bs, o_cs, 1).
 * nn.LSTM of DurationPredictor (ttv_v1/vits_models.py:101,125) and RangePredictor (ttv_v1/Gaussian.py:100-110). */
/* Limit: H <= 256 (4 H gate rows = the threads of one workgroup); larger returns HSP_EINVAL. */
int hsp_lstm_bidir_f32(const float* xproj, int64_t xp_bs, const float* whh_t, const float* bhh,
                       const int64_t* lengths, float* out, int64_t o_bs, int64_t o_cs, int32_t B, int32_t H,
                       int32_t N, void* stream);
/* dur[b, n] = n < lengths[b] ? ceil(exp(logw[b, n]) * length_scale) : 0, frames[b] = sum_n dur[b, n]
 * (ttv_v1/t2w2v_transformer.py:955-957,972-975).  logw == NULL keeps caller-supplied durations and only
 * zeroes the padding / sums. */
int hsp_duration_f32(const float* logw, int64_t lw_bs, const int64_t* lengths, float length_scale, float* dur,
                     int64_t d_bs, float* frames, int32_t B, int32_t N, void* stream);
/* y[b, c, t] = max_{j < k} x[b, c, k t + j], t < L / k : nn.MaxPool1d(kernel_size = 8, stride = 8) of the legacy
 * prosody path (ttv_v1/t2w2v_transformer.py:795,1046).  y contiguous [B, C, L / k]. */
int hsp_maxpool1d_f32(const float* x, int64_t x_bs, int64_t x_cs, float* y, int32_t B, int32_t C, int32_t L,
                      int32_t k, void* stream);
/* codes[b, rep t + r] = argmax_e -(|x_t|^2 - 2 x_t . embed[e] + |embed[e]|^2) for r < rep and rep t + r < Tout:
 * EuclideanCodebook.quantize (ttv_v1/core_vq.py:175-183; first maximum on ties) followed by the "repeat every
 * pooled frame `stride` times, cut to the mel length" of ttv_v1/t2w2v_transformer.py:1051-1052.
 * x [B, D, T] with strides (x_bs, x_cs, 1), embed [bins, D], codes int64 rows of stride c_bs. */
int hsp_vq_nearest_f32(const float* x, int64_t x_bs, int64_t x_cs, const float* embed, int64_t* codes, int64_t c_bs,
                       int32_t B, int32_t D, int32_t T, int32_t bins, int32_t rep, int32_t Tout, void* stream);
/* GaussianUpsampling.forward (ttv_v1/Gaussian.py:35-69) with the range clamp
 * min(range, 2 dur), max(., 1e-5) of ttv_v1/t2w2v_transformer.py:961-963: x [B, C, N] (strides x_bs, x_cs, 1)
 * -> out [B, C, T] contiguous, frames t >= frames[b] zero. */
int hsp_gaussian_upsample_f32(const float* x, int64_t x_bs, int64_t x_cs, const float* dur, int64_t d_bs,
                              const float* rng, int64_t r_bs, const int64_t* lengths, const float* frames,
                              float* out, int32_t B, int32_t C, int32_t N, int32_t T, void* stream);
/* y[b, c, t] (contiguous) = x[b, c, t] + cb[b, c] : `x + self.cond(g)` with a per-utterance vector
 * (ttv_v1/vits_models.py:119, ttv_v1/t2w2v_transformer.py:222) */
int hsp_add_cbias_f32(const float* x, int64_t x_bs, int64_t x_cs, const float* cb, int64_t cb_bs, float* y,
                      int32_t B, int32_t C, int32_t T, void* stream);

/* ------------------------------------------------ inference_plm.py:tts post-steps (row A19) */
/* y[i] = x[i] < thr ? 0 : x[i] : pitch clipping `pitch[pitch < log(55)] = 0` (inference_plm.py:166) */
int hsp_zero_below_f32(const float* x, float thr, float* y, int64_t n, void* stream);
/* out[b, i] = (int16) (x[b, i] / max_j |x[b, j]| * 32767 * gain) over the first lengths[b] samples
 * (NULL = all n), zeros after : `audio / max(abs(audio)) * 32767.0 * 0.999` then numpy
 * astype('int16') (inference_plm.py:183-190) */
int hsp_peak_int16(const float* x, int64_t x_bs, const int64_t* lengths, float gain, int16_t* out, int64_t o_bs,
                   int32_t B, int64_t n, void* stream);

/* ------------------------------------------------ prompt front-end */
/* torchaudio MelSpectrogram as wrapped by MelSpectrogramFixed (Mels_preprocess.py:8-18; built with the
 * kwargs of inference_plm.py:204-213, applied at :134,150).  Three steps: these two kernels around one
 * hsp_conv1d_mfma_f32 launch (K = 1, Cin = n_fft) against the host-built DFT matrix.
 * frames[b][n][t] = window[n] * x[b][reflect(t * hop + n - n_fft / 2)], t < T = 1 + L / hop
 *   (torch.stft center=True, pad_mode="reflect": index i < 0 -> -i, i >= L -> 2 (L - 1) - i; needs
 *   L > n_fft / 2); x [B, L] contiguous, frames [B, n_fft, f_ld] with row pitch f_ld >= T. */
int hsp_stft_frames_f32(const float* x, const float* window, float* frames, int32_t B, int32_t L, int32_t n_fft,
                        int32_t hop, int32_t T, int32_t f_ld, void* stream);
/* out[b][m][t] = log(sum_f fb[f][m] * (re[b][f][t]^2 + im[b][f][t]^2) + eps), t < T_out :
 *   Spectrogram(power=2) -> MelScale -> log(. + 0.001)[..., :-1].  spec[b] holds rows 0 .. n_freqs-1 (real)
 *   and n_freqs .. 2 n_freqs - 1 (imaginary) with row pitch s_ld and batch stride s_bs; fb [n_freqs, n_mels]
 *   row-major; filter m is non-zero only on bins [f_lo[m], f_hi[m]); out [B, n_mels, T_out] contiguous. */
int hsp_power_mel_log_f32(const float* spec, int64_t s_bs, int32_t s_ld, const float* fb, const int32_t* f_lo,
                          const int32_t* f_hi, float* out, int32_t B, int32_t n_freqs, int32_t n_mels,
                          int32_t T_out, float eps, void* stream);

/* ------------------------------------------------ frequency-domain form of a long dilated Conv1d (round 4)
 * The same-length, stride-1 Conv1d(C -> C, k taps, dilation d, zero padding `pad`) of an AMP block
 * (hierspeechpp_speechsynthesizer.py:340-392) as overlap-save with a 128-point real DFT (csrc/hsp_dftseg.hip):
 *   hsp_dftseg_fwd_f32   x [B][C][L] -> xf [64 bins][2 C][Np]: row part * C + c of bin j holds Re (part 0) / Im (part 1)
 *                        of bin j of channel c (bin 0: DC in part 0, Nyquist in part 1); column n = (b * d + p) * nseg + s
 *                        is segment s of phase p (the samples xpad[p + d i]) of utterance b;
 *                        nseg = ceil(ceil(L / d) / (129 - k)), Np >= B d nseg (a multiple of 4)
 *   the channel product  ONE launch over the 64 bins, either of
 *     hsp_cprod3_f32       (round 5, below; the forward transform then runs with prod3 = 1) three real C x C products per bin, or
 *     hsp_conv1d_mfma_f32  B = 64 (the bins), Cin = M = 2 C, K = 1, Lin = Np, w_bs = 4 C^2 -- per bin the real block matrix
 *                          [[Wr, Wi], [-Wi, Wr]] of conj(rfft(w, 128)) (bin 0: [[W_dc, 0], [0, W_nyquist]]); prod3 = 0
 *   hsp_dftseg_inv_f32   yf [64][2 C][Np] -> y [B][C][L] = ((corr + bias[c] + res) [+ y]) * post_scale
 * `dft` = the transform's constant table in device memory (HSP_DFTSEG_TABLE_FLOATS floats; hsp_dftseg_tables_f32 fills
 * the forward and the inverse one into host buffers): the 64 x 64 matrix of the radix-2 half-length real transform in
 * the row order the kernel's accumulator layout wants, then (cos, sin)(2 pi k / 128), k < 32.  fp32 arithmetic on the
 * fp32 MFMA; against float64 as close as the direct fp32 sum (tools/fft_conv_err.py). */
#define HSP_DFTSEG_TABLE_FLOATS 4160
typedef struct hsp_dftseg_args {
  const float* x;   /* forward: input [B][C][L], unit time stride */
  int64_t x_bs, x_cs;
  float* y;         /* inverse: output */
  int64_t y_bs, y_cs;
  int32_t B, C, L;
  int32_t k, dil, pad, nseg, Np;
  float* xf;        /* forward: written; inverse: read */
  int64_t xf_bs;    /* plane (bin) stride >= 2 C Np */
  const float* dft;
  const float* bias; /* inverse epilogue: optional */
  const float* res;
  int64_t res_bs, res_cs;
  int32_t accumulate;
  float post_scale;
  /* forward, optional: the anti-aliased SnakeBeta in front of the conv (Activation1d; the arguments of
   * hsp_act1d_snakebeta_f32) applied while the input is staged -- the transform of act(x) without act(x) ever being
   * written.  NULL act_alpha_exp = none.  Needs 16-B addressable rows (L, x_bs, x_cs multiples of 4, x 16-B aligned). */
  const float* act_alpha_exp;
  const float* act_beta_inv;
  const float* act_filt;
  /* forward (and the forward half of the pair launch): non-zero = the spectrum feeds hsp_cprod3_f32, whose three-product
   * form cannot keep the two real bins of slot 0 apart by its matrices alone: slot 0 then holds (-E0, -O0) -- the 64-point
   * transforms of the even / odd samples at bin 0 -- instead of (DC, Nyquist) = (E0 + O0, E0 - O0).  0 = the layout above
   * (the [2C x 2C] block product on hsp_conv1d_mfma_f32). */
  int32_t prod3;
} hsp_dftseg_args;
int hsp_dftseg_fwd_f32(const hsp_dftseg_args* a, void* stream);
int hsp_dftseg_inv_f32(const hsp_dftseg_args* a, void* stream);
int hsp_dftseg_tables_f32(float* fwd, float* inv); /* host buffers of HSP_DFTSEG_TABLE_FLOATS floats each */
/* The two convs of an AMP pair (hierspeechpp_speechsynthesizer.py:380-384: xt = c1(a1(x)); xt = c2(a2(xt))) met in ONE
 * launch: `inv` describes the inverse transform of c1's product (xf = c1's product output, dft = the inverse table, bias;
 * no residual / running sum / post_scale), `fwd` the forward transform of c2's input (xf = c2's spectrum to write, dft =
 * the forward table, act_* = a2, required); same B, C, L.  y of `inv` and x of `fwd` are not read: the tensor between
 * the convs exists in LDS only.  hsp_dftseg_pair_supported: 1 if the pair fits (both unchunked, two row stretches in one
 * CU's LDS), else 0 and the caller runs hsp_dftseg_inv_f32 + hsp_dftseg_fwd_f32.
 * Pass-through form (round 6; version 102): `inv->y` given (with `inv->res` or without) -- the tensor between the two
 * transforms IS needed elsewhere: the seam between two iterations of an AMP block (hierspeechpp_speechsynthesizer.py:
 * 380-384: x = c2(...) + x; xt = a1'(x) ...), where x_new = inverse(c2's product) + bias + res is the residual of the next
 * iteration.  The launch then writes x_new to `inv->y` (16-B addressable rows) and transforms act(x_new) for the next
 * conv: the inverse launch, the forward launch and one read of x_new become one launch.  Same arithmetic in the same
 * order as the two launches. */
int hsp_dftseg_pair_supported(const hsp_dftseg_args* inv, const hsp_dftseg_args* fwd);
int hsp_dftseg_pair_f32(const hsp_dftseg_args* inv, const hsp_dftseg_args* fwd, void* stream);

/* Is the frequency-domain form available for this geometry?  1 when hsp_dftseg_fwd_f32 / hsp_dftseg_inv_f32 accept the
 * arguments' B, C, L, k, dil, pad, nseg, Np, xf_bs (pointers are not read): dilation <= 8, the spectrum of one launch
 * below 4 GiB (the kernels address it with 32-bit byte offsets: xf_bs * 256 <= 0xffffffff), item counts within int, the
 * LDS stretch of one row chunk within a CU's 160 KB.  0 = the caller keeps the direct conv (hsp_conv1d_mfma_f32), which
 * has no such limit (hierspeechpp_speechsynthesizer.py:377-386,635-651: the reference has no batch or length limit). */
int hsp_dftseg_supported(const hsp_dftseg_args* a);

/* ------------------------------------------------ channel product in three real products per bin (round 5)
 * yf[bin] = conj(rfft(w, 128))[bin] xf[bin] for every bin of a spectrum written by hsp_dftseg_fwd_f32 with prod3 = 1
 * (csrc/hsp_cprod3.hip): with conj(W) = a + i b and X = Xr + i Xi (rows [0, C) / [C, 2C) of a bin's [2C][Np] matrix)
 *     k1 = (a + b) Xr,  k2 = a (Xi - Xr),  k3 = b (Xr + Xi);   Yr = k1 - k3,  Yi = k1 + k2
 * -- 6 C^2 instead of 8 C^2 flops per column and 3 C^2 instead of 4 C^2 weights per bin.  w: [bins][3][C][C], the matrices
 * (a + b), a, b of every bin stored [input channel][output row] (hsp_dftseg_weight_spectrum_f32 writes them; bin 0:
 * (0, W_nyquist, W_dc)).  C a multiple of 64, Np a multiple of 4, xf / w 16-B aligned, xf_bs / yf_bs multiples of 4;
 * `zeros`: >= 16 B of zeros in device memory (the source of DMA lanes past Np).  Replaces the one launch of
 * hsp_conv1d_mfma_f32 with w_bs != 0 between the two transforms (reference: the C x C channel mix of the k = 7 / 11
 * convs of AMPBlock1, hierspeechpp_speechsynthesizer.py:349-364). */
typedef struct hsp_cprod3_args {
  const float* xf;
  float* yf;
  const float* w;
  const float* zeros;
  int64_t xf_bs, yf_bs; /* floats between bins, >= 2 C Np */
  int32_t bins, C, Np;
  int32_t debug;        /* must be 0 (kernel decomposition switches of the tuning build, libhsp_tune.so) */
} hsp_cprod3_args;
int hsp_cprod3_f32(const hsp_cprod3_args* a, void* stream);
int hsp_cprod3_supported(const hsp_cprod3_args* a);
/* The per-bin matrices of a conv from its packed taps w[k][C][w_ld] (hsp_conv1d_args.w of a C -> C conv, row m of input
 * channel ci and tap j at w[(j * C + ci) * w_ld + m]), on the device: 128-point DFT of the k <= 64 taps of every (ci, m)
 * pair in float64 with the twiddles of `tw` (256 doubles in device memory: cos(2 pi n / 128), n < 128, then sin),
 * conjugated and rounded once to fp32.  form HSP_WSPEC_THREE: out [64][3][C][C] for hsp_cprod3_f32; HSP_WSPEC_BLOCK: out
 * [64][2C][2C], the block matrices of the hsp_conv1d_mfma_f32 form (w_bs = 4 C^2).  Every rank of a multi-GPU job derives
 * them from the broadcast taps (the broadcast carries the folded weights only). */
#define HSP_WSPEC_BLOCK 0
#define HSP_WSPEC_THREE 1
int hsp_dftseg_weight_spectrum_f32(const float* w, int32_t k, int32_t C, int32_t w_ld, const double* tw, float* out,
                                   int32_t form, void* stream);

/* ------------------------------------------------ SURVEY.md §8(b) names (dispatching entry points) */
/* The minimum export set of SURVEY.md §8(b) under its own names; each forwards to the entry points above.
 * hsp_conv1d_f32: any weight-normed / plain Conv1d or Linear of the path with its fused prologue / epilogue
 *   (PLAIN or GATE rows); picks the MFMA or the VALU kernel by shape exactly as the host mirror does
 *   (stride != 1, Cin / Cout / Lout < 8 or the SiLU prologue -> hsp_conv1d_direct_f32).
 * hsp_convtr1d_f32: ConvTranspose1d as its polyphase conv (rows == HSP_ROWS_SHUFFLE, weights packed with
 *   hip_layers.convtr_pack_map): hierspeechpp_speechsynthesizer.py:292-300,430-436.
 * hsp_wn_layer_f32: one layer of modules.WN (modules.py:156-175): `in_layer` (HSP_ROWS_GATE_WN: conv + g_l +
 *   tanh*sigmoid), then the residual half `res` (x = (x + rs[:H]) * mask) and the skip half `skip`
 *   (out += rs[H:]) of res_skip_layers; either of the two may be NULL (last layer: skip only).  `res` and `skip`
 *   read in_layer->y.  Issued layer by layer (round 2's one-launch kernel was retired in round 4: it lost to the
 *   separate launches at every batch grouping of the step, profiles/r04_stage_split_policies.txt).
 * hsp_ffn_conv_f32: modules.FFN_Conv (modules.py:382-388) = `fc1` (conv k + bias + pointwise function) then `fc2`
 *   (1x1 conv reading fc1->y, with its mask / per-channel gate / residual epilogue); two launches.
 * hsp_layernorm_modulate_f32: LayerNorm (no affine) + mask + modulate of the DiT blocks (modules.py:346-347,
 *   409-410) = hsp_layernorm_mod_f32 without gamma / beta. */
int hsp_conv1d_f32(const hsp_conv1d_args* a, void* stream);
int hsp_convtr1d_f32(const hsp_conv1d_args* a, void* stream);
int hsp_wn_layer_f32(const hsp_conv1d_args* in_layer, const hsp_conv1d_args* res, const hsp_conv1d_args* skip,
                     void* stream);
int hsp_ffn_conv_f32(const hsp_conv1d_args* fc1, const hsp_conv1d_args* fc2, void* stream);
int hsp_layernorm_modulate_f32(const float* x, float* y, int32_t B, int32_t C, int32_t T, float eps,
                               const float* mask, const float* shift, const float* scale, int64_t mod_bs,
                               void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* HSP_H_ */
