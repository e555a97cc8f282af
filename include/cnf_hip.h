/*
 * cnf_hip.h — C ABI of libcnf_hip.so, the MI355X (gfx950) implementation of CategoricalNF's
 * coupling-layer hot path (forward / inverse / log-det-Jacobian).
 *
 * The reference (phlippe/CategoricalNF) is pure eager PyTorch and has no FFI of its own
 * (SURVEY.md §8b); each entry point below replaces one eager op chain and cites it as
 * <file>:<lines> relative to the reference root.  A maintainer binds these from Python with
 * ctypes; categoricalnf_amd/_lib.py is that binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated otherwise; tensors are dense, row-major,
 *     fp32 latents `[B, N, D]`, int64 categories `[B, N]`, fp32 0/1 padding masks `[B, N]`
 *     (the reference's `[B,N,1]` channel_padding_mask, same memory);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous;
 *   - return value: CNF_OK or a CNF_ERR_* code (cnf_last_error() gives a message; host side only);
 *   - `flags` (nullable) points to one int32 in device memory into which kernels OR the CNF_FLAG_*
 *     bits; the Python layer turns them into the reference's AssertionError / RuntimeError once
 *     per forward instead of one host sync per layer (flow_model.py:42);
 *   - `ldj_in` (nullable, may alias `ldj_out`) is the running log-det `[B]`; kernels write
 *     `ldj_out[b] = (ldj_in ? ldj_in[b] : 0) + layer_ldj[b]` — the reference's `ldj = ldj + layer_ldj`
 *     (flow_model.py:44) folded into the kernel epilogue;
 *   - masks: `mask` is the small coupling mask buffer `[mask_rows, mask_cols]` (1 = channel is
 *     fed to the subnet and left unchanged); `mask_rows` is 1 (channel mask, `mask_cols == D`) or
 *     a period along N (chess mask `[2,1]`), expanded exactly like
 *     coupling_layer.py:67-74 (`_prepare_mask`); NULL = no mask (everything transformed).
 */
#ifndef CNF_HIP_H
#define CNF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNF_OK 0
#define CNF_ERR_ARG 1          /* bad size / NULL pointer / unsupported combination */
#define CNF_ERR_LAUNCH 2       /* hipGetLastError() != hipSuccess after the launch */
#define CNF_ERR_UNSUPPORTED 3  /* shape outside what the kernels are built for */

#define CNF_FLAG_NAN_Z 1       /* a latent output is NaN          (flow_model.py:42) */
#define CNF_FLAG_NAN_LDJ 2     /* a log-det output is NaN         (activation_normalization.py:46).  The token-pass mixture kernels
                                * sum a row's terms in 31.32 fixed point (order-independent bits); terms that format cannot take
                                * (|term| >= 2^14, +-inf, NaN) meet in an fp64 word beside it, so a row follows the reference's
                                * floating-point sum: finite for large finite terms, +-inf for infinite ones, NaN (and this flag) for NaN */
#define CNF_FLAG_RANGE 4       /* inverse-CDF input outside (0,1) (mixture_cdf_layer.py:238-239) */
#define CNF_FLAG_CATEGORY 8    /* a category index outside [0, C)     (general/mutils.py:264, the one_hot assert) */

typedef void* cnf_stream_t;

int cnf_abi_version(void);
const char* cnf_last_error(void);

/* Arithmetic mode.  1 (default, "fast"): hardware v_exp_f32 / v_log_f32 / v_rcp_f32 in the affine, prior and
 * sampling kernels (absolute error ~1e-7), and the module-form mixture-CDF coupling (cnf_mixture_coupling,
 * forward and Newton inverse) in fp32 on LDS-staged parameter rows with two-sided tail sums and an in-kernel
 * fp64 branch for |logit| > 20.7 (<= 1e-5 from the fp64 kernel).
 * 0 ("exact"): ocml expf / tanhf, fp64 logit in the sampler, the mixture-CDF coupling in fp64 throughout like the
 * reference (mixture_cdf_layer.py:62,173-178): the reference's expressions on the same DMA-staged rows as mode 1, library
 * exp, log / log1p / 1/(1+e) to <= 1 ulp (csrc/cnf_f64_math.h); with inverse mode 1 the inverse is the fp32 Newton root
 * polished by the same safeguarded iteration in fp64 (step <= 1e-11 of the smallest scale), with inverse mode 0 the
 * reference's bisection on the round-1 fp64 kernel.
 * The static fp64 API (cnf_mixture_transform) is fp64 in both modes. */
void cnf_set_math_mode(int mode);
/* Mixture-CDF inverse: 0 = the reference's bisection (mixture_cdf_layer.py:235-264, per-element stop at
 * |dx| <= 1e-10); 1 (default) = safeguarded Newton (rtsafe) on the same equation, bracketed by the
 * component quantiles mu_k + s_k logit(u) and started at their weighted mean, same stop: same root to
 * ~1e-10, several times fewer CDF evaluations. */
void cnf_set_inverse_mode(int mode);

/* ---- affine coupling -------------------------------------------------------------------- */

/* coupling_layer.py:42-65 (CouplingLayer.forward after the subnet), :76-98.
 * nn_out [B,N,2D] holds interleaved (s_raw, t) per channel.  scaling_factor [D] or NULL
 * (NULL = no tanh bound, get_coup_params(scaling_factor=None)).
 * fwd: z' = (z + t) e^s, ldj += sum s ; rev: z' = z e^-s - t, ldj -= sum s. */
int cnf_affine_coupling(const float* z, const float* nn_out, const float* scaling_factor,
                        const float* mask, int mask_rows, int mask_cols,
                        const float* ldj_in, float* z_out, float* ldj_out,
                        int B, int N, int D, int reverse, int* flags, cnf_stream_t stream);

/* CouplingLayer.forward of one flow step (coupling_layer.py:42-65) followed by the ActNormFlow and InvertibleConv of the NEXT step
 * (activation_normalization.py:24-48, permutation_layers.py:106-136) in ONE pass — or, reverse != 0, the coupling's inverse followed
 * by the inverted convolution and ActNorm of its own step (conv_weight = the inverse weight, as InvertibleConv._get_weight(inverse =
 * True) hands it out): the order FlowModel walks the layers in either direction.  The coupling's output stays in LDS; z and the
 * log-det are the bits of cnf_affine_coupling followed by cnf_actnorm_invconv, the [B,N,D] round trip between them is gone
 * (S*: 17.2 + 9.7 us as two kernels).  pad / length are the pair's (the affine coupling ignores padding like the reference).
 * D in {2, 3, 4, 6, 8}, N * D a multiple of 4, rows short enough for a wave tile, math mode 1; otherwise CNF_ERR_UNSUPPORTED with
 * nothing launched (run the two entry points). */
int cnf_affine_coupling_actconv(const float* z, const float* nn_out, const float* scaling_factor,
                                const float* mask, int mask_rows, int mask_cols,
                                const float* ldj_in, float* z_out, float* ldj_out,
                                const float* an_bias, const float* an_scales, const float* conv_weight, const float* conv_sldj,
                                const float* pad, const float* length,
                                int B, int N, int D, int reverse, int* flags, cnf_stream_t stream);

/* CouplingLayer.get_coup_params (coupling_layer.py:76-86): materialise s, t [B,N,D]. */
int cnf_affine_params(const float* nn_out, const float* scaling_factor,
                      const float* mask, int mask_rows, int mask_cols,
                      float* s_out, float* t_out, int B, int N, int D, cnf_stream_t stream);

/* CouplingLayer.run_with_params (coupling_layer.py:88-98) on materialised s, t. */
int cnf_affine_transform(const float* z, const float* s, const float* t,
                         const float* ldj_in, float* z_out, float* ldj_out,
                         int B, int N, int D, int reverse, int* flags, cnf_stream_t stream);

/* ---- activation normalisation ------------------------------------------------------------- */

/* ActNormFlow.forward (activation_normalization.py:24-48).  bias, scales [D].
 * ldj += (+-sum_d scales) * len_b with len_b = length[b] (fp32) if given, else sum_n pad[b,n] if
 * pad given, else N.  z' *= pad if given. */
int cnf_actnorm(const float* z, const float* bias, const float* scales,
                const float* pad, const float* length,
                const float* ldj_in, float* z_out, float* ldj_out,
                int B, int N, int D, int reverse, int* flags, cnf_stream_t stream);

/* ExtActNormFlow.forward (activation_normalization.py:116-144) with the predictor output given:
 * nn_out [B,N,2D] = [bias(D) | scales_raw(D)] per token; scales = tanh(scales_raw);
 * ldj += +-sum_{n,d} scales * pad.  z' is NOT multiplied by pad (as in the reference). */
int cnf_ext_actnorm(const float* z, const float* nn_out, const float* pad,
                    const float* ldj_in, float* z_out, float* ldj_out,
                    int B, int N, int D, int reverse, int* flags, cnf_stream_t stream);

/* ActNormFlow.data_init_forward (activation_normalization.py:55-67): per-channel weighted
 * sum / sum of squared deviations.  pass 0: out[0..D) = sum_x, out[D] = count (fp64);
 * pass 1 (given mean[D] fp64): out[0..D) = sum (x-mean)^2.  `out` must be zeroed by the caller. */
int cnf_actnorm_stats(const float* z, const float* pad, const double* mean, double* out,
                      int B, int N, int D, int pass, cnf_stream_t stream);

/* ---- invertible 1x1 convolution ------------------------------------------------------------ */

/* InvertibleConv.forward (permutation_layers.py:106-136): z' = x @ W (right multiply, W [D,D]
 * row-major), z' *= pad, ldj +-= sldj * len_b (len_b = length[b] or N).  For reverse the caller
 * passes W^-1 (computed in fp64 like the reference, :77,:85) and reverse=1 flips the sign.
 * `sldj` points to ONE fp32 in device memory. */
int cnf_invconv(const float* x, const float* weight, const float* sldj,
                const float* pad, const float* length,
                const float* ldj_in, float* z_out, float* ldj_out,
                int B, int N, int D, int reverse, int* flags, cnf_stream_t stream);

/* The LU-parametrised weight of InvertibleConv (permutation_layers.py:61-71, `_get_weight` in training mode):
 * weight_out [D,D] = P (L o tril(-1) + I)(U o triu(1) + diag(sign_s e^log_s)), sldj_out [1] = sum log_s — one launch
 * instead of ~11 tiny eager ops (and ~15 more in their autograd); cnf_invconv_lu_weight_bwd gives g_l, g_u [D,D] (zero
 * outside the strict triangles) and g_log_s [D] from g_weight [D,D] and g_sldj [1] (either may be NULL = zero).  D <= 16,
 * else CNF_ERR_UNSUPPORTED (the caller assembles the weight from tensor ops as the reference does). */
int cnf_invconv_lu_weight(const float* p, const float* l, const float* u, const float* log_s, const float* sign_s,
                          float* weight_out, float* sldj_out, int D, cnf_stream_t stream);
/* cnf_invconv_lu_weight that also hands out W^-1 [D,D] (fp64 Gauss-Jordan, rounded to fp32: permutation_layers.py:76), D <= 8:
 * the bits cnf_actnorm_invconv_bwd(saved_is_output = 1, weight_inv = NULL) computes in a launch of its own — pass it there
 * as weight_inv and that launch disappears. */
int cnf_invconv_lu_weight_inv(const float* p, const float* l, const float* u, const float* log_s, const float* sign_s,
                              float* weight_out, float* sldj_out, float* weight_inv_out, int D, cnf_stream_t stream);
int cnf_invconv_lu_weight_bwd(const float* p, const float* l, const float* u, const float* log_s, const float* sign_s,
                              const float* g_weight, const float* g_sldj, float* g_l, float* g_u, float* g_log_s, int D,
                              cnf_stream_t stream);

/* ActNormFlow followed by InvertibleConv in ONE pass (the first two layers of every flow step in
 * experiments/set_modeling/flow_model.py:55-57, graph_node_flow.py, graphCNF.py): same arithmetic, same order,
 * identical results to cnf_actnorm + cnf_invconv, without the intermediate [B,N,D] round trip.  reverse = 1 runs
 * the pair backwards (convolution with the given inverse weight first, then the inverse ActNorm).
 * D in {1,2,3,4,5,6,8}; otherwise CNF_ERR_UNSUPPORTED (run the two kernels). */
int cnf_actnorm_invconv(const float* z, const float* bias, const float* scales, const float* weight,
                        const float* sldj, const float* pad, const float* length,
                        const float* ldj_in, float* z_out, float* ldj_out,
                        int B, int N, int D, int reverse, int* flags, cnf_stream_t stream);

/* ---- logistic-mixture CDF coupling ----------------------------------------------------------- */

/* MixtureCDFCoupling.forward after the subnet (mixture_cdf_layer.py:45-92) = get_mixt_params
 * (:145-180) + run_with_params (:95-142) + mixture_inv_cdf (:235-264); fp64 inside in math mode 0, fp32 with an
 * fp64 tail branch in math mode 1 (cnf_set_math_mode).
 * nn_out [B,N,D*(2+3K)] channel-major blocks [t, log_s, log_pi[K], mixt_t[K], mixt_log_s[K]].
 * scaling_factor [D] / mixture_scaling_factor [D,K] nullable.  pad [B,N] nullable.
 * reg_out [B] (nullable) receives sum_{n,d} reg_ldj (detail "regularizer_ldj", :79-80).
 * act_host (HOST pointer, nullable): for a channel mask (mask_rows == 1) the n_act channel indices
 * with mask == 0, so that only transformed channels get a lane; NULL = every lane tests the mask.
 * pad_output: multiply z' by pad (MixtureCDFCoupling :76; the autoregressive variant passes
 * mask=NULL and transforms padded tokens too, autoregressive_coupling.py:38-45 -> pad_in_transform=0). */
int cnf_mixture_coupling(const float* z, const float* nn_out,
                         const float* scaling_factor, const float* mixture_scaling_factor,
                         const float* mask, int mask_rows, int mask_cols,
                         const int* act_host, int n_act,
                         const float* pad, int pad_in_transform, int pad_output,
                         const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                         int B, int N, int D, int K, int reverse,
                         double reg_max, double reg_factor, int is_training,
                         int* flags, cnf_stream_t stream);

/* The same with a workspace for small batches of long rows (B < ~2000 samples of >= 2 passes each: graphs,
 * sentences).  There a row is shared by several workgroups so that the 256 CUs are filled; their partial log-det
 * sums meet in fixed-point words of `workspace` (cnf_mixture_workspace_bytes(B) bytes, 8-byte aligned, device
 * memory).  The caller zero-fills the workspace ONCE; every launch leaves it zeroed.  One workspace per stream that
 * may run these kernels concurrently.  workspace == NULL: as cnf_mixture_coupling (one workgroup per row at most). */
int64_t cnf_mixture_workspace_bytes(int B);
int cnf_mixture_coupling_ws(const float* z, const float* nn_out,
                            const float* scaling_factor, const float* mixture_scaling_factor,
                            const float* mask, int mask_rows, int mask_cols,
                            const int* act_host, int n_act,
                            const float* pad, int pad_in_transform, int pad_output,
                            const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                            int B, int N, int D, int K, int reverse,
                            double reg_max, double reg_factor, int is_training,
                            void* workspace, int64_t workspace_bytes,
                            int* flags, cnf_stream_t stream);

/* Forward mixture coupling as the LAST layer of a flow, with the NLL assembly as its epilogue
 * (mixture_cdf_layer.py:45-92 followed by the task's loss: experiments/set_modeling/task.py:96-118,
 * graph_coloring/task.py:122-130, general/task.py:148-149): besides z' and ldj_out it writes
 * neglog_out[b] = -sum_{n,d} log p(z'[b,n,d]) pad[b,n]  (logistic prior, mu = 0; nullable) and
 * nll_out[b] = (neglog_out[b] - ldj_out[b]) / length[b]  (length NULL = N), so the prior term never re-reads z'.
 * nll_acc (nullable): the batch sum in 64 fixed-point words, as cnf_affine_coupling_nll_acc.  Needs math mode 1. */
int cnf_mixture_coupling_nll(const float* z, const float* nn_out,
                             const float* scaling_factor, const float* mixture_scaling_factor,
                             const float* mask, int mask_rows, int mask_cols,
                             const int* act_host, int n_act,
                             const float* pad, int pad_in_transform, int pad_output,
                             const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                             const float* length, float* neglog_out, float* nll_out, int64_t* nll_acc,
                             int B, int N, int D, int K,
                             double reg_max, double reg_factor, int is_training,
                             float sigma, float log_sigma,
                             void* workspace, int64_t workspace_bytes,
                             int* flags, cnf_stream_t stream);

/* MixtureCDFCoupling.forward of flow step i (mixture_cdf_layer.py:45-92, with z' *= pad) followed by the ActNormFlow
 * and InvertibleConv of step i+1 (activation_normalization.py:24-48, permutation_layers.py:106-136; forward direction)
 * in ONE pass: the coupling's output gets z'' = (((z' + bias) e^{scales}) pad) @ W pad before it is written, and
 * ldj_out = ldj_in + coupling log-det + sum(scales) len_a + sldj len_c  (len = length[b] if given, else sum(pad) for
 * the ActNorm and N for the convolution).  Bit-identical to cnf_mixture_coupling(_ws) followed by cnf_actnorm_invconv;
 * the intermediate [B,N,D] tensor is neither written nor read (8 B/elem of HBM traffic less per flow step).
 * conv_weight [D,D] is the forward weight (InvertibleConv._get_weight), conv_sldj [1] its log|det|. */
int cnf_mixture_coupling_actconv(const float* z, const float* nn_out,
                                 const float* scaling_factor, const float* mixture_scaling_factor,
                                 const float* mask, int mask_rows, int mask_cols,
                                 const int* act_host, int n_act,
                                 const float* pad,
                                 const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                                 const float* an_bias, const float* an_scales, const float* conv_weight, const float* conv_sldj,
                                 const float* length,
                                 int B, int N, int D, int K,
                                 double reg_max, double reg_factor, int is_training,
                                 void* workspace, int64_t workspace_bytes,
                                 int* flags, cnf_stream_t stream);

/* ---- compact parameter layout --------------------------------------------------------------------
 * The reference's coupling sub-network emits parameter blocks for ALL D channels and get_mixt_params multiplies the blocks of
 * the untransformed channels by the zero mask (mixture_cdf_layer.py:65-78, 163-171): with a channel mask half of nn_out is
 * computed, written, fetched (whole 128-byte lines) and — in the backward — written again as zeros, for nothing.  The three
 * entry points below take `nn_compact` = fp32 [B, N, n_act * (2 + 3K)]: the blocks of the transformed channels only, in
 * channel order (what the sub-network's last Linear produces when only its rows d0 P .. (d0 + n_act) P are applied:
 * layers.flows.MixtureCDFCoupling(compact_params=True) slices those rows at call time, parameters and checkpoints
 * unchanged).  Everything else as cnf_mixture_coupling_ws / _nll / _actconv; results are identical to theirs on the expanded
 * tensor (same kernels, same arithmetic: only the address of a token's span changes).  Needs a channel mask ([1,D]) with its
 * host channel list (act_host, one contiguous range).  Served by the token-pass kernels only (math mode 1, or math mode 0 with
 * inverse mode 1) and for tensors of a multiple of 4 floats (B N n_act (2+3K) % 4 == 0: the staging DMA's last 16-byte chunk):
 * a shape or mode they decline returns CNF_ERR_UNSUPPORTED with nothing launched — the caller expands
 * nn_compact to the reference layout and calls the plain entry point. */
int cnf_mixture_coupling_compact(const float* z, const float* nn_compact,
                                 const float* scaling_factor, const float* mixture_scaling_factor,
                                 const float* mask, int mask_rows, int mask_cols,
                                 const int* act_host, int n_act,
                                 const float* pad, int pad_in_transform, int pad_output,
                                 const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                                 int B, int N, int D, int K, int reverse,
                                 double reg_max, double reg_factor, int is_training,
                                 void* workspace, int64_t workspace_bytes,
                                 int* flags, cnf_stream_t stream);
int cnf_mixture_coupling_compact_nll(const float* z, const float* nn_compact,
                                     const float* scaling_factor, const float* mixture_scaling_factor,
                                     const float* mask, int mask_rows, int mask_cols,
                                     const int* act_host, int n_act,
                                     const float* pad, int pad_in_transform, int pad_output,
                                     const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                                     const float* length, float* neglog_out, float* nll_out, int64_t* nll_acc,
                                     int B, int N, int D, int K,
                                     double reg_max, double reg_factor, int is_training,
                                     float sigma, float log_sigma,
                                     void* workspace, int64_t workspace_bytes,
                                     int* flags, cnf_stream_t stream);
int cnf_mixture_coupling_compact_actconv(const float* z, const float* nn_compact,
                                         const float* scaling_factor, const float* mixture_scaling_factor,
                                         const float* mask, int mask_rows, int mask_cols,
                                         const int* act_host, int n_act,
                                         const float* pad,
                                         const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                                         const float* an_bias, const float* an_scales, const float* conv_weight, const float* conv_sldj,
                                         const float* length,
                                         int B, int N, int D, int K,
                                         double reg_max, double reg_factor, int is_training,
                                         void* workspace, int64_t workspace_bytes,
                                         int* flags, cnf_stream_t stream);

/* MixtureCDFCoupling.get_mixt_params (mixture_cdf_layer.py:145-180): split + bound + mask in
 * fp32, results cast to fp64: t, log_s [B,N,D]; log_pi, mixt_t, mixt_log_s [B,N,D,K]. */
int cnf_mixture_params(const float* nn_out, const float* scaling_factor,
                       const float* mixture_scaling_factor,
                       const float* mask, int mask_rows, int mask_cols,
                       double* t, double* log_s, double* log_pi, double* mixt_t, double* mixt_log_s,
                       int B, int N, int D, int K, cnf_stream_t stream);

/* MixtureCDFCoupling.run_with_params (mixture_cdf_layer.py:95-142) on fp64 split parameters.
 * mask_full / pad: fp64-free fp32 [B,N,D]-broadcast description as above.  Outputs fp64:
 * z_out [B,N,D], ldj_out [B] (NOT accumulated: the reference overwrites), reg_ldj [B,N,D] nullable. */
int cnf_mixture_transform(const double* z, const double* t, const double* log_s,
                          const double* log_pi, const double* mixt_t, const double* mixt_log_s,
                          const float* mask, int mask_rows, int mask_cols,
                          const int* act_host, int n_act, const float* pad,
                          double* z_out, double* ldj_out, double* reg_ldj,
                          int B, int N, int D, int K, int reverse,
                          double reg_max, double reg_factor, int is_training,
                          int* flags, cnf_stream_t stream);

/* ---- logistic prior and NLL ------------------------------------------------------------------ */

/* LogisticDistribution.log_prob (distributions.py:129-136,154-163), element-wise. */
int cnf_logistic_log_prob(const float* x, float* logp, int64_t n, float mu, float sigma,
                          float log_sigma, int* flags, cnf_stream_t stream);

/* LogisticDistribution.sample given the uniform draw (distributions.py:139-145,117-127):
 * u' = u(1-eps)+eps/2, x = logit(u') (fp64 -> fp32 in math mode 0; log u' - log(1-u') in fp32 in mode 1),
 * x*sigma+mu. */
int cnf_logistic_from_uniform(const float* u, float* x, int64_t n, float mu, float sigma, float eps,
                              cnf_stream_t stream);

/* NLL assembly (experiments/set_modeling/task.py:96-118, general/task.py:148-149):
 * neglog[b] = -sum_{n,d} logp(z)*pad ; nll[b] = (-ldj[b] + neglog[b]) / length[b].
 * sums (nullable, 2 fp64; needs nll_out) = {sum_b nll[b], B} of THIS call (overwritten, fixed summation
 * order) — the pair that is all-reduced over ranks. */
int cnf_prior_nll(const float* z, const float* pad, const float* ldj, const float* length,
                  float* neglog_out, float* nll_out, double* sums,
                  int B, int N, int D, float sigma, float log_sigma, cnf_stream_t stream);

/* sums[0..1] = {sum_b nll[b] (fp64, fixed order), B}: the batch reduction of cnf_prior_nll on its own
 * (general/task.py:148-149 mean over the batch, as the pair that is all-reduced over ranks). */
int cnf_nll_sum(const float* nll, int B, double* sums, cnf_stream_t stream);

/* Last coupling layer of a flow + NLL assembly in one pass: cnf_affine_coupling(reverse = 0) followed by
 * cnf_prior_nll on its z_out / ldj_out (coupling_layer.py:42-65 then set_modeling/task.py:96-118), with the
 * prior term accumulated while z_out is still in registers (saves the 4 B/elem re-read of z_out and one launch).
 * Outputs z_out, ldj_out as cnf_affine_coupling; neglog_out (nullable), nll_out (required), sums (nullable)
 * as cnf_prior_nll. */
int cnf_affine_coupling_nll(const float* z, const float* nn_out, const float* scaling_factor,
                            const float* mask, int mask_rows, int mask_cols,
                            const float* ldj_in, float* z_out, float* ldj_out,
                            const float* pad, const float* length,
                            float* neglog_out, float* nll_out, double* sums,
                            int B, int N, int D, float sigma, float log_sigma,
                            int* flags, cnf_stream_t stream);

/* The same with the batch sum taken INSIDE the kernel: every row adds nll[b] * 2^32 (rounded, signed 64-bit fixed point)
 * to a word in LDS, and the last wave of a workgroup to finish (LDS ticket) sends the workgroup's total with ONE integer
 * atomic to one of 64 global words — integer adds are associative, so the sum is deterministic; the 64 words sit 128
 * bytes apart (one cache line each).  (One global atomic per ROW, rounds 1-2, kept every wave slot occupied until its
 * atomic was acknowledged: 18.3 -> 17.9 us per launch at B = 16384, N = 64, D = 6.)  `acc` =
 * CNF_NLL_ACC_WORDS int64, zeroed by the caller; it may be accumulated over several calls.  Word 16 k is the fixed-point sum
 * of slot k; word 16 k + 1 is the slot's fp64 escape word: a per-sample value the fixed-point word cannot take safely
 * (|nll| >= 4096, +-inf, NaN) is added there with a floating-point atomic, so the batch sum follows the reference's
 * floating-point mean (task.py:96-118) instead of wrapping; with per-sample values below 4096 the integer words hold 3.3e7
 * samples.  cnf_nll_acc_read turns n such words into sums = {sum of (fixed / 2^32 + escape), count}. */
#define CNF_NLL_ACC_WORDS 1024
int cnf_affine_coupling_nll_acc(const float* z, const float* nn_out, const float* scaling_factor,
                                const float* mask, int mask_rows, int mask_cols,
                                const float* ldj_in, float* z_out, float* ldj_out,
                                const float* pad, const float* length,
                                float* neglog_out, float* nll_out, int64_t* acc,
                                int B, int N, int D, float sigma, float log_sigma,
                                int* flags, cnf_stream_t stream);
int cnf_nll_acc_read(const int64_t* acc, int64_t n_words, double count, double* sums, cnf_stream_t stream);

/* ---- mixture-model categorical encoder --------------------------------------------------------- */

/* LinearCategoricalEncoding.forward, num_flows == 0 (linear_encoding.py:59-106,120-133,153-174).
 * categ int64 [B,N]; eps fp32 [B*N,D] logistic noise (cnf_logistic_from_uniform);
 * table fp32 [C,2D] = pred_net(embed_layer.weight) rows [bias | scales_raw];
 * category_prior [C] (log-softmax'ed buffer); pad [B,N] nullable.
 * Outputs: z [B,N,D], ldj_out [B] (+= ldj_in), class_prob_log [B*N] (nullable). */
int cnf_encoder_forward(const int64_t* categ, const float* eps, const float* table,
                        const float* category_prior, const float* pad, float beta,
                        const float* ldj_in, float* z_out, float* ldj_out, float* class_prob_log,
                        int B, int N, int D, int C, float sigma, float log_sigma,
                        int* flags, cnf_stream_t stream);

/* cnf_encoder_forward with LogisticDistribution.sample fused in (distributions.py:139-145,117-127 + linear_encoding.py:59-106):
 * `u` fp32 [B*N,D] is the UNIFORM draw; the kernel squeezes it (u (1 - squeeze_eps) + squeeze_eps / 2), takes the logit and
 * scales by sigma — the arithmetic of cnf_logistic_from_uniform in math mode 1, mu = 0 — and goes on as cnf_encoder_forward:
 * one launch and 8 D bytes per token less than the two calls, the same bits.  eps_out (optional, fp32 [B*N,D]) receives the
 * logistic noise (the backward kernels take it).  Math mode 0 (fp64 logit): CNF_ERR_UNSUPPORTED — run the two calls. */
int cnf_encoder_forward_sampled(const int64_t* categ, const float* u, float squeeze_eps, const float* table,
                                const float* category_prior, const float* pad, float beta,
                                const float* ldj_in, float* z_out, float* ldj_out, float* class_prob_log, float* eps_out,
                                int B, int N, int D, int C, float sigma, float log_sigma,
                                int* flags, cnf_stream_t stream);

/* The sampled encoder forward with the ActNorm + 1x1 convolution of the flow step that follows it applied to the latents
 * before they are written (activation_normalization.py:24-48, permutation_layers.py:106-136; every flow of the reference
 * starts with that pair): the results of cnf_encoder_forward_sampled followed by cnf_actnorm_invconv (forward), bit for bit,
 * in one launch and without the [B,N,D] round trip between them.  ldj_out = ldj_in + encoder log-det + both layers' log-det
 * (ActNorm over `length` | sum(pad) | N, the convolution over `length` | N).  D in {1,2,3,4,5,6,8} and a class table that
 * fits LDS, math mode 1; otherwise CNF_ERR_UNSUPPORTED (run the layers separately). */
int cnf_encoder_forward_actconv(const int64_t* categ, const float* u, float squeeze_eps, const float* table,
                                const float* category_prior, const float* pad, float beta,
                                const float* act_bias, const float* act_scales, const float* conv_weight, const float* conv_sldj,
                                const float* length,
                                const float* ldj_in, float* z_out, float* ldj_out,
                                int B, int N, int D, int C, float sigma, float log_sigma,
                                int* flags, cnf_stream_t stream);
/* cnf_encoder_forward_actconv that also hands out class_prob_log [B*N] (log q_c of every token, taken at the encoder's own
 * latents: linear_encoding.py:163-171) — what cnf_encoder_forward_bwd_cpl wants back in the backward pass. */
int cnf_encoder_forward_actconv_cpl(const int64_t* categ, const float* u, float squeeze_eps, const float* table,
                                    const float* category_prior, const float* pad, float beta,
                                    const float* act_bias, const float* act_scales, const float* conv_weight, const float* conv_sldj,
                                    const float* length,
                                    const float* ldj_in, float* z_out, float* ldj_out, float* class_prob_log,
                                    int B, int N, int D, int C, float sigma, float log_sigma,
                                    int* flags, cnf_stream_t stream);

/* LinearCategoricalEncoding reverse / _posterior_sample (linear_encoding.py:108-118,184-196):
 * argmax_c of (reverse-flow log-prob + category prior) -> int64 [B,N]; first max wins. */
int cnf_encoder_decode(const float* z, const float* table, const float* category_prior,
                       int64_t* categ_out, int B, int N, int D, int C, float sigma, float log_sigma,
                       cnf_stream_t stream);

/* The sampling direction's last three layers in one launch: InvertibleConv.forward(reverse=True) with the inverse weight
 * (permutation_layers.py:106-136), ActNormFlow.forward(reverse=True) (activation_normalization.py:24-48) and the arg-max decode
 * (linear_encoding.py:108-118,184-196): the results of cnf_actnorm_invconv(reverse = 1) followed by cnf_encoder_decode, bit for
 * bit.  ldj_out = ldj_in - both layers' log-det (the decode adds zero).  D in {1,2,3,4,5,6,8} and a class table that fits
 * LDS; otherwise CNF_ERR_UNSUPPORTED (run the layers separately). */
int cnf_encoder_decode_actconv(const float* z, const float* act_bias, const float* act_scales, const float* conv_weight_inv,
                               const float* conv_sldj, const float* pad, const float* length,
                               const float* table, const float* category_prior,
                               const float* ldj_in, int64_t* categ_out, float* ldj_out,
                               int B, int N, int D, int C, float sigma, float log_sigma, int* flags, cnf_stream_t stream);

/* The same two for vocabularies whose class table does not fit LDS (wikitext: 10^4 classes): the classes are walked
 * in chunks whose score constants a workgroup rebuilds in LDS, the sum of class densities (forward) / the arg-max
 * (decode) runs across chunks, and nothing of size [T*C, ...] is materialised (linear_encoding.py:155-160 expands to
 * [T*C, 1, D]).  Same results as cnf_encoder_forward / cnf_encoder_decode.  Beyond 1024 classes the class range is also
 * split over workgroups (up to 32 splits, a function of C only, so a sample's result does not depend on its batch): a
 * second launch merges the per-split density sums / (best, arg-max) pairs in split order.  workspace: cnf_encoder_workspace_floats(B, N, D, C) floats (token
 * log-det terms, which a small kernel sums per row in a fixed order, and the split partials); decode needs it only
 * above 1024 classes (else it may be null). */
int64_t cnf_encoder_workspace_floats(int B, int N, int D, int C);
int cnf_encoder_forward_tiled(const int64_t* categ, const float* eps, const float* table,
                              const float* category_prior, const float* pad, float beta,
                              const float* ldj_in, float* z_out, float* ldj_out, float* class_prob_log,
                              float* workspace,
                              int B, int N, int D, int C, float sigma, float log_sigma,
                              int* flags, cnf_stream_t stream);
/* cnf_encoder_forward_tiled with the sampler fused in (see cnf_encoder_forward_sampled). */
int cnf_encoder_forward_tiled_sampled(const int64_t* categ, const float* u, float squeeze_eps, const float* table,
                                      const float* category_prior, const float* pad, float beta,
                                      const float* ldj_in, float* z_out, float* ldj_out, float* class_prob_log,
                                      float* eps_out, float* workspace,
                                      int B, int N, int D, int C, float sigma, float log_sigma,
                                      int* flags, cnf_stream_t stream);
int cnf_encoder_decode_tiled(const float* z, const float* table, const float* category_prior,
                             int64_t* categ_out, float* workspace, int B, int N, int D, int C, float sigma, float log_sigma,
                             cnf_stream_t stream);

/* ---- sigmoid / logit flow ------------------------------------------------------------------- */

/* SigmoidFlow.forward (sigmoid_layer.py:24-47) after the XOR of the two reverse flags:
 * reverse=0: ldj += sum(-z - 2 softplus(-z)), z' = sigmoid(z);
 * reverse=1: z = z(1-a)+a/2, ldj += sum(-log z - log(1-z) + log(1-a)), z' = log z - log(1-z). */
int cnf_sigmoid_flow(const float* z, const float* ldj_in, float* z_out, float* ldj_out,
                     int B, int L, int reverse, float alpha, int* flags, cnf_stream_t stream);

/* ---- backward (vector-Jacobian products) ------------------------------------------------------------
 * The reference differentiates its eager op chains with autograd (general/train.py:144-155); these entry
 * points return the same gradients.  g_zout [B,N,D] / g_ldj [B] are the upstream gradients of a layer's two
 * outputs (either may be NULL = zero).  Parameter gradients are batch reductions: the caller passes
 * `workspace` with cnf_bwd_workspace_floats(P) floats (P = number of parameter entries of that call); the
 * sum is formed in fp64 in a fixed order (deterministic). */
int64_t cnf_bwd_workspace_floats(int param_count);

/* d(CouplingLayer.forward) (coupling_layer.py:53-63,88-98).  z_out = the forward OUTPUT of the same
 * direction.  g_scaling_factor [D] only when scaling_factor != NULL (P = D). */
int cnf_affine_coupling_bwd(const float* z_out, const float* nn_out, const float* scaling_factor,
                            const float* mask, int mask_rows, int mask_cols,
                            const float* g_zout, const float* g_ldj,
                            float* g_z, float* g_nn, float* g_scaling_factor, float* workspace,
                            int B, int N, int D, int reverse, cnf_stream_t stream);

/* d(CouplingLayer.get_coup_params) / d(CouplingLayer.run_with_params), the static split forms
 * (coupling_layer.py:76-86 / :88-98).  P = D for cnf_affine_params_bwd. */
int cnf_affine_params_bwd(const float* nn_out, const float* scaling_factor, const float* mask, int mask_rows, int mask_cols,
                          const float* g_s, const float* g_t, float* g_nn, float* g_scaling_factor, float* workspace,
                          int B, int N, int D, cnf_stream_t stream);
int cnf_affine_transform_bwd(const float* z_out, const float* s, const float* t, const float* g_zout, const float* g_ldj,
                             float* g_z, float* g_s, float* g_t, int B, int N, int D, int reverse, cnf_stream_t stream);

/* d(ExtActNormFlow.forward) w.r.t. z and the predictor output nn_out [B,N,2D] (activation_normalization.py:127-139). */
int cnf_ext_actnorm_bwd(const float* z_out, const float* nn_out, const float* pad,
                        const float* g_zout, const float* g_ldj, float* g_z, float* g_nn,
                        int B, int N, int D, int reverse, cnf_stream_t stream);

/* d(ActNormFlow.forward) (activation_normalization.py:35-43): g_z, g_bias [D], g_scales [D]; P = 2D. */
int cnf_actnorm_bwd(const float* z_out, const float* bias, const float* scales,
                    const float* pad, const float* length, const float* g_zout, const float* g_ldj,
                    float* g_z, float* g_bias, float* g_scales, float* workspace,
                    int B, int N, int D, int reverse, cnf_stream_t stream);

/* d(InvertibleConv.forward) (permutation_layers.py:112-121): g_x, g_weight [D,D] (of the matrix that was
 * applied), g_sldj [1]; P = D*D + 1. */
int cnf_invconv_bwd(const float* x, const float* weight, const float* pad, const float* length,
                    const float* g_zout, const float* g_ldj,
                    float* g_x, float* g_weight, float* g_sldj, float* workspace,
                    int B, int N, int D, int reverse, cnf_stream_t stream);

/* d(ActNormFlow.forward -> InvertibleConv.forward) of one flow step in ONE kernel (activation_normalization.py:35-43 then
 * permutation_layers.py:112-121, forward direction; the forward is cnf_actnorm_invconv).  The pair's intermediate is
 * recomputed per token: from the pair's input (saved_is_output = 0: the forward's arithmetic, the same bits) or from the
 * pair's output through W^-1 (saved_is_output = 1: the pair ran fused behind a coupling layer / the encoder, whose own
 * output never reached HBM); weight_inv may then be NULL — the inverse is computed on the device in fp64, as the reference
 * inverts (permutation_layers.py:76).  g_params [D*D + 1 + 2D] = d weight | d sldj | d bias | d scales, summed in a fixed
 * order (bit-reproducible); workspace: cnf_bwd_workspace_floats(D*D + 2D + 2).  D in {1..6, 8}, else CNF_ERR_UNSUPPORTED. */
int cnf_actnorm_invconv_bwd(const float* saved, int saved_is_output, const float* bias, const float* scales, const float* weight,
                            const float* weight_inv, const float* pad, const float* length,
                            const float* g_zout, const float* g_ldj, float* g_z, float* g_params, float* workspace,
                            int B, int N, int D, cnf_stream_t stream);

/* d(LogisticDistribution.log_prob) and d(NLL assembly) w.r.t. z (and ldj). */
int cnf_logistic_log_prob_bwd(const float* x, const float* g_logp, float* g_x, int64_t n, float mu, float sigma,
                              cnf_stream_t stream);
int cnf_prior_nll_bwd(const float* z, const float* pad, const float* length, const float* g_nll,
                      float* g_z, float* g_ldj, int B, int N, int D, float sigma, cnf_stream_t stream);

/* d(MixtureCDFCoupling.forward, reverse=False) (mixture_cdf_layer.py:45-123,145-180): g_z, g_nn (same layout as
 * nn_out; zero for untransformed elements), g_scaling_factor [D], g_mixture_scaling_factor [D,K] (each only if the
 * parameter is given); workspace = cnf_bwd_workspace_floats(D + D*K).  The inverse has no backward (the reference
 * never differentiates it). */
int cnf_mixture_coupling_bwd(const float* z, const float* nn_out,
                             const float* scaling_factor, const float* mixture_scaling_factor,
                             const float* mask, int mask_rows, int mask_cols,
                             const float* pad, int pad_in_transform, int pad_output,
                             const float* g_zout, const float* g_ldj,
                             float* g_z, float* g_nn, float* g_scaling_factor, float* g_mixture_scaling_factor,
                             float* workspace,
                             int B, int N, int D, int K, double reg_max, double reg_factor, int is_training,
                             cnf_stream_t stream);

/* The same gradients from the fp32 token-pass kernel (math mode 1; DMA-staged parameter rows, gradients written over
 * the staged rows and stored back coalesced, zeros for untransformed blocks from the same kernel, fp64 branch for the
 * tails); act_host / n_act as in cnf_mixture_coupling.  Shapes the kernel is not built for, and math mode 0, run
 * cnf_mixture_coupling_bwd.  Same workspace. */
int cnf_mixture_coupling_bwd_f32(const float* z, const float* nn_out,
                                 const float* scaling_factor, const float* mixture_scaling_factor,
                                 const float* mask, int mask_rows, int mask_cols,
                                 const int* act_host, int n_act,
                                 const float* pad, int pad_in_transform, int pad_output,
                                 const float* g_zout, const float* g_ldj,
                                 float* g_z, float* g_nn, float* g_scaling_factor, float* g_mixture_scaling_factor,
                                 float* workspace,
                                 int B, int N, int D, int K, double reg_max, double reg_factor, int is_training,
                                 cnf_stream_t stream);

/* The same on the compact parameter layout (cnf_mixture_coupling_compact): nn_compact and g_nn_compact are fp32
 * [B, N, n_act * (2 + 3K)]; there are no untransformed blocks, so nothing is zero-filled — at a half channel mask the kernel
 * reads and writes half of what cnf_mixture_coupling_bwd_f32 does.  d loss / d (last Linear) follows from g_nn_compact through
 * the row slice (autograd: zero rows for the untransformed channels, as the zero blocks give the reference).  Math mode 1 and
 * shapes of the token-pass backward only; otherwise CNF_ERR_UNSUPPORTED (expand, call cnf_mixture_coupling_bwd_f32, slice). */
int cnf_mixture_coupling_compact_bwd_f32(const float* z, const float* nn_compact,
                                         const float* scaling_factor, const float* mixture_scaling_factor,
                                         const float* mask, int mask_rows, int mask_cols,
                                         const int* act_host, int n_act,
                                         const float* pad, int pad_in_transform, int pad_output,
                                         const float* g_zout, const float* g_ldj,
                                         float* g_z, float* g_nn_compact, float* g_scaling_factor, float* g_mixture_scaling_factor,
                                         float* workspace,
                                         int B, int N, int D, int K, double reg_max, double reg_factor, int is_training,
                                         cnf_stream_t stream);

/* d(MixtureCDFCoupling.run_with_params, reverse=False) on fp64 split parameters (static API, :95-123):
 * fp64 gradients for z and the five parameter tensors (zero where nothing is transformed). */
int cnf_mixture_transform_bwd(const double* z, const double* t, const double* log_s, const double* log_pi,
                              const double* mixt_t, const double* mixt_log_s,
                              const float* mask, int mask_rows, int mask_cols, const float* pad,
                              const double* g_zout, const double* g_ldj,
                              double* g_z, double* g_t, double* g_log_s, double* g_log_pi, double* g_mixt_t,
                              double* g_mixt_log_s,
                              int B, int N, int D, int K, double reg_max, double reg_factor, int is_training,
                              cnf_stream_t stream);

/* d(MixtureCDFCoupling.get_mixt_params) (:145-180): five fp64 upstream gradients (nullable) -> g_nn fp32,
 * g_scaling_factor [D], g_mixture_scaling_factor [D,K]; workspace = cnf_bwd_workspace_floats(D + D*K). */
int cnf_mixture_params_bwd(const float* nn_out, const float* scaling_factor, const float* mixture_scaling_factor,
                           const float* mask, int mask_rows, int mask_cols,
                           const double* g_t, const double* g_log_s, const double* g_log_pi, const double* g_mixt_t,
                           const double* g_mixt_log_s,
                           float* g_nn, float* g_scaling_factor, float* g_mixture_scaling_factor, float* workspace,
                           int B, int N, int D, int K, cnf_stream_t stream);

/* d(LinearCategoricalEncoding.forward, num_flows == 0) w.r.t. the class table [C,2D] (linear_encoding.py:59-106,
 * 153-174): g_table [C,2D]; eps / categories / prior are constants.  Tables of any size (word-level vocabularies included):
 * a token-lane pass (log-denominator and d loss / d z per token in one sweep over the class chunks) and a class-lane pass (every
 * lane owns one class and accumulates its table row over the token records), partial tables summed in a fixed order;
 * no [T*C] tensor, no floating-point atomics, bit-reproducible.  On small batches (up to 16 384 tokens, 16 ... 448 classes,
 * D in {1,2,3,4,6,8}) it repeats the forward's density sum and runs the pair kernel of cnf_encoder_forward_bwd_cpl instead.
 * workspace = cnf_encoder_bwd_tiled_workspace_floats(B, N, D, C) floats. */
int64_t cnf_encoder_bwd_tiled_workspace_floats(int B, int N, int D, int C);
int cnf_encoder_forward_bwd_tiled(const int64_t* categ, const float* eps, const float* table,
                                  const float* category_prior, const float* pad, float beta,
                                  const float* g_zout, const float* g_ldj, float* g_table, float* workspace,
                                  int B, int N, int D, int C, float sigma, float log_sigma, cnf_stream_t stream);
/* cnf_encoder_forward_bwd_tiled for a caller that kept the forward's class_prob_log [B*N] (log q_c of every token: an output
 * of cnf_encoder_forward* on the same inputs; linear_encoding.py:163-171).  Every token's denominator is then known up
 * front, and up to 448 classes (D in {1,2,3,4,6,8}) ONE kernel can walk the (token, class) pairs once: a lane owns a class
 * and a few tokens of a stage, the pair terms it computes feed both reductions (over the tokens in its registers for the
 * class row, over the classes through LDS for the token's own-class gradient).  Two workgroup shapes (3 or 7 waves of pair
 * lanes + one token wave); taken where it is faster than the two passes — 9 ... 64 classes at any size (10^6 tokens: 94 vs
 * 117 us at 16 classes, 265 vs 316 at 51), more classes on smaller batches, every class count it can hold up to 16 384 tokens
 * (4 096 tokens x 27 classes: 14 vs 27 us) — the two passes otherwise.  class_prob_log == NULL =
 * cnf_encoder_forward_bwd_tiled.  Same workspace, same reproducibility. */
int cnf_encoder_forward_bwd_cpl(const int64_t* categ, const float* eps, const float* table,
                                const float* category_prior, const float* pad, float beta, const float* class_prob_log,
                                const float* g_zout, const float* g_ldj, float* g_table, float* workspace,
                                int B, int N, int D, int C, float sigma, float log_sigma, cnf_stream_t stream);

/* d(SigmoidFlow.forward) w.r.t. its input (sigmoid_layer.py:31-37). */
int cnf_sigmoid_flow_bwd(const float* z_in, const float* g_zout, const float* g_ldj, float* g_z,
                         int B, int L, int reverse, float alpha, cnf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CNF_HIP_H */
