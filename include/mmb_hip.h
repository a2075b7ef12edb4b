/*
 * mmb_hip.h -- C ABI of libmmb_hip.so: the MI355X (gfx950) hot path of hybrid
 * continuous/discrete bridge matching (reference: dfaroughy/Multimodal-Bridges).
 *
 * The reference has no FFI: its pluggable boundary is Python duck typing through the
 * registry dicts (pipeline/registry.py:9-55).  Each entry point below therefore cites the
 * reference METHOD whose arithmetic it replaces; the Python classes of the same names in
 * multimodal-bridges_amd/ bind these symbols through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - tensors are dense row-major: continuous (B,N,Dc) f32, discrete (B,N) int64
 *     (the reference's (B,N,1) int64 viewed flat), per-jet vectors (B) f32;
 *   - maskf is the canonical f32 0/1 mask (B,N) produced by mmb_mask_prepare;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it, nothing syncs,
 *     nothing allocates (workspaces are caller-provided);
 *   - return 0 on success, <0 on error (MMB_ERR_*); mmb_last_error() gives the message of
 *     the calling thread's last failure.  Invalid arguments never launch a kernel.
 *   - RNG: counter-based Philox4x32-10, key = seed, counter = (element index, offset);
 *     the same (seed, offset) reproduces the same draws bit for bit.
 */
#ifndef MMB_HIP_H
#define MMB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMB_OK 0
#define MMB_ERR_INVALID_ARG (-1)
#define MMB_ERR_LAUNCH (-2)
#define MMB_ERR_UNSUPPORTED (-3)

/* thermostat schedules, model/thermostats.py:24-52 */
#define MMB_THERMO_CONSTANT 0
#define MMB_THERMO_INVERSE 1
#define MMB_THERMO_LINEAR 2
#define MMB_THERMO_INVERSE_SQUARE 3
#define MMB_THERMO_SIGMOID 4

/* continuous bridges, model/bridges.py:8-40 (UniformFlow) and :74-108 (SchrodingerBridge) */
#define MMB_BRIDGE_UNIFORM_FLOW 0
#define MMB_BRIDGE_SCHRODINGER 1

/* activations used by the EPiC layers */
#define MMB_ACT_NONE 0
#define MMB_ACT_GELU 1   /* exact erf form, nn.GELU() default  (epic.py:124) */
#define MMB_ACT_LRELU 2  /* leaky_relu slope 0.01              (epic.py:172) */
#define MMB_ACT_RELU 3   /* nn.TransformerEncoderLayer's FFN activation (particle_transformer.py:29-35) */
#define MMB_ACT_SELU 4   /* torch.nn.SELU of the toy MLP (notebooks/data_toys.ipynb cell 3) */

/* segment kinds of a concatenated linear input (mmb_linear_cat) */
#define MMB_SEG_ROW 0       /* (M,K_s) one row per particle                              */
#define MMB_SEG_JET 1       /* (B,K_s) one row per jet, broadcast over its N particles    */
#define MMB_SEG_JET_MASK 2  /* (B,K_s) per jet, multiplied by maskf of the particle       */

const char* mmb_last_error(void);
int mmb_version(void);

/* ---------------------------------------------------------------- masks / source ---- */

/* Canonical mask: maskf = (mask != 0) as f32, counts[b] = sum_n maskf, total[0] = sum.
 * Replaces the implicit `mask.sum()` / `mask.sum(1)` of multimodal_bridge_matching.py:180,190
 * and epic.py:105.  mask_is_i64: 1 -> int64 input (aoj.py:160, noise.py:226), 0 -> f32. */
int mmb_mask_prepare(int B, int N, const void* mask, int mask_is_i64,
                     float* maskf, int32_t* counts, int32_t* total, void* stream);

/* SourceUniform.__call__ (distributions/noise.py:212-243) on device with Philox:
 * x0 ~ N(0,1) (B,N,Dc) [NULL to skip], k0 ~ U{0..V-1} (B,N) int64 [NULL to skip],
 * mask: all ones, or (sample_masks) n ~ Categorical(pmf[0..n_pmf-1]) then ones in
 * min(n,N) uniformly random slots (noise.py:215-226).  pmf is a device array of n_pmf
 * floats (need not be normalised).  mask_i64 (B,N) and/or maskf (B,N) may be NULL. */
int mmb_source_uniform(int B, int N, int Dc, int V, int sample_masks,
                       const float* pmf, int n_pmf, uint64_t seed, uint64_t offset,
                       float* x0, int64_t* k0, int64_t* mask_i64, float* maskf,
                       int32_t* counts, void* stream);

/* ---------------------------------------------------------------- K-A bridge sampling */

/* sample_bridges (multimodal_bridge_matching.py:194-225) with INJECTED noise:
 *   t[b]  = eps + (1-eps)*t_u[b]                                         (:202)
 *   x_t   = t*x1 + (1-t)*x0 + std*z, std = sigma (UniformFlow, bridges.py:22-28)
 *           or sigma*sqrt(t(1-t)) (SchrodingerBridge, bridges.py:87-93)
 *   k_t   = inverse-CDF draw with uniform u from the telegraph bridge posterior
 *           p(v) = p(v->k1;t,1) p(k0->v;0,t) / p(k0->k1;0,1)   (bridges.py:124-135,165-213)
 * Continuous part is skipped when x0 == NULL, discrete part when k0 == NULL. */
int mmb_bridge_sample(int B, int N, int Dc, int V,
                      const float* t_u, float eps,
                      const float* x0, const float* x1, const float* z,
                      const int64_t* k0, const int64_t* k1, const float* u,
                      int bridge_kind, float sigma, int thermo_kind, float gamma,
                      float* t_out, float* xt, int64_t* kt, void* stream);

/* Same, fused-RNG mode (training): draws t, the SourceUniform source (x0,k0), z and u from
 * Philox in registers.  Reads x1,k1; writes t_out (B), x0_out (kept for the drift target
 * x1-x0), xt, kt.  60 B/particle algorithmic. */
int mmb_bridge_sample_rng(int B, int N, int Dc, int V, float eps,
                          const float* x1, const int64_t* k1,
                          int bridge_kind, float sigma, int thermo_kind, float gamma,
                          uint64_t seed, uint64_t offset,
                          float* t_out, float* x0_out, int64_t* k0_out, float* xt, int64_t* kt,
                          void* stream);

/* TelegraphBridge.transition_probability (bridges.py:165-191): probs (B,N,V), unnormalised */
int mmb_telegraph_probs(int B, int N, int V, const float* t, const int64_t* k0, const int64_t* k1,
                        int thermo_kind, float gamma, float* probs, void* stream);

/* bridge drift targets: UniformFlow.drift (bridges.py:30-37) / SchrodingerBridge.drift (:95-103) */
int mmb_bridge_drift(int B, int N, int Dc, const float* t, const float* xt, const float* x0,
                     const float* x1, int bridge_kind, float* out, void* stream);

/* Jet-level observables of a post-processed sample on device (evaluation; datamodules/utils.py:437-475 JetFeatures with
 * the kinematics of ParticleClouds :21-42, the flavour counts of :558-574 and the jet charge of :548-556).
 * x (B,N,Dc>=3) = (pt, eta_rel, phi_rel, ...) | NULL, k (B,N) tokens | NULL, maskf (B,N).
 * out (B,19): numParticles, px, py, pz, e, pt, m, eta, phi, 8 flavour counts, charge, jet charge (kappa = 1). */
int mmb_jet_observables(int B, int N, int Dc, const float* x, const int64_t* k, const float* maskf, float* out,
                        void* stream);

/* Generation post-processing, in place (pipeline/callbacks_generator.py:99-128 + TensorMultiModal.apply_mask,
 * tensorclass.py:96-106): x = x*scale[d] + shift[d] (scale/shift of length Dc, both NULL for no transform:
 * "standardize" -> std/mean, "normalize" -> (max-min)/min); log_pt (:112-116) additionally exp(.) - 1e-6 on what the
 * reference's `paths.continuous[:, :, 0]` selects: log_pt = 1 feature 0 of every particle (a (J,N,Dc) sample),
 * log_pt = 2 every feature of slot 0 of every jet (the same expression on the stacked (T,J,N,Dc) paths that the
 * callback really passes, :69); then x *= mask and k = (k*mask).long().  x (P,Dc) | NULL, k (P) | NULL, maskf (P) f32,
 * N particles per jet (P a multiple of N). */
int mmb_postprocess(int64_t P, int N, int Dc, float* x, int64_t* k, const float* maskf, const float* scale,
                    const float* shift, int log_pt, void* stream);

/* The "onehot" branch of the same function (callbacks_generator.py:118-125): the last V of the D continuous channels
 * carry token scores.  Inverse transform of all D channels (scale/shift of length D), k = argmax of the last V (first
 * maximum), x = the first D - V channels (xout NULL when D == V: the reference deletes the continuous mode), then
 * apply_mask.  xin (P,D) -> xout (P,D-V), kout (P). */
int mmb_postprocess_onehot(int64_t P, int N, int D, int V, const float* xin, float* xout, int64_t* kout,
                           const float* maskf, const float* scale, const float* shift, int log_pt, void* stream);

/* ---------------------------------------------------------------- K-B continuous solver */

/* ContinuousSolver.euler_step (solvers.py:21-25): x += dt*v, in place, n = B*N*Dc floats */
int mmb_euler_step(int64_t n, float* x, const float* v, float dt, void* stream);

/* ContinuousSolver.euler_maruyama_step (solvers.py:27-33): x += dt*v + sigma_t*dW with
 * sigma_t = sigma*sqrt(t(1-t)) (bridges.py:105-108) -- NO sqrt(dt) factor, as the reference.
 * t: per-jet (B) or NULL -> t_scalar.  dW injected (B,N,Dc) or NULL -> Philox(seed,offset). */
int mmb_euler_maruyama_step(int B, int N, int Dc, float* x, const float* v, float dt,
                            const float* t, float t_scalar, float sigma,
                            const float* dW, uint64_t seed, uint64_t offset, void* stream);

/* ---------------------------------------------------------------- K-C jump solver ------ */

/* TelegraphBridge.rate (bridges.py:137-163): rates (B,N,V) = 1 + (wV/(1-w)) softmax(logits) + w q_k */
int mmb_telegraph_rate(int B, int N, int V, const float* logits, const int64_t* k,
                       const float* t, float t_scalar, int thermo_kind, float gamma,
                       float* rates, void* stream);

/* DiscreteSolver.tauleap_step (solvers.py:59-87) fused with rate():
 *   delta_n != NULL : injected Poisson counts (B,N,V) f32 (parity mode), bit-exact k
 *   delta_n == NULL : Philox mode; samples the only events that matter ({no jump},
 *                     {exactly one jump, to v}) directly -- distribution-equivalent.
 *   overflow_clamp  : 0 = "wrap" (default, :78-81), 1 = "clamp" (:83-87)
 *   final_argmax    : 1 on the LAST grid point: k = argmax_v rates, as
 *                     multimodal_bridge_matching.py:253-255 (the jump is discarded)
 *   rates_out       : optional (B,N,V) store of the rates
 * k is updated in place (padded slots too, quirk K3). */
int mmb_rate_tauleap(int B, int N, int V, const float* logits, int64_t* k,
                     const float* t, float t_scalar, float dt, int thermo_kind, float gamma,
                     const float* delta_n, uint64_t seed, uint64_t offset,
                     int overflow_clamp, int final_argmax, float* rates_out, void* stream);

/* DiscreteSolver.euler_step (solvers.py:89-109) fused with rate(); u injected (B,N) or NULL -> Philox */
int mmb_rate_euler_jump(int B, int N, int V, const float* logits, int64_t* k,
                        const float* t, float t_scalar, float dt, int thermo_kind, float gamma,
                        const float* u, uint64_t seed, uint64_t offset,
                        int final_argmax, float* rates_out, void* stream);

/* ---------------------------------------------------------------- K-F loss ------------ */

#define MMB_LOSS_MAX_BLOCKS 2048
/* loss_fn (multimodal_bridge_matching.py:165-192) forward + gradient w.r.t. the heads in
 * one pass.  head_c (B,N,Dc) | NULL, head_d (B,N,V) | NULL.  The drift target is formed on
 * the fly from (t, xt, x0, x1) (bridges.py:30-37 / :95-103).  vocab_freq (V) | NULL.
 * coef (2): dLoss/dL_c, dLoss/dL_d (MultiModeLoss weights).  total: sum(mask) (int32).
 * partials: workspace of 2*MMB_LOSS_MAX_BLOCKS floats.  Outputs: losses (2) = L_c, L_d;
 * d_head_c, d_head_d (same shapes as the heads, may be NULL to skip the gradient). */
int mmb_loss_fwd_bwd(int B, int N, int Dc, int V,
                     const float* head_c, const float* head_d,
                     const float* t, const float* xt, const float* x0, const float* x1, int bridge_kind,
                     const int64_t* k1, const float* maskf, const int32_t* total,
                     const float* vocab_freq, const float* coef,
                     float* partials, float* losses, float* d_head_c, float* d_head_d, void* stream);

/* mmb_loss_fwd_bwd for callers that CANNOT ALLOCATE (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  Same arguments minus the `partials` workspace: the
 * per-block partial sums live in a file-scope `static __device__` array of the library (2*MMB_LOSS_MAX_BLOCKS floats
 * = 16 KiB, held by every GPU the library runs on until the process ends), whose device address is resolved with
 * hipGetSymbolAddress once per device on the first call made with that device current (make that first call outside
 * a graph capture).  Calls no hipMalloc / hipFree / allocating routine.  Limit: B*N <= 2^31 - 1 particles (rejected
 * with MMB_ERR_INVALID_ARG beyond; the grid is capped at MMB_LOSS_MAX_BLOCKS blocks, so the scratch never grows).
 * All calls on one GPU share that GPU's array: two calls that OVERLAP on one GPU (different streams) race on it;
 * calls on one stream are ordered and need no clearing (every launch overwrites the partials it reads). */
int mmb_loss_fwd_bwd_noalloc(int B, int N, int Dc, int V,
                             const float* head_c, const float* head_d,
                             const float* t, const float* xt, const float* x0, const float* x1, int bridge_kind,
                             const int64_t* k1, const float* maskf, const int32_t* total,
                             const float* vocab_freq, const float* coef,
                             float* losses, float* d_head_c, float* d_head_d, void* stream);

/* MultiModeLoss (multimodal_bridge_matching.py:268-306): learnable (s != NULL):
 * loss = sum_i 0.5*exp(-s_i)*L_i + 0.5*s_i ; fixed: sum_i w_i*L_i.
 * coef_out (2) = dLoss/dL_i ; d_s (2) | NULL = dLoss/ds_i. */
int mmb_multimode_coef(const float* s, const float* fixed_w_host2, float* coef_out, void* stream);
int mmb_multimode_loss(const float* losses, const float* s, const float* fixed_w_host2,
                       float* loss_out, float* d_s, void* stream);

/* ---------------------------------------------------------------- K-D/K-E encoder ----- */

/* weight_norm (old API, dim=0; epic.py:4, K5): W[o,:] = g[o] * v[o,:] / ||v[o,:]||.
 * Rows of several layers are processed in one launch: v and W are flat buffers, row r of
 * length row_len[r] starts at row_off[r] (device int32 arrays, n_rows entries). */
int mmb_weight_norm(int n_rows, const int32_t* row_off, const int32_t* row_len,
                    const float* v, const float* g, float* W, float* inv_norm, void* stream);
int mmb_weight_norm_bwd(int n_rows, const int32_t* row_off, const int32_t* row_len,
                        const float* v, const float* g, const float* inv_norm, const float* dW,
                        float* dv, float* dg, void* stream);
/* as above with dW given as `replicas` partial copies rep_stride floats apart (summed on read) and,
 * optionally, the bias gradient: db_out[r] = sum_q db_rep[q*db_stride + r] (both NULL to skip). */
int mmb_weight_norm_bwd_rep(int n_rows, const int32_t* row_off, const int32_t* row_len,
                            const float* v, const float* g, const float* inv_norm, const float* dW,
                            int replicas, int64_t rep_stride, const float* db_rep, int64_t db_stride,
                            float* dv, float* dg, float* db_out, void* stream);

/* Y = act( cat_s[X_s] @ W^T + bias ) (+ residual): the weight-normed Linear of epic.py
 * (F.linear on a torch.cat of [time | x_local | broadcast global], epic.py:95,155,203) with
 * the concatenation never materialised.  M rows, rows_per_jet = N (jet of row r is r/N).
 * Segments (HOST arrays of length n_seg <= 4): seg_ptr (device pointers), seg_width,
 * seg_kind (MMB_SEG_*), seg_ld (leading dimension in elements).  W is (Nout, K) row-major
 * with leading dimension w_ld (a column slice of a wider matrix is allowed).
 * pre_act (M,Nout) | NULL stores the pre-activation (needed by the GELU backward).
 * row_scale (M) | NULL multiplies the finished row.
 * Second-stage epilogue (EPiCLayer, epic.py:196-206 and the encoder skip :89-90):
 *   Y1 = act(z) + residual            -> Y  (may be NULL when only Y2 is wanted)
 *   Y2 = dropout(post_act(Y1)) + res2 -> Y2 (NULL to skip)
 * dropout is the inverted form with a Philox keep-mask (seed, offset, element index);
 * drop_p = 0 disables it. */
int mmb_linear_cat(int M, int rows_per_jet, int n_seg,
                   const void* const* seg_ptr_host, const int* seg_width_host,
                   const int* seg_kind_host, const int* seg_ld_host,
                   const float* maskf,
                   const float* W, int64_t w_ld, const float* bias,
                   int Nout, int act, const float* residual, int64_t res_ld,
                   const float* row_scale,
                   float* Y, int64_t y_ld, float* pre_act,
                   int post_act, float drop_p, uint64_t seed, uint64_t offset,
                   const float* res2, int64_t res2_ld, float* Y2, int64_t y2_ld, void* stream);

/* The post-LN sublayers of nn.TransformerEncoderLayer (particle_transformer.py:29-50; torch's
 * x = norm1(x + sa(x)), x = norm2(x + ff(x))) in one call:
 *   Y1  = act( cat_s[X_s] @ W^T + bias ) + residual      -> Y1 (M,Nout) | NULL (the reverse pass wants it)
 *   Yln = LayerNorm(Y1) * gamma + beta                    -> Yln; stats (M,2) | NULL = (mean, 1/sqrt(var+eps)) per row
 * Segment arguments as mmb_linear_cat.  For M > 4096 rows, 16-aligned segments and Nout in {32, 64, 128} with the
 * weight block within 160 KB of LDS the normalisation runs in the epilogue of the row-streaming kernel (a wave owns
 * whole rows); every other shape runs the linear kernel and mmb_layernorm_fwd back to back (Yln may then serve as
 * the scratch of the first stage when Y1 is NULL).  Same results either way. */
int mmb_linear_cat_ln(int M, int rows_per_jet, int n_seg, const void* const* seg_ptr_host,
                      const int* seg_width_host, const int* seg_kind_host, const int* seg_ld_host,
                      const float* maskf, const float* W, int64_t w_ld, const float* bias, int Nout, int act,
                      const float* residual, int64_t res_ld, float* Y1, int64_t y1_ld, const float* gamma,
                      const float* beta, float eps, float* Yln, int64_t yln_ld, float* stats, void* stream);

/* Reverse pass of an activation folded into the GEMM that produces the gradient of its output (autograd of
 * `linear2(relu(linear1(x)))` in nn.TransformerEncoderLayer, particle_transformer.py:29-50; of the SELU / GELU
 * MLPs of the heads and of the toy model):
 *   Y = ( cat_s[X_s] @ W^T ) * act'(gate)     gate (M,Nout) = the layer's pre-activation (GELU), anything with its
 *                                             sign (MMB_ACT_RELU, _LRELU) or its output (MMB_ACT_SELU)
 * i.e. dZ of the activated layer straight from dZ of its consumer, without the round trip through mmb_act_bwd. */
int mmb_linear_cat_gate(int M, int rows_per_jet, int n_seg, const void* const* seg_ptr_host,
                        const int* seg_width_host, const int* seg_kind_host, const int* seg_ld_host,
                        const float* maskf, const float* W, int64_t w_ld, int Nout, int gate_act, const float* gate,
                        int64_t gate_ld, float* Y, int64_t y_ld, void* stream);

/* EPiCEncoder._meansum_pool (epic.py:100-107) without the concat:
 * pooled[b, 0:H] = sum_n x*mask / counts[b] ; pooled[b, H:2H] = 0.01 * sum_n x*mask. */
int mmb_meansum_pool(int B, int N, int H, const float* x, int64_t x_ld, const float* maskf,
                     const int32_t* counts, float* pooled, int64_t pooled_ld, void* stream);

/* MultiModalEmbedder (embedder.py:63-97) + apply_mask (tensorclass.py:96-106):
 *   temb (B,Et)        = Linear(cat[sin(2 pi t W), cos(2 pi t W)])      per jet
 *   cont (B,N,Ec) f32  = (x @ Wc^T + bc) * mask                          (NULL to skip)
 *   disc (B,N,Ed) f32  = trunc( table[k] * mask )  -- quirk Q1: the reference truncates the
 *                        embedded features to int64; we store the truncated value as f32
 *                        (every consumer promotes it back to f32, epic.py:155)
 * table (V,Ed) is the weight-normed embedding  g*v/||v||. */
int mmb_embed(int B, int N, int Dc, int V, int Et, int Ec, int Ed,
              const float* t, float t_scalar, const float* x, const int64_t* k, const float* maskf,
              const float* fourier_W, const float* time_W, const float* time_b,
              const float* cont_W, const float* cont_b, const float* table,
              float* temb, float* cont, float* disc, float* fourier_feat, void* stream);
/* fourier_feat (B,Et) | NULL: the sin/cos features (kept for the time-Linear weight gradient). */

/* ---------------------------------------------------------------- backward (training) -- */
/* These have no single reference line: they are the hand-written reverse of the layers above
 * (what torch.autograd derives for encoders/epic.py:145-208 and embedder.py:63-83).

 * Epilogue/activation gradient of mmb_linear_cat.  Forward was a = act(z); y1 = a + res1;
 * y2 = dropout(post(y1)) + res2.  Given gY1 and/or gY2:
 *   g1 = gY1 + gY2 * keep/(1-p) * post'(y1) ;  dZ = g1 * act'(z)
 * post' takes its sign from y1, or from y2 - res2 when y1 was not stored; act' needs `pre`
 * (GELU) or y1 (leaky-relu).  All matrices are (M,Nout) with their own leading dimension. */
int mmb_act_bwd(int64_t M, int Nout, const float* gY1, int64_t gY1_ld, const float* gY2, int64_t gY2_ld,
                const float* y1, int64_t y1_ld, const float* y2, int64_t y2_ld,
                const float* res2, int64_t res2_ld, const float* pre, int64_t pre_ld,
                int act, int post, float drop_p, uint64_t seed, uint64_t offset,
                float* dZ, int64_t dz_ld, void* stream);

/* out[b,:] (+)= sum_n x[b,n,:] * (maskf ? maskf[b,n] : 1): gradient of a per-jet segment. */
int mmb_jet_sum(int B, int N, int H, const float* x, int64_t x_ld, const float* maskf,
                float* out, int64_t out_ld, int accumulate, void* stream);

/* gradient of mmb_meansum_pool: dX[b,n,:] (+)= mask*(dP[b,:H]/count[b] + 0.01*dP[b,H:2H]) */
int mmb_pool_bwd(int B, int N, int H, const float* dP, int64_t dp_ld, const float* maskf,
                 const int32_t* counts, float* dX, int64_t dx_ld, int accumulate, void* stream);

/* dW (Nout,K) += dZ^T @ cat_s[X_s] and db (Nout) += column sums of dZ, same segment
 * description as mmb_linear_cat; accumulates with fp32 atomics (zero the buffers first). */
int mmb_grad_weight(int M, int rows_per_jet, int n_seg,
                    const void* const* seg_ptr_host, const int* seg_width_host,
                    const int* seg_kind_host, const int* seg_ld_host, const float* maskf,
                    const float* dZ, int64_t dz_ld, int Nout, float* dW, int64_t dw_ld, float* db,
                    void* stream);

/* Row-streaming weight gradient of ONE row-major input: dW (Nout,K) += dZ^T X, db (Nout) += colsum(dZ)
 * (db may be NULL).  X is a per-particle input (M = B*N rows) or a per-jet matrix (M = B rows; the
 * broadcast time / global-vector segments of a layer are handled by passing the per-jet sums of dZ,
 * mmb_jet_sum, as dZ).  Replaces the autograd of F.linear's weight (Lightning's loss.backward() over
 * epic.py:95-96,155,195-205) like mmb_grad_weight; buffers must be pre-zeroed (atomic accumulation).
 * Same-address atomics serialise at the memory side, so the accumulators may be spread over
 * `replicas` copies (dW + r*rep_stride, db + r*db_stride, block x uses copy x % replicas) that the
 * consumer sums (mmb_weight_norm_bwd_rep does it on the way in); replicas = 1 for a single copy. */
int mmb_grad_weight_rows(int M, const float* X, int64_t x_ld, int K, const float* dZ, int64_t dz_ld,
                         int Nout, float* dW, int64_t dw_ld, float* db,
                         int replicas, int64_t rep_stride, int64_t db_stride, void* stream);

/* The same for a very narrow input (K <= 4, Nout <= 32): the continuous embedding nn.Linear(dim_continuous, Ec) of
 * embedder.py:57 -- a streaming reduction instead of an MFMA tile that would be 29/32 padding.  Accumulates. */
int mmb_grad_weight_smallk(int M, const float* X, int64_t x_ld, int K, const float* dZ, int64_t dz_ld, int Nout,
                           float* dW, int64_t dw_ld, float* db, void* stream);

/* ---- one launch for all layers of a model (a training step re-normalises ~60 small matrices and has ~100
 * weight-gradient problems: launch-bound otherwise).  The tables live in device memory and hold 8-byte fields only
 * (built on the host as int64 tensors).
 * mmb_weight_norm_multi: one record per row {v row, g element | NULL = plain copy, destination row, 1/||v||
 *   destination | NULL, length} -- torch.nn.utils.weight_norm as used in epic.py:4,67,132-141,180-183, embedder.py:61.
 * mmb_weight_norm_bwd_multi: one record per row {v, g, inv, dW row (replica 0), db element (replica 0) | NULL,
 *   dv out, dg out, db out | NULL, length, replicas, replica stride of dW, of db}; outputs are ACCUMULATED (+=), like
 *   autograd accumulates into .grad: zero them before the first backward of a step.
 * mmb_grad_weight_rows_multi: records {X, dZ, dW, db | NULL, x_ld, dz_ld, dw_ld, rep_stride, db_stride, M, K, Nout,
 *   rows per block, replicas, blocks along rows, blocks along K windows, first block id, end block id, row list |
 *   NULL, its length (device int32) | NULL -- with a row list (mmb_mask_rows) the reduction runs over those rows of X
 *   and dZ only}, all of tile
 *   class (OTb, KTb) = mmb_hip.ops.gw_class(K, Nout); semantics per record as mmb_grad_weight_rows (fast path only:
 *   K, Nout, leading dimensions multiples of 4, 16-byte aligned operands). */
int mmb_weight_norm_multi(int n_rows, const void* rows, void* stream);
int mmb_weight_norm_bwd_multi(int n_rows, const void* rows, void* stream);
/* Parameter block of the fused entry points (`packed` / `packed_bwd` below) from the flat effective-weight buffer
 * that mmb_weight_norm_multi writes, one launch per block after every optimizer step (the reference re-normalises
 * inside each Linear.forward, epic.py:67,132-141): dst[i] = flat[gather[i]] (+ flat[gather2[i]] when gather2 is not
 * NULL) for i < n_main, then n_x operand values flat[gather_x[j]] each split exactly into three bf16 planes and
 * stored [fragment j/512][plane][j%512] after the fp32 part.  n_main % 4 == 0, n_x % 512 == 0, dst 16-byte aligned
 * and n_main + 3 n_x / 2 floats long. */
int mmb_pack_params(const float* flat, const int64_t* gather, const int64_t* gather2, int64_t n_main,
                    const int64_t* gather_x, int64_t n_x, float* dst, void* stream);
int mmb_grad_weight_rows_multi(int OTb, int KTb, const void* probs, int n_probs, int total_blocks, void* stream);

/* ---------------------------------------------------------------- optimizer step ----- */
#define MMB_ADAMW_MAX_PARTIALS 1024
/* The optimizer step of the training loop on flat buffers: Lightning's gradient_clip_val (global L2 norm,
 * pipeline/experiment.py:217: g *= min(1, max_norm/(||g|| + 1e-6)) over all n gradients) followed by
 * torch.optim.AdamW (multimodal_bridge_matching.py:151-154; decoupled weight decay, bias-corrected moments, no
 * amsgrad) on the first n_active elements -- parameters that never receive a gradient upstream (grad None there:
 * the truncated embedding Q1, the dead fused layer K2) are laid out behind n_active and left untouched, as
 * torch skips them.  lr is this step's learning rate, step counts from 1.  max_norm <= 0 disables clipping.
 * partials: workspace of MMB_ADAMW_MAX_PARTIALS floats (clipping only); norm_out (1) | NULL receives ||g||;
 * write_clipped != 0 also stores the clipped gradients back (what clip_grad_norm_ leaves in .grad). */
int mmb_adamw_clip_step(int64_t n, int64_t n_active, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                        float max_norm, float* partials, float* norm_out, int write_clipped, void* stream);

/* ---------------------------------------------------------------- transformer set encoder */
/* TransformerEncoderBlock (encoders/particle_transformer.py:8-50) = torch's nn.TransformerEncoderLayer (post-LN,
 * ReLU FFN, batch_first) over the particles of one jet.  The Linear layers (input/time projection, in_proj,
 * out_proj, linear1/2, + residual) are mmb_linear_cat calls; these are the two ops it lacks.
 *
 * mmb_attention_fwd: multi-head scaled-dot-product self-attention per jet.  qkv (B*N, 3H) rows = [q | k | v] as
 * nn.MultiheadAttention's in_proj packs them, head h owns columns h*H/nheads...; out (B*N, H) heads concatenated
 * (the input of out_proj).  Keys are restricted by the mask: valid_is_zero = 1 keeps the keys whose maskf is 0 --
 * what the reference does, src_key_padding_mask = mask > 0 (particle_transformer.py:50, quirk Q4: the REAL particles
 * are the ignored keys) -- valid_is_zero = 0 keeps the keys with maskf != 0.  A jet with no valid key gets zeros
 * (torch's need_weights=False path).  lse (B, nheads, N) | NULL: log-sum-exp of the scaled scores, for the backward.
 * H/nheads <= 64 and 8*N*pad(H/nheads) + 12*N bytes <= 64 KiB of LDS. */
int mmb_attention_fwd(int B, int N, int H, int nheads, const float* qkv, int64_t qkv_ld, const float* maskf,
                      int valid_is_zero, float* out, int64_t out_ld, float* lse, void* stream);
/* its reverse (autograd of F.scaled_dot_product_attention): dqkv (B*N, 3H) is OVERWRITTEN with [dq | dk | dv]
 * (zeros for the keys that took no part) from dout = d loss / d out and the saved out, lse. */
int mmb_attention_bwd(int B, int N, int H, int nheads, const float* qkv, int64_t qkv_ld, const float* maskf,
                      int valid_is_zero, const float* out, int64_t out_ld, const float* dout, int64_t dout_ld,
                      const float* lse, float* dqkv, int64_t dqkv_ld, void* stream);
/* The same two ops with dropout on the attention probabilities, as nn.MultiheadAttention applies it in training mode
 * (nn.TransformerEncoderLayer(dropout=p), particle_transformer.py:29-35): out = dropout(softmax(.)) @ V.  The keep
 * decision of (jet b, head h, query q, j-th valid key) is word j & 3 of Philox4x32(seed, offset, index =
 * ((b*nheads + h)*N + q) * ceil(N/4) + (j >> 2)), kept iff u >= drop_p, kept probabilities scaled by 1/(1 - drop_p);
 * the reverse pass regenerates the bits from the same (seed, offset).  lse is that of the UNdropped scores.
 * drop_p = 0 is mmb_attention_fwd / _bwd.  (torch draws its mask from its own generator: same law, other stream.)
 * keep_bits: NULL, or B*nheads*N*ceil(N/32) words: the forward's matrix-core kernels (head width <= 32, N <= 256) leave
 * the keep bits there, one word per (query, 32-key tile), and the reverse pass reads them instead of regenerating (its
 * key-major phase needs one Philox call per element otherwise); the other kernels ignore the buffer -- the bits are the
 * same function of (seed, offset) either way. */
int mmb_attention_dropout_fwd(int B, int N, int H, int nheads, const float* qkv, int64_t qkv_ld, const float* maskf,
                              int valid_is_zero, float* out, int64_t out_ld, float* lse, float drop_p, uint64_t seed,
                              uint64_t offset, uint32_t* keep_bits, void* stream);
int mmb_attention_dropout_bwd(int B, int N, int H, int nheads, const float* qkv, int64_t qkv_ld, const float* maskf,
                              int valid_is_zero, const float* out, int64_t out_ld, const float* dout, int64_t dout_ld,
                              const float* lse, float* dqkv, int64_t dqkv_ld, float drop_p, uint64_t seed,
                              uint64_t offset, const uint32_t* keep_bits, void* stream);
/* the keep factors of the call above as a dense tensor out (B, nheads, N, N) [b,h,q,n]: 1/(1-drop_p) or 0 at the valid
 * keys n, 0 at the others -- for tests that rebuild the dropped attention with another implementation. */
int mmb_attention_dropout_mask(int B, int N, int nheads, const float* maskf, int valid_is_zero, float drop_p,
                               uint64_t seed, uint64_t offset, float* out, void* stream);
/* The per-particle tail of one encoder layer in one launch, inference arithmetic (no dropout):
 *   h1 = LayerNorm1(h + out_proj(a));  out = LayerNorm2(h1 + linear2(relu(linear1(h1))));  qkv = in_proj_next(out)
 * (nn.TransformerEncoderLayer.forward as built at particle_transformer.py:29-35, post-LN, ReLU), replacing two
 * mmb_linear_cat_ln and two mmb_linear_cat launches: h1 and the FFN's hidden tensor never reach HBM.  a (M,d): attention
 * output; h (M,d): the layer's input (residual); out (M,d); qkv (M,3d) | NULL: the next layer's in_proj applied to out.
 * d = 64 or 128, ff a multiple of 64 (<= 1024).  packed = n_vec = 9d + ff floats [b_o | gamma1 | beta1 | b2 | gamma2 |
 * beta2 | b1 | b_in] followed by the weights in MFMA fragment order, three bf16 planes per fp32 value (INTEGRATION.md,
 * "Parameter block"; mmb_hip/fused.py: pack_transformer_tail): W_o, then per 64-unit FFN chunk c W1[64c:64c+64, :] (block
 * wx_1) and W2[:, 64c:64c+64] (block wx_2), then the three d x d blocks of in_proj; wx_*: float offsets from the start of
 * the fragment part.  a_idx: NULL, or per row the row of `a` to use (< 0: zeros) -- the compacted inference path reads one V
 * row per jet straight out of the qkv buffer.  All row strides multiples of 4 floats, all pointers 16-byte aligned.  Results are fp32 results
 * (exact three-plane split, six partial products). */
int mmb_transformer_tail(int64_t M, int d, int ff, const float* packed, int n_vec, int wx_o, int wx_1, int wx_2, int wx_in,
                         const float* a, int64_t a_ld, const float* h, int64_t h_ld, float* out, int64_t out_ld, float* qkv,
                         int64_t qkv_ld, float eps1, float eps2, const int32_t* a_idx, void* stream);
/* nn.LayerNorm over the last dim (norm1 / norm2 of the encoder layer; biased variance, eps inside the root):
 * y = (x - mean)/sqrt(var + eps) * gamma + beta; stats (M,2) | NULL receives (mean, 1/sqrt(var+eps)) per row. */
int mmb_layernorm_fwd(int64_t M, int D, const float* x, int64_t x_ld, const float* gamma, const float* beta, float eps,
                      float* y, int64_t y_ld, float* stats, void* stream);
/* its reverse: dx (overwritten, or += when accumulate) and dgamma, dbeta (ACCUMULATED with atomics: zero them
 * first).  D <= 512. */
int mmb_layernorm_bwd(int64_t M, int D, const float* x, int64_t x_ld, const float* gamma, const float* stats,
                      const float* dy, int64_t dy_ld, float* dx, int64_t dx_ld, int accumulate, float* dgamma,
                      float* dbeta, void* stream);
/* dst[i] (+)= sum_r src[r*stride + i], i < n: folds the replicated accumulators of mmb_grad_weight_rows for the
 * plain (not weight-normed) nn.Linear layers of the transformer (mmb_weight_norm_bwd_rep does it for the others). */
int mmb_sum_replicas(int64_t n, int replicas, int64_t stride, const float* src, float* dst, int accumulate,
                     void* stream);

/* The whole generation loop of the toy hybrid experiment in ONE launch: MultiModalMLP (notebooks/data_toys.ipynb
 * cell 3: embed -> 4 Linear with SELU -> two heads) inside simulate_dynamics (multimodal_bridge_matching.py:227-259;
 * Euler / Euler-Maruyama drift steps interleaved with tau-leap jumps, final state = argmax of the last rates), as
 * mmb_fused_epic_generate does for the EPiC model.  Compiled for dim_hidden H = 64, Dc = 2, V = 8 (the notebook's and
 * configs/S1_toy.yaml's); other widths run the layered entry points.  weights: 12 device pointers
 * {body.0.weight (H,1+Dc+H/4), body.0.bias, body.2.*, body.4.*, body.6.*, continous_head.weight (Dc,H), .bias,
 * discrete_head.weight (V,H), .bias}; table = embed.weight (V,H/4).  x (B*N,Dc), k (B*N) are updated in place where
 * maskf != 0.  delta_n (steps,B,N,V) / delta_w (steps,B,N,Dc): injected noise or NULL -> Philox(seed, offset + step). */
int mmb_toy_generate(int B, int N, int H, int Dc, int V, const float* table, const float* const* weights,
                     const float* maskf, float* x, int64_t* k, int steps, const float* times, float dt,
                     int thermo_kind, float gamma, const float* delta_n, uint64_t seed, uint64_t offset, int em,
                     float em_sigma, const float* delta_w, void* stream);

/* ---------------------------------------------------------------- K-E fused (generation) */
/* Whole MultiModalEPiC forward (multimodal_epic.py:158-194 incl. embedder.py:74-80) for one jet
 * per workgroup with activations resident in registers; `packed`/`desc` come from
 * mmb_hip.fused.pack_model (fp32 part: transposed per-jet matrices, time/global parts, biases; after it the
 * weight-normed effective weights of the per-particle layers in MFMA fragment order, each fp32 value split exactly
 * into three bf16 planes hi+mid+lo, [KT][2][OT][3][64 lanes][8 bf16]: the kernels form every fp32 product from
 * six bf16 x bf16 partial products accumulated in fp32, results are fp32 results).  Compiled width sets: variant 0: S3
 * (Et32 Ec16 Ed16 H64/64/128 G16/16/32); 1: Et16 Ec8 Ed8 H32/32/64 G8/8/16; 2: Et64 Ec32 Ed32 H64/64/128 G32/32/64;
 * 3: Et32 Ec16 Ed16 H32/32/64 G16/16/32.  nb_branch / nb_branch1: EPiC layers of the continuous / discrete branch
 * encoder (they may differ), nb_fused_exec: executed layers of the fused encoder (K2).  N <= 160 (N > 128 runs the five-wave compilation of the
 * same kernels, one workgroup per CU), Dc = 3, V = 8.  temb: (B,Et) with row stride
 * temb_stride (0 = one row shared by all jets).  which: 0 continuous head, 1 discrete head, 2 both. */
int mmb_fused_epic_forward(int variant, int B, int N, const float* packed, const int32_t* desc,
                           int nb_branch, int nb_branch1, int nb_fused_exec, int use_skip,
                           int off_cw, int off_cb, int off_table,
                           const float* maskf, const int32_t* counts,
                           const float* x, const int64_t* k, const float* temb, int64_t temb_stride,
                           int which, float* head_c, float* head_d, void* stream);

/* The same forward in TRAINING mode (the encoder call of training_step, multimodal_bridge_matching.py:108-121):
 * both heads, the inverted dropout of EPiCLayer (epic.py:196-206; Philox keep-mask indexed like mmb_linear_cat's,
 * offsets offset0+1, +2, ... in execution order) and every activation the reverse pass reads, written row-major
 * from the accumulator registers.  slots: device array of n_slots records {float* p; int64 ld; int64 n}
 * (destination, row pitch, floats available from p) in execution order: per branch [projection: local.0 pre/out,
 * local.2 pre/out, pooled, global.0 pre/out, global.2 pre/out; per layer: pooled, fc_glob1 out, g', g_out, sign words
 * of fc_loc1, fc_loc1 out, sign words of l_out, l_out (sign words: 256 x 64-bit per jet: bit 16t+r of thread i = sign of the leaky-relu
 * input behind register r of tile t); output], the concatenated global vector, fused encoder [layers; output], heads [.0 pre/out] x 2.
 * head_c (B*N,3), head_d (B*N,8).
 * Padded slots: a 32-particle tile without a real particle is idle in both kernels of the pair (the real particles of a
 * jet are put first inside the kernels; rows stay indexed by the true slot).  Its rows of the saved activations are
 * left unwritten -- allocate those buffers zeroed once: the weight-gradient GEMMs multiply them by gradient rows that
 * are exactly zero, so they only have to be finite -- and its heads are written as zeros (the losses are masked).
 * jet_order (B) | NULL: launch order of the jets (mmb_jet_schedule), as for mmb_fused_epic_generate. */
int mmb_fused_epic_train_forward(int variant, int B, int N, const float* packed, const int32_t* desc,
                                 int nb_branch, int nb_branch1, int nb_fused_exec, int use_skip,
                                 int off_cw, int off_cb, int off_table,
                                 const float* maskf, const int32_t* counts,
                                 const float* x, const int64_t* k, const float* temb, int64_t temb_stride,
                                 const void* slots, int n_slots, float drop_p, uint64_t seed, uint64_t offset0,
                                 float* head_c, float* head_d, const int32_t* jet_order, void* stream);

/* Reverse pass of that forward for every jet in ONE launch (what loss.backward() does to the encoder in the
 * reference's training_step): gradients of activations stay in registers; the kernel reads the saved activations and
 * writes, row-major, the dZ of every weight-normed Linear plus the per-jet column sums of dZ (gradients of the
 * broadcast time / global-vector inputs) -- the operands of mmb_grad_weight_rows -- and d temb (B,Et).
 * packed_bwd/desc_bwd: mmb_hip.fused.packed_bwd (transposed x-parts in fragment order).  slots: records as above:
 * [0] d f (rows,2Hb) [1] d h from the heads (rows,2Hb) [2] d cat[h1,h2] (rows,2Hb) [3] d temb (B,Et), then per head
 * {d head, .0 pre, .0 dZ, column sums}, fused encoder {column sums of the output layer; per layer, last first:
 * sign words of l_out, fc_loc2 dZ, sign words of fc_loc1, fc_loc1 dZ, its column sums, g', fc_glob2 dZ, fc_glob1 out,
 * fc_glob1 dZ},
 * per branch the same followed by the projection {global.2 pre, dZ, global.0 pre, dZ, local.2 pre, dZ, local.0 pre,
 * dZ, column sums} and, for the continuous branch, d (continuous embedding) (rows,Ec).
 * use_skip (this entry point and mmb_fused_uni_train_backward): bit 0 = the encoder-level skip connection; bit 1 =
 * the caller reads the per-particle dZ slots at the rows of REAL particles only (weight gradients over the
 * mmb_mask_rows list): padded slots then leave their rows of those slots unwritten (they would be zero rows; the
 * continuous-embedding gradient keeps every row, its consumer reads them all). */
int mmb_fused_epic_train_backward(int variant, int B, int N, const float* packed_bwd, const int32_t* desc_bwd,
                                  int nb_branch, int nb_branch1, int nb_fused_exec, int use_skip,
                                  const float* maskf, const int32_t* counts,
                                  const void* slots, int n_slots, float drop_p, uint64_t seed, uint64_t offset0,
                                  const int32_t* jet_order, void* stream);

/* The same pair for the discrete-only UniModalEPiC (multimodal_epic.py:44-64; one EPiC encoder with projection on
 * cat[t, t, k_emb] whose output layer emits the V logits).  Forward slots: projection 9, layers 8 each.  Backward
 * slots: [0] d temb (B,Et), d logits (rows,8), column sums of the output layer (B,8), layers last..first, projection. */
int mmb_fused_uni_train_forward(int variant, int B, int N, const float* packed, const int32_t* desc,
                                int num_blocks, int use_skip, int off_table,
                                const float* maskf, const int32_t* counts, const int64_t* k,
                                const float* temb, int64_t temb_stride,
                                const void* slots, int n_slots, float drop_p, uint64_t seed, uint64_t offset0,
                                float* head_d, const int32_t* jet_order, void* stream);
int mmb_fused_uni_train_backward(int variant, int B, int N, const float* packed_bwd, const int32_t* desc_bwd,
                                 int num_blocks, int use_skip, const float* maskf, const int32_t* counts,
                                 const void* slots, int n_slots, float drop_p, uint64_t seed, uint64_t offset0,
                                 const int32_t* jet_order, void* stream);

/* simulate_dynamics (multimodal_bridge_matching.py:227-259) in ONE launch: per grid point
 * [forward -> Euler step on x (solvers.py:21-25)] [forward -> telegraph rates (bridges.py:137-163)
 * -> tau-leap jump (solvers.py:59-81)], final k = argmax of the last rates.  x (B,N,3) and k (B,N)
 * are updated in place; padded slots move like real ones (K3).  times (steps) and temb_steps
 * (steps,Et) are device arrays; ctab_steps (steps,n_ctab,128) | NULL holds the precomputed time
 * vectors W_t temb of every layer that consumes the time embedding (t is the same for all jets
 * of a grid point, so this per-jet matvec is hoisted out of the kernel); delta_n (steps,B,N,V)
 * injected Poisson counts or NULL for the Philox jump sampler (seed, offset).
 * em != 0 replaces the Euler step by Euler-Maruyama (solvers.py:27-33, Q5-repaired diffusion
 * sigma_t = em_sigma*sqrt(t(1-t)), NO sqrt(dt)): x += dt v + sigma_t dW with dW from delta_w
 * (steps,B,N,3) or, when NULL, Philox + Box-Muller.
 * Padded slots: a 32-particle tile without a real particle does no per-particle work -- the kernel puts the real
 * particles of a jet first (whatever slots the mask marks; noise.py:36-37 permutes every mask) and the padded slots of
 * idle tiles take drift and logits from one padded slot of a working tile: the same row, so they move exactly as before.
 * jet_order (B) | NULL: a permutation of the jets; workgroup i starts with jet jet_order[i].  A jet costs what its
 * working tiles cost, so longest-first (mmb_jet_schedule) keeps the last workgroups short; results do not depend on it.
 * 128 < N <= 160 (AOJ's 150-slot clouds): the jets of fewer than 128 particles run on four waves, two to a CU, their
 * slots beyond 128 -- padded ones -- carried as shadow slots; the others run in the five-wave kernel beside them, on a
 * side stream of the library that is fenced by events on `stream`.  long_jets (n_long) | NULL: the jets of 128 or more
 * particles and their number if the caller knows them (n_long = 0: none, one launch), n_long = -1 if it does not (every
 * jet then gets a five-wave workgroup that leaves at once when its jet is short: ~15 ms slower).  Ignored for N <= 128. */
int mmb_fused_epic_generate(int variant, int B, int N, const float* packed, const int32_t* desc,
                            int nb_branch, int nb_branch1, int nb_fused_exec, int use_skip,
                            int off_cw, int off_cb, int off_table,
                            const float* maskf, const int32_t* counts, float* x, int64_t* k,
                            int steps, const float* times, const float* temb_steps,
                            const float* ctab_steps, int n_ctab, float dt,
                            int thermo_kind, float gamma, const float* delta_n,
                            uint64_t seed, uint64_t offset,
                            int em, float em_sigma, const float* delta_w, const int32_t* jet_order,
                            const int32_t* long_jets, int n_long, void* stream);

/* rows[offs[b] + i] = b*N + (slot of the i-th real particle of jet b): the flat rows of the real particles of a batch,
 * ascending; offs (B) = exclusive prefix sum of the per-jet particle counts.  The reduction range of the per-particle
 * weight-gradient problems of mmb_grad_weight_rows_multi (GwProb.rows / n_rows): the reference's autograd sums over
 * every slot, but the rows of padded slots carry gradients that are exactly zero (masked losses). */
int mmb_mask_rows(int B, int N, const float* maskf, const int32_t* offs, int32_t* rows, void* stream);

/* Launch order for the generation loops: order (B) = the jets sorted by particle count (counts, as passed to the
 * generation call), descending.  No counterpart in the reference (its loop runs whole batches through dense layers);
 * one workgroup, a counting sort. */
int mmb_jet_schedule(int B, const int32_t* counts, int32_t* order, void* stream);

/* UniModalEPiC (discrete-only model): MultiModalBridgeMatching.forward with
 * encoders/multimodal_epic.py:44-64 (+ embedder.py:74-80, epic.py:71-98) for one jet per workgroup.
 * The projection sees cat[t, t, trunc(k_emb*mask)]; both time blocks are summed on the host.
 * variant 0: Et128 Ed128 H128 G16 (configs/S2_discrete.yaml); 1: Et16 Ed16 H32 G8; 2: Et64 Ed64 H64 G16;
 * 3: Et32 Ed32 H32 G16.
 * N <= 160, V = 8.  head_d: (B,N,8) logits. */
int mmb_fused_uni_forward(int variant, int B, int N, const float* packed, const int32_t* desc,
                          int num_blocks, int use_skip, int off_table,
                          const float* maskf, const int32_t* counts, const int64_t* k,
                          const float* temb, int64_t temb_stride, float* head_d, void* stream);

/* simulate_dynamics (multimodal_bridge_matching.py:227-259) of the discrete-only model in ONE
 * launch: per grid point forward -> telegraph rates -> tau-leap jump; final k = argmax of the last
 * rates.  Arguments as mmb_fused_epic_generate. */
int mmb_fused_uni_generate(int variant, int B, int N, const float* packed, const int32_t* desc,
                           int num_blocks, int use_skip, int off_table,
                           const float* maskf, const int32_t* counts, int64_t* k,
                           int steps, const float* times, const float* temb_steps,
                           const float* ctab_steps, int n_ctab, float dt,
                           int thermo_kind, float gamma, const float* delta_n,
                           uint64_t seed, uint64_t offset, const int32_t* jet_order,
                           const int32_t* long_jets, int n_long, void* stream);

/* ---------------------------------------------------------------- toy config (BASELINE configs[0]) */
/* Input rows of the toy MLP: rows[p,:] = [t(jet of p), x[p,0:Dc], table[k[p],0:E]]  (P, 1+Dc+E) --
 * MultiModalMLP.forward's torch.cat([time, continuous, embed(discrete)]) (notebooks/data_toys.ipynb cell 3).
 * t: one value per jet (t_stride 1) or one scalar for all (t_stride 0); table (V,E) is nn.Embedding.weight. */
int mmb_toy_rows(int64_t P, int N, int Dc, int E, int V, const float* t, int64_t t_stride, const float* x,
                 const int64_t* k, const float* table, float* rows, void* stream);
/* autograd of that nn.Embedding: d_table[k[p], e] += d_rows[p, col0+e]  (d_table accumulates). */
int mmb_toy_embedding_bwd(int64_t P, int E, int V, const int64_t* k, const float* d_rows, int64_t ld, int col0,
                          float* d_table, void* stream);
/* Toy distributions on device (distributions/toys.py): kind 0 = NGaussians.__call__ (:15-43; point i in gaussian
 * i % V, centres on the radius-5 circle, covariance sqrt(0.1) I), kind 1 = TwoMoons.__call__ (:49-55;
 * generate_moons(n, noise) * 3 - 1, labels {0,1}).  x (n,2) f32, k (n) int64; the permutation is the caller's. */
int mmb_toy_sample(int kind, int64_t n, int V, float noise, uint64_t seed, uint64_t offset, float* x, int64_t* k,
                   void* stream);

/* ---------------------------------------------------------------- AspenOpenJets reader */
/* AspenOpenJets._read_aoj_file's arithmetic (datamodules/aoj.py:147-245) on a raw PFCands block (J,M,F) f32, pid at
 * column F-2: rows with |pid| < 11 are zeroed (_filter_particles :179-184); feats (J,M,3) = (pt, eta_rel, phi_rel)
 * relative to the jet axis (the summed four-vector), times mask (_compute_continuous_coordinates :202-233);
 * tokens (J,M) = the 8-token flavour x charge map, unknown pids -> 0 (_map_pid_to_tokens :186-200);
 * mask (J,M) = (E > 0) (:222). */
int mmb_aoj_features(int J, int M, int F, const float* pfcands, float* feats, int64_t* tokens, int64_t* mask,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMB_HIP_H */
