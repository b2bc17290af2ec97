/* ctdd.h -- C ABI of libctdd.so: the MI355X (gfx950) engine for the tauLDR / SDDM hot path.
 *
 * Drop-in boundary: the reference exposes this path through a Python registry
 * API (TAUnSDDM/lib/{models,losses,sampling,training}), not an FFI.  The host-side mirror of
 * that API lives in continuous-time-diffusion-models-for-discrete-data_amd/lib and binds the
 * entry points below with ctypes (INTEGRATION.md shows the stub).  Each entry point names the
 * reference code it replaces (paths relative to /root/reference/TAUnSDDM).
 *
 * Conventions
 *  - every pointer is a caller-owned DEVICE buffer (hipMalloc / torch CUDA storage) unless the
 *    name ends in _host; nothing is allocated, retained or freed by the library;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work;
 *  - return value: 0 on success, negative CTDD_E* on failure; ctdd_last_error() gives the
 *    message of the calling thread's last failure.  No exceptions cross the ABI;
 *  - layouts are dense row-major: logits (N,D,S) f32 with S contiguous, states (N,D) int32,
 *    tables (nT,S,S) f32;
 *  - randomness is explicit: either a noise tensor supplied by the caller, or a Philox4x32-10
 *    (seed, offset) pair; counter = (row_lo,row_hi,offset,draw), key = seed (csrc/philox.hpp).
 */
#ifndef CTDD_H
#define CTDD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTDD_OK 0
#define CTDD_EINVAL (-22)   /* bad argument (null pointer, size out of range, unknown enum) */
#define CTDD_ERANGE (-34)   /* size not supported by this build (e.g. S > CTDD_MAX_S)      */
#define CTDD_EHIP (-5)      /* HIP runtime error at launch                                  */

#define CTDD_MAX_S 1024

/* logit_type of the SDDM branch (lib/models/model_utils.py:30-60) */
#define CTDD_LOGIT_DIRECT 0
#define CTDD_LOGIT_REVERSE_PROB 1
#define CTDD_LOGIT_REVERSE_LOGSCALE 2

/* reverse-rate branch of get_reverse_rates (lib/sampling/sampling.py:31-78) */
#define CTDD_BRANCH_CTELBO 0   /* loss in {CTElbo, NLL, CTElboLambda}: lines 32-59 */
#define CTDD_BRANCH_CRM 1      /* everything else: lines 61-73                      */

/* flags of ctdd_tauleap_step */
#define CTDD_STEP_ORDINAL 1u     /* cfg.sampler.is_ordinal (sampling.py:135-138)                 */
#define CTDD_STEP_CORRECTOR 2u   /* add rate[x][s] to R^ (corrector, sampling.py:182-198)        */
#define CTDD_STEP_COUNT_RAW 4u   /* out_changed counts the UNCLIPPED move != 0 (sampling.py:505) */
#define CTDD_STEP_COUNT_JUMPS 16u /* out_changed is int32[3]: [1] += #dims with >= 1 jump event, [2] += #dims with > 1
                                    (MidPointTauL's change_jump = [2]/[1], sampling.py:489-495) */
#define CTDD_STEP_CRM 8u         /* ctdd_tauleap_step_s256 only: CRM-branch rates with logit_type reverse_prob (sampling.py:61-73);
                                  * the step tables must come from ctdd_s256_prepare_crm */
#define CTDD_STEP_BF16 32u       /* ctdd_tauleap_step_s256 only: ONE bf16 product for the S x S contraction (the mode the bf16 score
                                  * network runs with; all terms are >= 0, relative rate error <= 3 * 2^-8) instead of the three
                                  * split-bf16 products of the fp32 parity mode.  Same tables, same draw rule. */
#define CTDD_STEP_LOGITS_BF16 64u /* with CTDD_STEP_BF16: `logits` points at (N,D,256) bf16 values (what the bf16 U-Net engine's
                                  * output convolution writes on request) instead of f32 */

int ctdd_abi_version(void);
const char* ctdd_last_error(void);

/* K1  q_{t|0} and R_t tables for nT times.
 * Replaces {GaussianTargetRate,UniformRate,UniformVariantRate,BirthDeathForwardBase}.transition/
 * .rate/.transit_between (lib/models/forward_model.py:43-75,95-129,166-204,252-306):
 *   P_t = V diag(exp(integral[t] * eigvals)) W ; optional row-normalise ; entries < clamp_below -> 0
 *   R_t = beta[t] * base_rate
 * W = inv_eigvecs (or V^T).  Any output pointer may be NULL.  out_qt0T[t][x][s0] = P_t[s0][x];
 * out_noise_probs[t][x0][:] = the probabilities torch's Categorical(logits=where(row<=0,-1e9,
 * log row)) hands to multinomial (lib/losses/losses.py:46-55). */
int ctdd_rate_table(const float* eigvecs, const float* right, const float* eigvals,
                    const float* base_rate, const float* integral, const float* beta,
                    int nT, int S, int normalise, float clamp_below,
                    float* out_qt0, float* out_qt0T, float* out_rate, float* out_noise_probs,
                    void* stream);

/* K2  x_t ~ Categorical(probs[tidx[b]][x0[b,d]][:]) for every (b,d)  (lib/losses/losses.py:46-59,
 * 859-874, 1211-1225).  One draw per row as ATen does it: argmax_s(p_s / E_s), first index on
 * ties.  E (B*D,S) explicit Exp(1) noise, or NULL to draw E = -log(u) from Philox(seed,offset).
 * tidx may be NULL (table b for batch row b). */
int ctdd_noise_categorical(const float* probs, const int32_t* tidx, const int32_t* x0,
                           const float* E, uint64_t seed, uint64_t offset,
                           int B, int D, int S, int32_t* out_xt, void* stream);

/* K3  one-jump neighbour x~ of x_t (lib/losses/losses.py:61-101): per batch row pick the
 * dimension with weight w_d = sum_{s != x_d} rate[t][x_d][s] and the new value from row
 * rate[t][x_dim][:] (own state excluded), both by exponential race.  E_dim (B,D), E_val (B,S)
 * explicit, or both NULL for Philox.  rate is (nT,S,S) = R_t; tidx as above. */
int ctdd_xtilde_sample(const float* rate, const int32_t* tidx, const int32_t* x_t,
                       const float* E_dim, const float* E_val, uint64_t seed, uint64_t offset,
                       int B, int D, int S, int32_t* out_dims, int32_t* out_newval,
                       int32_t* out_xtilde, void* stream);

/* A6  SDDM log-probabilities (lib/models/model_utils.py:30-60): ll_all (N,D,S), ll_xt (N,D). */
int ctdd_logprob(const float* logits, const int32_t* x, const float* qt0, const int32_t* tidx,
                 int logit_type, int N, int D, int S, float* out_ll_all, float* out_ll_xt,
                 void* stream);

/* K4/K5  reverse rates R^ and ratio, own state NOT zeroed (lib/sampling/sampling.py:31-78).
 * qt0/rate are (nT,S,S); tidx[n] picks the table of batch row n (NULL: table 0 for all rows,
 * the sampling case where every row shares t).  out_ratio may be NULL. */
int ctdd_reverse_rates(int branch, int logit_type, const float* logits, const int32_t* x,
                       const float* qt0, const float* rate, const int32_t* tidx, float eps,
                       int N, int D, int S, float* out_rates, float* out_ratio, void* stream);

/* K6a  tau-leaping state update from explicit jump counts (lib/sampling/sampling.py:135-160,
 * 478-503): x_new = clamp(x + sum_s k_s (s - base), 0, S-1); base = x unless x_base given
 * (midpoint stage 2).  Non-ordinal: dimensions with more than one jump are left unchanged.
 * out_changed (1 int32, may be NULL) accumulates #(x_new != x). */
int ctdd_tauleap_apply(const int32_t* x, const int32_t* x_base, const float* jump_nums,
                       int is_ordinal, int N, int D, int S, int32_t* out_x, int32_t* out_changed,
                       void* stream);

/* K6b  Poisson jump draw + update from given rates (N,D,S).  The rates belong to the state
 * x_base (default x): its own entry is masked and moves are measured from it, then added to x
 * (midpoint stage 2: rates at x', x_new = clip(x + sum k_s (s - x'))).
 * Draw rule (distribution-equal to independent Poisson(rate_s*h) per s, csrc/draw.hpp):
 * K ~ Poisson(h * sum_s rate_s) then K destinations ~ Categorical(rate). */
int ctdd_tauleap_draw(const float* rates, const int32_t* x, const int32_t* x_base, float h,
                      uint32_t flags, uint64_t seed, uint64_t offset, int N, int D, int S,
                      int32_t* out_x, int32_t* out_changed, void* stream);

/* K4+K6 fused: one tau-leaping (or corrector) step straight from the logits
 * (TauL.sample body, lib/sampling/sampling.py:119-160 and 165-221; PCTauL 559-640;
 * MidPointTauL stage 2, 459-508).  R^ never touches HBM.  beta = scalar R_t = beta*base_rate.
 * logits must be the network output at x_base when x_base is given (else at x). */
int ctdd_tauleap_step(int branch, int logit_type, const float* logits, const int32_t* x,
                      const int32_t* x_base, const float* qt0, const float* base_rate, float beta,
                      float eps, float h, uint32_t flags, uint64_t seed, uint64_t offset,
                      int N, int D, int S, int32_t* out_x, int32_t* out_changed, void* stream);

/* K7  Euler / LBJF step (lib/sampling/sampling.py:278-293, corrector 296-341): posterior row
 * P = h*R^*(1-onehot) + clip(1-h*sum,0)*onehot, normalised, then x_new ~ Categorical(log(P+1e-35))
 * by exponential race.  E (N*D,S) explicit or NULL for Philox.  out_probs (N,D,S) may be NULL. */
int ctdd_lbjf_step(int branch, int logit_type, const float* logits, const int32_t* x,
                   const float* qt0, const float* base_rate, float beta, float eps, float h,
                   uint32_t flags, const float* E, uint64_t seed, uint64_t offset,
                   int N, int D, int S, int32_t* out_x, float* out_probs, int32_t* out_changed,
                   void* stream);

/* ExactSampling step (lib/sampling/sampling.py:975-1061): per dimension
 * post[s] = (sum_x0 softmax(logits)[x0] q_lo[x0][s]) * q_step[s][x],  q_lo = q_{t-h|0}, q_step = q_{t|t-h}  (S,S each),
 * x_new ~ Categorical(post / sum post) by the exponential race of K7 (E (N*D,S) explicit, or NULL: Philox(seed, offset, row, s)).
 * out_probs (N,D,S) and out_changed (count of dimensions that moved) may be NULL. */
int ctdd_exact_step(const float* logits, const int32_t* x, const float* q_lo, const float* q_step, const float* E,
                    uint64_t seed, uint64_t offset, int N, int D, int S, int32_t* out_x, float* out_probs,
                    int32_t* out_changed, void* stream);

/* K8  midpoint predictor (lib/sampling/sampling.py:417-453):
 * x' = clip(x + round_half_even(0.5*h*sum_s R^_s (s-x)), 0, S-1) with own state excluded. */
int ctdd_midpoint_predict(int branch, int logit_type, const float* logits, const int32_t* x,
                          const float* qt0, const float* base_rate, float beta, float eps, float h,
                          int N, int D, int S, int32_t* out_x, void* stream);

/* K7 / K8 on reverse rates that are already computed and masked -- rates (N,D,S) as written by ctdd_tauleap_step_s256's
 * out_rates (corrector term included there when asked for): the LBJF posterior + categorical draw (sampling.py:278-293)
 * and the midpoint predictor (417-453).  The S = 256 Euler / midpoint samplers take their S x S contraction from the
 * matrix-core kernel this way instead of the generic fp32 one. */
int ctdd_lbjf_from_rates(const float* rates, const int32_t* x, float h, const float* E, uint64_t seed, uint64_t offset,
                         int N, int D, int S, int32_t* out_x, float* out_probs, int32_t* out_changed, void* stream);
int ctdd_midpoint_from_rates(const float* rates, const int32_t* x, float h, int N, int D, int S, int32_t* out_x, void* stream);

/* K10 final denoise: argmax_s softmax(logits) = first argmax of logits (sampling.py:223-229). */
int ctdd_argmax(const float* logits, int N, int D, int S, int32_t* out_x, void* stream);

/* A7  initial state (lib/sampling/sampling.py:14-28): uniform randint or inverse-CDF draws from
 * a host-computed pmf cdf (S floats, device).  cdf NULL = uniform. */
int ctdd_initial_samples(const float* cdf, uint64_t seed, uint64_t offset, int N, int D, int S,
                         int32_t* out_x, void* stream);

/* ---- S = 256 fast path (csrc/steps_s256.hip): the CT-ELBO-branch tau-leaping step on the matrix
 * cores.  The contraction runs as split-precision bf16 MFMA (hi*hi + hi*lo + lo*hi, fp32
 * accumulate; relative error of the non-negative sums < 3e-5, inside the 1e-4 parity bar).
 * ctdd_s256_prepare builds, for nT steps, the per-step derived tables the kernel streams
 * (1/(qt0[s0][x]+eps) and the bf16 hi/lo MFMA image of qt0) into caller-owned memory of
 * nT * ctdd_s256_step_table_bytes() bytes, and the two per-model views of the base rate:
 * RT0[x][s] = R[s][x], R0[x][s] = R[x][s], both with a zero diagonal ((256,256) f32 each). */
int64_t ctdd_s256_step_table_bytes(void);
int ctdd_s256_prepare(const float* qt0, const float* base_rate, float eps, int nT,
                      void* out_step_tables, float* out_RT0, float* out_R0, void* stream);
/* CRM branch, logit_type reverse_prob (sampling.py:61-73 with model_utils.py:40-47): ratio = exp(ll_all - ll_xt) with
 * ll_all = log(softmax(logits) @ qt0 + 1e-35), rates = ratio * beta * R[x][s].  The same contraction with a unit left
 * scaling: tables from ctdd_s256_prepare_crm, step call with CTDD_STEP_CRM in flags. */
int ctdd_s256_prepare_crm(const float* qt0, const float* base_rate, int nT, void* out_step_tables, float* out_RT0,
                          float* out_R0, void* stream);
/* Same semantics as ctdd_tauleap_step(branch = CTELBO) for S = 256 (lib/sampling/sampling.py:
 * 119-160, 165-221, 459-508).  step_tables points at THIS step's block.  out_rates (N,D,256),
 * optional, receives the masked reverse rates (validation / unfused use); out_x may then be NULL. */
int ctdd_tauleap_step_s256(const void* logits /* f32; bf16 with CTDD_STEP_LOGITS_BF16 */, const int32_t* x, const int32_t* x_base,
                           const void* step_tables, const float* RT0, const float* R0, float beta,
                           float h, uint32_t flags, uint64_t seed, uint64_t offset, int N, int D,
                           float* out_rates, int32_t* out_x, int32_t* out_changed, void* stream);

/* test hook: the raw uniforms a kernel would see: out[row*4*nblk + 4*j + i]. */
int ctdd_philox_uniform(uint64_t seed, uint64_t offset, int64_t nrows, int nblk, float* out,
                        void* stream);

/* ---- K12: categorical ratio matching objectives with the 'direct' logit type, value and d/dlogits in one pass
 * (lib/losses/losses.py: CatRM._comp_loss 794-836, CatRM.calc_loss 838-890, CatRMNLL 1146-1242;
 * model_utils.py:30-38).  loss_type 0 'rm', 1 'mle', 2 'elbo' (needs qt0 (B,S,S)).
 * out_loss = scale * sum_{b,d} loss_bd + nll_scale * sum_{b,d} -log_softmax(logits)[x0]   (x0 null: no CE term)
 * grad_logits (B,D,S) = d out_loss / d logits.  row_scratch: B*D doubles. */
int ctdd_crm_loss(const float* logits, const int32_t* xt, const int32_t* x0, const float* qt0, int B, int D, int S,
                  int loss_type, float scale, float nll_scale, float* grad_logits, double* row_scratch,
                  float* out_loss, void* stream);

/* ---- K11: tauLDR CT-ELBO with one_forward_pass (logits = model(x_t), reg_x = x~), value and d/dlogits
 * (lib/losses/losses.py:106-286; NLL 1598-1778 and CTElboLambda 1879-2058 share the body).
 * out_loss = elbo_scale * (mean_b(-sig_b / norm_b) + mean_b(reg_b)) + nll_scale * sum_{b,d} -log_softmax(logits)[x0]
 * (CTElbo: elbo_scale 1, nll_scale nll_weight/(B*D); CTElboLambda: w and (1-w)/(B*D)).  qt0, qt0T, rate: (B,S,S) per-sample
 * tables (K1); S <= 256.  scratch: ctdd_ctelbo_scratch_bytes(B,D,S) bytes. */
int64_t ctdd_ctelbo_scratch_bytes(int B, int D, int S);
int ctdd_ctelbo_loss(const float* logits, const int32_t* x0, const int32_t* x_tilde, const float* qt0, const float* qt0T,
                     const float* rate, int B, int D, int S, float eps, float elbo_scale, float nll_scale,
                     void* scratch, float* grad_logits, float* out_loss, void* stream);
/* The same kernels with the two ELBO terms weighted separately -- the building block of one_forward_pass = False
 * (lib/losses/losses.py:150-158: p0t_reg = softmax(model(x_t)) with reg_x = x_t, p0t_sig = softmax(model(x~))):
 *   out_loss = sig_scale * mean_b(-sig_b / norm_b) + reg_scale * mean_b(reg_b) + nll_scale * sum_{b,d} -log_softmax(logits)[x0]
 * with every term evaluated at the state passed as x_tilde.  The two-pass objective is
 *   terms(model(x_t), x_tilde := x_t, sig 0, reg w, nll) + terms(model(x~), x_tilde := x~, sig w, reg 0, nll 0). */
int ctdd_ctelbo_loss_terms(const float* logits, const int32_t* x0, const int32_t* x_tilde, const float* qt0, const float* qt0T,
                           const float* rate, int B, int D, int S, float eps, float sig_scale, float reg_scale, float nll_scale,
                           void* scratch, float* grad_logits, float* out_loss, void* stream);

/* ---- ScoreElbo with direct logits (lib/losses/losses.py:1255-1500), value and d/dlogits:
 * out_loss = mean_b(-sig_b / norm_b) + mean_b(reg_b) + nll_scale * sum_{b,d} -log_softmax(logits)[x~]   (nll_scale = nll_weight / B).
 * reg_x = x~ with one forward pass, x_t with two (logits = model(reg_x)).  scratch: ctdd_ctelbo_scratch_bytes(B,D,S). */
int ctdd_score_elbo_loss(const float* logits, const int32_t* x0, const int32_t* x_tilde, const int32_t* reg_x,
                         const float* qt0, const float* rate, int B, int D, int S, float eps, float nll_scale,
                         void* scratch, float* grad_logits, float* out_loss, void* stream);

/* The same two for logit_type reverse_prob with per-sample tables and S % 32 == 0 (<= 256), the S x S contractions on the
 * exact-fp32 matrix instruction (model_utils.py:42-48 and its autograd backward):
 * forward  ll_all = log(softmax(logits) @ q_{t|0} + 1e-35), ll_xt = ll_all[x]   (qt0T (B,S,S) = the transposed tables; scratch (B,D,S) fp32);
 * backward d/dlogits given ll_all (the forward's output) and dll = d loss / d ll_all, + the cross-entropy term as ctdd_logprob_bwd
 *          (scratch 2 x (B,D,S) fp32). */
int ctdd_logprob_rp_mfma(const float* logits, const int32_t* x, const float* qt0T, int B, int D, int S, float* scratch,
                         float* out_ll_all, float* out_ll_xt, void* stream);
int ctdd_logprob_rp_bwd_mfma(const float* logits, const float* qt0, const float* ll_all, const float* dll, const int32_t* x0,
                             float nll_scale, int B, int D, int S, float* scratch, float* grad_logits, double* ce_rows,
                             float* out_ce, void* stream);

/* ---- the same objectives for logit_type 'reverse_prob' / 'reverse_logscale' (model_utils.py:42-56; every shipped hollow
 * config uses reverse_prob): three launches chained on the device --
 *   ctdd_logprob (above)            logits -> ll_all = log p_t(. | x^{\d})
 *   ctdd_crm_loss_ll / ctdd_score_elbo_loss_ll   the objective on ll_all: value and grad_ll = d out_loss / d ll_all
 *   ctdd_logprob_bwd                grad_ll -> d/dlogits through ll = log(softmax(logits) @ q_{t|0} (+1e-35)); x0 non-null adds
 *                                   CatRMNLL's cross-entropy term nll_scale * sum -log_softmax(logits)[x0] (losses.py:1240-1242):
 *                                   its gradient to grad_logits, its value to out_ce (ce_rows: B*D doubles).  S <= 256. */
int ctdd_crm_loss_ll(const float* ll_all, const int32_t* xt, const float* qt0, int B, int D, int S, int loss_type, float scale,
                     float* grad_ll, double* row_scratch, float* out_loss, void* stream);
int ctdd_score_elbo_loss_ll(const float* ll_all, const int32_t* x0, const int32_t* x_tilde, const int32_t* reg_x,
                            const float* qt0, const float* rate, int B, int D, int S, float eps, float nll_scale,
                            void* scratch, float* grad_ll, float* out_loss, void* stream);
int ctdd_logprob_bwd(int logit_type, const float* logits, const float* qt0, const float* qt0T, const float* dll,
                     const int32_t* x0, float nll_scale, int B, int D, int S, float* grad_logits, double* ce_rows,
                     float* out_ce, void* stream);

/* ---- K28: clip_grad_norm_ + Adam.step + EMA update over all parameter tensors in two launches
 * (lib/training/training.py:17-40, lib/models/models.py:745-758, torch.optim.Adam single-tensor formulas).
 * tensors: device array of ctdd_opt_tensor; chunks: device array of ctdd_opt_chunk covering every tensor in
 * pieces of ctdd_opt_chunk_elems() elements.  max_norm <= 0: no clipping; ema_decay < 0: no EMA; a null
 * shadow pointer skips the EMA of that tensor.  Gradients are read, not rescaled in place. */
typedef struct { float* p; const float* g; float* m; float* v; float* shadow; int64_t n; } ctdd_opt_tensor;
typedef struct { int tensor; int pad; int64_t start; } ctdd_opt_chunk;
int ctdd_opt_chunk_elems(void);
int ctdd_adam_ema_step(const void* tensors, const void* chunks, int nchunks, float lr, float beta1, float beta2,
                       float eps, int64_t step, float max_norm, float ema_decay, double* sumsq_scratch,
                       void* stream);
/* The two launches on their own, for parameters that fall into several tables (param groups with different
 * hyper-parameters; tensors whose Adam step counts differ -- torch.optim.Adam keeps `step` per parameter): the
 * squared gradient norm of every table summed into ONE scalar (clip_grad_norm_ over model.parameters(),
 * training.py:28-29), then one update per table with its own step / lr. */
int ctdd_grad_sumsq(const void* tensors, const void* chunks, int nchunks, double* sumsq_scratch, int zero_first, void* stream);
int ctdd_adam_ema_apply(const void* tensors, const void* chunks, int nchunks, float lr, float beta1, float beta2,
                        float eps, int64_t step, float max_norm, float ema_decay, const double* sumsq, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTDD_H */
