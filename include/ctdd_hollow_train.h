/* ctdd_hollow_train.h -- C ABI of the training-side kernels of the SDDM hollow transformer (csrc/hollow_train_kernels.hip).
 *
 * They replace what `l.backward()` (TAUnSDDM/lib/training/training.py:27) runs through
 * TAUnSDDM/lib/networks/hollow_networks.py: LayerNorm / FiLM (311-447, 90-132), the masked attentions (204-280, 534-560),
 * ReLU / GELU, the input embedding (729-753) -- plus the training-mode forward pieces the inference kernels of
 * ctdd_hollow.h lack (attention dropout inside nn.MultiheadAttention, dropout after activations).  The linear layers'
 * gradients run on ctdd_unet_conv* (data gradient = a GEMM with the transposed weight) and ctdd_unet_wgrad (kind 1x1).
 * fp32 device buffers, caller-owned; status 0 / negative CTDD_E*.  Dropout masks: Philox(seed, step * 4096 + layer, element)
 * with {seed, step} read from device memory `rng`, so forward and backward of one step see the same mask.
 */
#ifndef CTDD_HOLLOW_TRAIN_H
#define CTDD_HOLLOW_TRAIN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* backward of ctdd_hollow_layernorm: out = FiLM_b(LayerNorm(x (+ y))).  dx (and dy) (+)= ...; dgamma / dbeta [E] and
 * dfilm [B][2E] are ATOMICALLY ACCUMULATED (zero them first).  E <= 512.  rpw: rows per wave, 0 = the launcher's choice (one
 * pass of the kernel's row groups).  Widths of whole 16-byte vectors per lane (E = 128, 256, 512 with 16-byte aligned rows) run
 * a vector kernel: a lane owns four consecutive columns, E / 4 lanes a row. */
typedef struct {
  const float* x; const float* y; int64_t x_bs, y_bs; const float* gamma; const float* beta; float eps;
  const float* film; int film_stride; const float* dout; int64_t dout_bs; int B, T, E, rpw;
  float* dx; int64_t dx_bs; int acc_dx; float* dy; int64_t dy_bs; int acc_dy; float* dgamma; float* dbeta; float* dfilm;
  int nrep, rep_stride;   /* nrep > 1: workgroup w accumulates into dgamma / dbeta + (w % nrep) * rep_stride; the caller sums the replicas
                             (thousands of workgroups on the same 2E addresses serialise in L2) */
  const float* dres; int64_t dres_bs;   /* optional: dx = dres + gradient, out of place (the residual stream's incoming gradient) */
} ctdd_hollow_ln_bwd_args;
int ctdd_hollow_layernorm_bwd(const void* ln_bwd_args, void* stream);

/* masked multi-head attention in training mode (masks / strides as ctdd_hollow_attention; head dimension 4, 8, 16, 32):
 * forward with dropout on the probabilities (nn.MultiheadAttention's attention dropout), `stats` [B][H][Tq][4] receives the
 * softmax row maximum and sum; backward (two launches, no atomics: dQ per query, dK / dV per key, scores recomputed). */
typedef struct {
  const float* q; const float* k; const float* v; int64_t q_bs, k_bs, v_bs; int q_rs, k_rs, v_rs;
  int B, Tq, Tk, H, hd, mode; float scale; float* out; int out_rs; float* stats;
  float drop_p; const uint64_t* rng; uint64_t layer;
  const float* d_out; float* dq; float* dk; float* dv; int64_t dq_bs, dk_bs, dv_bs; int dq_rs, dk_rs, dv_rs;
  void* out_bf16; void* dq_bf16; void* dk_bf16; void* dv_bf16;   /* optional bf16 copies (the *_bf16 entry points), strides of the fp32 ones */
} ctdd_hollow_attn_train_args;
int ctdd_hollow_attention_train(const void* attn_train_args, void* stream);
int ctdd_hollow_attention_bwd(const void* attn_train_args, void* stream);
/* the same two on v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 softmax / dropout / accumulation; head dimension 16 or 32;
 * rows 16-byte aligned): identical Philox masks; out / dq / dk / dv may also (or only, for the gradients) be written as bf16 */
int ctdd_hollow_attention_train_bf16(const void* attn_train_args, void* stream);
int ctdd_hollow_attention_bwd_bf16(const void* attn_train_args, void* stream);

/* dout == NULL: out = dropout(act(pre)) (+ bf16 copy);  dout != NULL: out = dropout(dout) * act'(pre); out or out_bf16 may be NULL.
 * act 0 identity, 1 ReLU, 2 GELU (erf); n % 4 == 0 */
int ctdd_hollow_act(const float* pre, const float* dout, float* out, void* out_bf16, int64_t n, int act, float drop_p,
                    const uint64_t* rng, uint64_t layer, void* stream);

/* out = dropout(x) (+ res) as fp32 and / or bf16 (n % 4 == 0; p = 0: add / cast only); masks as ctdd_hollow_act's */
int ctdd_hollow_dropout(const float* x, const float* res, float* out, void* out_bf16, int64_t n, float drop_p, const uint64_t* rng,
                        uint64_t layer, void* stream);

/* column sums of a (rows, N) matrix (bias gradients), first stage: partial[blk % nrep][n] (+)= the sum over workgroup blk's run
 * of rows (nblk workgroups; N % 8 == 0, N <= 2048); nrep < nblk: `partial` (nrep, N) must be zeroed, nblk / nrep workgroups
 * meet per address in float atomics; the caller adds the nrep partials (ctdd_unet_sum_batch). */
int ctdd_hollow_colsum(const float* x_f32, const void* x_bf16, int64_t rows, int N, int ld, float* partial, int nblk, int nrep,
                       void* stream);

/* bf16-only ReLU (+ dropout) of the MLP hidden tensor (bf16 mode), n % 8 == 0:
 * mask_u == NULL: out = dropout(relu(src))  (Philox masks as ctdd_hollow_act on the same element indices; in place allowed);
 * mask_u != NULL: out = src * [mask_u != 0] / (1 - p)  -- the backward, with the saved forward output as the mask. */
int ctdd_hollow_relu_bf16(const void* src, const void* mask_u, void* out, int64_t n, float drop_p, const uint64_t* rng,
                          uint64_t layer, void* stream);

/* gradient of the input embedding Linear(1 -> E) (hollow_networks.py:740-742): dw, db [E] atomically accumulated */
typedef struct { const int64_t* x64; const int32_t* x32; const float* dl2r; const float* dr2l; int B, D, E, S; float* dw; float* db; } ctdd_hollow_embed_bwd_args;
int ctdd_hollow_embed_bwd(const void* embed_bwd_args, void* stream);

#ifdef __cplusplus
}
#endif
#endif
