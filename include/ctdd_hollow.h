/* ctdd_hollow.h -- C ABI of the hollow-transformer inference kernels (libctdd.so, gfx950).
 * Reference: TAUnSDDM/lib/networks/hollow_networks.py (BidirectionalTransformer2 668-755 and the blocks it
 * is built from), lib/models/models.py:495-525.  fp32 device buffers, caller-owned; every call enqueues on
 * `stream` (a hipStream_t) and returns 0 or a negative CTDD_E* code (ctdd_last_error() has the message).
 * The linear layers of the network run on ctdd_unet_conv (ctdd_unet.h) with a 1x1 segment per input. */
#ifndef CTDD_HOLLOW_H
#define CTDD_HOLLOW_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* token sequences of the two causal directions (hollow_networks.py:729-753, 534-563, 1136-1156):
 * temb = [sin | cos](t * temb_scale * f_j); x_embed = Linear(1->E)(2 x/(S-1) - 1);
 * l2r = [temb, x_0..x_{D-2}] + pe, r2l = [x_1..x_{D-1}, temb] + pe */
typedef struct {
  const int64_t* x64; const int32_t* x32; const float* t; const float* w_in; const float* b_in; const float* pe;
  int B, D, E, S; float temb_scale; float* l2r; float* r2l; float* temb;
} ctdd_hollow_embed_args;
int ctdd_hollow_embed(const void* embed_args, void* stream);

/* out[b][j] = FiLM_b(LayerNorm(x[b][j] (+ y[b][j]))) with batch strides (in floats) on every operand
 * (nn.LayerNorm in SelfAttentionBlock 311-340, FeedForwardBlock 343-420, AttentionReadout 283-308, ResidualReadout 90-132;
 * apply_film: a = film[b][0:E], b = film[b][E:2E]) */
typedef struct {
  const float* x; const float* y; int64_t x_bs, y_bs, out_bs; const float* gamma; const float* beta; float eps;
  const float* film; int film_stride; int B, T, E; float* out;
  void* out_bf16; int64_t out_bf16_bs;            /* optional bf16 copy of the output (GEMM operand of the bf16 mode) */
  void* out_lo;                                   /* optional second bf16 term bf16(out - out_bf16), stride out_bf16_bs: out ~ hi + lo
                                                   * to 2^-17 (operands of the split mode: three bf16 products per GEMM) */
} ctdd_hollow_ln_args;
int ctdd_hollow_layernorm(const void* ln_args, void* stream);

/* per-sample linears (time-embedding MLP 1136-1156 / 90-132, FiLM layers): out[b][n] = act(bias[n] + sum_k x[b][k] w[n][k]), fp32,
 * w (N, K) as the module stores it; act 0 none, 1 ReLU, 2 GELU (erf).  K % 64 == 0, K <= 1024. */
int ctdd_hollow_small_linear(const float* x, const float* w, const float* bias, int B, int K, int N, int act, float* out, void* stream);
int ctdd_hollow_add(const float* p, int64_t p_bs, const float* q, int64_t q_bs, float* out, void* out_bf16, void* out_lo, int64_t out_bs,
                    int B, int64_t per_batch, void* stream);                            /* l2r + r2l (fp32 and/or bf16 hi (+ lo) result) */
int ctdd_hollow_put_rows(const float* src, float* dst, void* dst_bf16, void* dst_lo, int64_t dst_bs, int B, int E,
                         void* stream);                                                 /* temb into key slot 0 */

/* every nn.Linear of the network (hollow_networks.py:90-447) in the bf16 modes, and its data gradient dX = dY W in training:
 * out[M][N] = act([A_0 | A_1 | A_2][M][nseg K] . W[N][nseg K]^T + bias) + res, fp32 accumulation on the matrix cores.
 * K % 64 == 0, N % 4 == 0; out_f32 / out_hi (bf16) / out_lo (bf16(v - hi)): any subset; act 0 none, 1 ReLU, 2 GELU (erf).
 * Three segments against a concatenated weight are the hi / lo split-precision product [x_hi | x_lo | x_hi].[w_hi | w_hi | w_lo]. */
typedef struct {
  const void* a[3]; int nseg; const void* w; const float* bias; const float* res;
  float* out_f32; void* out_hi; void* out_lo; int M, N, K, act;
  /* training epilogues (hollow_networks.py:343-420: the Dropout layers of the blocks), the masks of ctdd_hollow_dropout / _act:
   * drop_p > 0 with rng = {seed, step}: out = dropout(act(.)) + res, keep flags Philox(seed, step * 4096 + layer, element / 4);
   * mask_u ([M][N] bf16): out = mask_u != 0 ? . / (1 - drop_p) : 0 -- the ReLU + dropout backward, the saved output as the mask */
  float drop_p; const uint64_t* rng; uint64_t layer; const void* mask_u;
} ctdd_gemm_args;
int ctdd_gemm_bf16(const void* gemm_args, void* stream);

/* masked multi-head attention, softmax(scale q.k) v: mode 0 causal (j <= i, UniDirectionalTransformer l2r 534-560),
 * 1 anti-causal (j >= i, r2l), 2 readout over [temb | l2r | r2l] with Tk = 2 Tq + 1 (CrossAttention 204-280).
 * q/k/v rows at base + b*bs + row*rs + head*hd. */
typedef struct {
  const float* q; const float* k; const float* v; int64_t q_bs, k_bs, v_bs; int q_rs, k_rs, v_rs;
  int B, Tq, Tk, H, hd, mode; float scale; float* out; int out_rs;
  void* out_bf16;                                 /* optional bf16 copy of the output */
  void* out_lo;                                   /* optional second bf16 term of the output */
  int split;                                      /* ctdd_hollow_attention_bf16: 1 = q, k, v and the probabilities enter as hi + lo bf16
                                                   * pairs, three products per contraction (~1e-5 relative instead of ~4e-3) */
} ctdd_hollow_attn_args;
int ctdd_hollow_attention(const void* attn_args, void* stream);
/* same contract with both products on the bf16 matrix cores (fp32 softmax); head dimension 16 or 32 */
int ctdd_hollow_attention_bf16(const void* attn_args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTDD_HOLLOW_H */
