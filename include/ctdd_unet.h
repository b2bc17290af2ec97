/* ctdd_unet.h -- C ABI of the U-Net inference kernels in libctdd.so (csrc/unet_kernels.hip).
 *
 * These entry points replace the forward pass of the reference's score network on the sampling
 * path: TAUnSDDM/lib/networks/unet.py:419-459 (UNet.forward) with its blocks (ResBlock 100-140,
 * Downsample/Upsample 79-97, SelfAttention/QKVAttention 152-200, TimeEmbedding 223-241) and the
 * logistic head of TAUnSDDM/lib/models/models.py:249-283.  Same conventions as ctdd.h: caller-
 * owned device pointers, hipStream_t as void*, 0 / negative status.  Argument blocks are plain C
 * structs passed by pointer (host memory); the Python host mirrors them with ctypes.Structure
 * (ctdd/unet_engine.py).  All activations are NHWC.
 */
#ifndef CTDD_UNET_H
#define CTDD_UNET_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* K-segment kinds of the implicit-GEMM convolution */
#define CTDD_SEG_3x3 0     /* 3x3, stride 1, pad 1                                   (unet.py:41-61)  */
#define CTDD_SEG_1x1 1     /* 1x1: ResBlock linear skip / attention qkv, proj        (119-138,165-167)*/
#define CTDD_SEG_3x3_S2 2  /* 3x3, stride 2, input padded (0,1,0,1)                  (88-97)          */
#define CTDD_SEG_3x3_UP 3  /* 3x3 pad 1 on the nearest-2x upsampled input            (79-85)          */
#define CTDD_SEG_3x3_S2T 4 /* transpose of CTDD_SEG_3x3_S2: data gradient of the Downsample conv (backward of 88-97); the
                              output grid (H, W) is the forward's input grid, (Hin, Win) the forward's output grid;
                              ctdd_unet_conv (generic kernel) only */

typedef struct { const void* hi; const float* f32; int C, kind; } ctdd_conv_seg;   /* [B][Hin][Win][C] bf16 | fp32 */
typedef struct {
  ctdd_conv_seg seg[3]; int nseg;
  const void* w_hi; const float* w_f32;        /* [N][Ktot] bf16 | fp32, K = segment -> tap -> channel */
  int B, H, W, Hin, Win, N, Ktot;
  const float* bias; const float* tbias; int tb_stride;
  const float* res_f32; const void* res_bf16;
  float* out_f32; void* out_hi; double* stats; int logits_C;   /* stats: [B][N][2] fp64 sum / sum of squares */
  int ksplit; float* acc_buf;                                  /* split-K: zeroed [M][N] fp32 partial-sum buffer */
  int act;                                                     /* 0 none, 1 ReLU, 2 GELU(erf) on acc + bias, before the residual (not in res / ring) */
  void* out_lo;                                                /* ctdd_unet_conv_patch: optional bf16(out - out_hi), the second term of a split operand */
} ctdd_conv_args;
/* out = conv(segments) + bias + tbias[b] + residual; bk in {96,64,32,16}, bnt = N-tile/32,
 * f32 = 0: bf16 MFMA, 1: exact-fp32 MFMA */
int ctdd_unet_conv(const void* conv_args, int bk, int bnt, int f32, void* stream);
/* bf16 throughput kernel for stride-1 3x3 / 1x1 segments: the input slab of a pixel tile is staged in
 * LDS once per channel chunk and shared by the nine taps; bk in {48,64,32,16}, wm = rows per wave */
int ctdd_unet_conv_patch(const void* conv_args, int bk, int bnt, int wm, void* stream);
/* same contract, 512-pixel tiles with the weights of all nine taps of a 32-channel chunk resident in LDS
 * (no barrier between taps); segment channel counts must be multiples of 32; bnt in {2,3,4} */
int ctdd_unet_conv_res(const void* conv_args, int bnt, void* stream);
/* same contract, 16-channel units moved global -> LDS by LDS-DMA into a ring of unit buffers (no staging
 * registers, one barrier per unit); segment channel counts must be multiples of 16; bnt in {2,3,4}: 512-pixel
 * tiles, eight waves; bnt in {12,13}: N-tile 64/96 with 256-pixel tiles, four waves, two workgroups per CU */
int ctdd_unet_conv_ring(const void* conv_args, int bnt, void* stream);
int ctdd_unet_upsample2x(const void* x_bf16, int B, int H, int W, int C, void* out_bf16, void* stream);   /* unet.py:79-85 */

typedef struct {
  const int64_t* x64; const int32_t* x32; float lo, hi; const float* w; const float* bias;
  int B, Cin, H, W, Cout; float* out_f32; void* out_hi; double* stats; float* x0_f32;
} ctdd_first_conv_args;
/* center_data + first conv3x3 on the integer state (unet.py:428, network_utils.py:23-25) */
int ctdd_unet_first_conv(const void* first_conv_args, void* stream);

typedef struct {
  const float* s1_f32; const void* s1_bf16; const double* st1; int C1;
  const float* s2_f32; const void* s2_bf16; const double* st2; int C2;
  const float* gamma; const float* beta; int B, HW, G; float eps; int swish; void* out_hi; float* out_f32;
} ctdd_gn_args;
/* GroupNorm (+Swish) of the channel concatenation of one or two tensors (unet.py:103-133, 403-416) */
int ctdd_unet_gn_apply(const void* gn_args, void* stream);
/* The same normalisation of bf16 tensors in one pass over memory, statistics included (st1 / st2 are not read): a workgroup owns
 * (sample, slab of whole groups) and holds it in registers between the reduction and the write.  slab_channels: a multiple of
 * lcm(C / G, 8) dividing C, or 0 to let the library choose; max_threads: workgroup size limit (0: 1024); CTDD_ERANGE when no slab
 * fits (fall back to ctdd_unet_gn_apply). */
int ctdd_unet_gn_onepass(const void* gn_args, int slab_channels, int max_threads, void* stream);
int ctdd_unet_channel_stats(const float* x, int B, int HW, int C, double* stats, void* stream);

typedef struct {
  const float* t; int B, ch, tdim;
  const float* w1; const float* b1; const float* w2; const float* b2;   /* Linear weights transposed to [in][out] */
  float* hid; float* act;                                                /* [B][tdim] scratch, [B][tdim] = swish(temb) */
} ctdd_time_args;
/* sinusoid -> Linear -> Swish -> Linear -> Swish, then every ResBlock's time projection (unet.py:223-241,332-337,110);
 * proj_w = the concatenated projection weights transposed to [tdim][Ntot] */
int ctdd_unet_time(const void* time_args, const float* proj_w, const float* proj_b, int Ntot, float* proj_out, void* stream);
/* the same for ONE time shared by every sample (the samplers pass t * ones((N,)), sampling.py:119-121): t[0] only, one launch,
 * proj_out is a single row (Ntot) that the convolutions read with a zero batch stride */
int ctdd_unet_time_uniform(const void* time_args, const float* proj_w, const float* proj_b, int Ntot, float* proj_out, void* stream);

typedef struct { const float* qkv; int B, T, C, heads; void* out_hi; float* out_f32; } ctdd_attn_args;
int ctdd_unet_attention(const void* attn_args, void* stream);              /* unet.py:176-200 */

typedef struct { const float* net; const float* x0; int B, C, HW, S, fix; float* out;
                 int fast;   /* 1: hardware exp2/log2 forms (bf16 engine mode) */
                 void* out_bf16;   /* or null; fast mode, S % 4 == 0: the (B, D, S) logits in bf16 instead of `out` (the sampler
                                    * loops of the bf16 engine: half the bytes of the head's write and the step's read) */ } ctdd_logistic_args;
int ctdd_unet_logistic_head(const void* logistic_args, void* stream);      /* models.py:249-283 */

#ifdef __cplusplus
}
#endif
#endif
