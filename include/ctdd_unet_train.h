/* ctdd_unet_train.h -- C ABI of the U-Net BACKWARD kernels in libctdd.so (csrc/unet_train_kernels.hip).
 *
 * These entry points replace what `l.backward()` (TAUnSDDM/lib/training/training.py:27) makes autograd run for
 * the score network TAUnSDDM/lib/networks/unet.py: the data and weight gradients of every convolution
 * (unet.py:41-61, 79-97, 100-140, 343, 403-416), GroupNorm + Swish + Dropout (103-133), the mid-block
 * self-attention (152-200) and the bias / time-projection reductions (110, 131).  The forward pass of a training
 * step runs on the same kernels as sampling (ctdd_unet.h), keeping its intermediate tensors.
 * Conventions as in ctdd.h / ctdd_unet.h: caller-owned device pointers, hipStream_t as void*, 0 / negative status,
 * argument blocks are plain C structs in host memory.  Tensors are NHWC, bf16 (`*_bf16`, unsigned short bits) or
 * fp32 (`*_f32`): exactly one of each pair is non-null.
 *
 * Data gradients need no entry point of their own: they are convolutions of the output gradient with the
 * tap-flipped, transposed weights that ctdd_unet_pack_weights writes, run by ctdd_unet_conv* (ctdd_unet.h);
 * the Downsample conv's transpose is segment kind CTDD_SEG_3x3_S2T.
 */
#ifndef CTDD_UNET_TRAIN_H
#define CTDD_UNET_TRAIN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CTDD_WG_3x3 0     /* 3x3, stride 1, pad 1   (unet.py:41-61)                            */
#define CTDD_WG_1x1 1     /* 1x1 / Linear on pixels (unet.py:119-138, 165-167)                 */
#define CTDD_WG_3x3_S2 2  /* 3x3, stride 2, input padded (0,1,0,1) (unet.py:88-97)             */

/* weight gradient of one K-segment of a convolution: gw[n][koff + tap*C + c] += sum_pixels dy[p][n] x[p + tap][c] */
typedef struct {
  const void* x;      /* [B][Hin][Win][C]  the segment's input activations                       */
  const void* dy;     /* [B][H][W][ldy]    gradient of the conv output; channels >= N are zeros  */
  float* gw;          /* [N][Ktot] fp32 packed gradient (same K order as the forward weights), atomically accumulated */
  int B, H, W, Hin, Win, N, ldy, C, Ktot, koff;
  int kind;           /* CTDD_WG_*                                                                */
  int nlr;            /* CTDD_WG_3x3: extended image rows per chunk; otherwise pixels per chunk (multiple of 16) */
  int nwn;            /* waves along N: 1, 2 or 4 (4 / nwn along C)                               */
  int nchunks;        /* CTDD_WG_3x3: ceil(B (H+1) / nlr); otherwise ceil(B H W / nlr)            */
  int grid_x;         /* workgroups that share the chunks of this entry (M-split)                 */
  int tap;            /* CTDD_WG_3x3_S2: the tap (0..8) this entry computes: nine entries per such convolution */
  float* gb;          /* [N] fp32 or null: += sum over all pixels of dy[p][n] (the bias gradient; one entry per convolution
                       * carries it -- of a stride-2 convolution's nine, the tap-0 entry), atomically accumulated   */
} ctdd_wgrad_args;
/* ONE launch for a table of n entries (blockIdx.z = entry): the backward plan defers all weight gradients to its end so
 * that equal-shaped convolutions fill the chip together with few M-split workgroups each (those meet in float atomics).
 * table_dev: the entries in device memory; table_host: the same entries in host memory (validated, sized from). */
int ctdd_unet_wgrad(const void* table_dev, const void* table_host, int n, int f32, void* stream);

/* GroupNorm (+Swish, +Dropout) backward over the channel concatenation of one or two tensors (unet.py:103-133): two
 * launches over (sample, pixel slice) workgroups -- per-(sample, channel) sums of dz and dz*xhat into `sums` (zeroed by
 * the caller), then dX into d1 / d2 (acc != 0: added to what is there).  dgamma / dbeta = sums over the samples of `sums`. */
typedef struct {
  const float* s1_f32; const void* s1_bf16; const double* st1; int C1;   /* forward inputs + their [B][C][2] fp64 statistics */
  const float* s2_f32; const void* s2_bf16; const double* st2; int C2;
  const float* gamma; const float* beta;
  int B, HW, G; float eps; int swish;
  const float* da_f32; const void* da_bf16;                              /* gradient w.r.t. the activated output */
  float* sums;                                                            /* [B][C1+C2][2] fp32                   */
  float* d1_f32; void* d1_bf16; int acc1;
  float* d2_f32; void* d2_bf16; int acc2;
  float drop_p; const uint64_t* rng; uint64_t layer;                      /* dropout applied after the activation (0: none); rng = device
                                                                             {seed, step}: the mask of element e is Philox(seed, step*4096 + layer, e) */
  float* dsum_bn; int dsum_stride; float* dsum_n;                         /* optional, single source: sum_p dX[b,p,c] -> dsum_bn[b*stride + c] and
                                                                             += into dsum_n[c]: the time-projection / conv1-bias gradients (unet.py:110,131) */
} ctdd_gn_bwd_args;
int ctdd_unet_gn_bwd(const void* gn_bwd_args, void* stream);
/* the forward side of that dropout, in place on the activated tensor: a *= keep / (1 - p); same mask rule */
int ctdd_unet_dropout(float* a_f32, void* a_bf16, int64_t n, float p, const uint64_t* rng, uint64_t layer, void* stream);

/* out_bn[b*stride + n] += sum_p g[b,p,n]  and/or  out_n[n] += sum_{b,p} g[b,p,n]   (time-projection / bias gradients) */
int ctdd_unet_colsum(const float* g_f32, const void* g_bf16, int B, int HW, int N, int ld, float* out_bn, int stride, float* out_n,
                     void* stream);
/* out[j] (+)= sum_b in[b*bstride + j*jstride] */
int ctdd_unet_sum_batch(const float* in, int B, int64_t bstride, int jstride, int n, float* out, int accumulate, void* stream);
/* the same for a table of jobs in ONE launch: job = outputs j0 .. j0+63 of out[j] (+)= sum_b in[b*bstride + j*jstride], j < n */
typedef struct { const float* in; float* out; int64_t bstride; int B, jstride, n, accumulate, j0, pad_; } ctdd_sum_job;
int ctdd_unet_sum_jobs(const void* jobs_dev, int njobs, void* stream);
/* dst (+)= src, n elements (n % 8 == 0): identity-skip / residual branch of a gradient */
int ctdd_unet_accumulate(const float* src_f32, const void* src_bf16, float* dst_f32, void* dst_bf16, int64_t n, int accumulate, void* stream);
/* backward of nearest-2x upsampling (unet.py:79-85): out[b,y,x,c] (+)= sum of the 2x2 block of up */
int ctdd_unet_downsum2x(const float* up_f32, const void* up_bf16, int B, int H, int W, int C, float* out_f32, void* out_bf16,
                        int accumulate, void* stream);
int ctdd_unet_upsample2x_f32(const float* x, int B, int H, int W, int C, float* out, void* stream);
/* rows of `n` fp32 values (row stride ld_in) -> bf16 or fp32 rows of ld_out with zero padding */
int ctdd_unet_cast_rows(const float* in, int64_t rows, int n, int ld_in, int ld_out, void* out_bf16, float* out_f32, void* stream);

/* mid-block attention backward (unet.py:176-200): d_out [B][T][C] -> d_qkv [B][T][3C] (fp32 and/or bf16) */
typedef struct { const float* qkv; const float* d_out_f32; const void* d_out_bf16; int B, T, C, heads; float* d_qkv; void* d_qkv_bf16; } ctdd_attn_bwd_args;
int ctdd_unet_attention_bwd(const void* attn_bwd_args, void* stream);

/* weight (torch layout [Cout][Cin][3][3]) and bias gradient of the first conv on the centred integer state (unet.py:343) */
typedef struct { const int64_t* x64; const int32_t* x32; float lo, hi; const float* dy_f32; const void* dy_bf16;
                 int B, Cin, H, W, Cout; float* gw; float* gbias;
                 float* partial;   /* scratch of ctdd_unet_first_conv_wgrad_scratch(B, H, Cin, Cout) floats */ } ctdd_first_wgrad_args;
int64_t ctdd_unet_first_conv_wgrad_scratch(int B, int H, int Cin, int Cout);
int ctdd_unet_first_conv_wgrad(const void* first_wgrad_args, void* stream);

/* One table for all convolution weights of a network: torch parameter -> forward layout [N][Ktot] and data-gradient
 * layout [C][ntap*N] (tap-flipped when flip != 0) in ONE launch per step; packed gradients -> torch layout in ONE launch. */
typedef struct {
  const float* w; void* fwd; void* dgrad; float* gw; float* grad;
  int N, Cin_tot, c_off, C, ntap, Ktot, koff, flip;
  int ldd, pad_;      /* dgrad: columns per tap (>= N; padding columns stay zero) */
  int64_t first;      /* running sum of N*C*ntap over the preceding entries */
} ctdd_pack_entry;
/* rng_bump (may be NULL): device {seed, step}; step += 1 (one dropout stream per training forward) */
int ctdd_unet_pack_weights(const void* table_dev, int nent, int64_t total, int f32, uint64_t* rng_bump, void* stream);
int ctdd_unet_unpack_grads(const void* table_dev, int nent, int64_t total, void* stream);

#ifdef __cplusplus
}
#endif
#endif
