// optim.hip -- K28: the parameter update of the train step in two launches over ALL parameter
// tensors (reference lib/training/training.py:17-40: clip_grad_norm_ -> Adam.step -> EMA update;
// lib/models/models.py:745-758).  torch's path is ~5 multi-tensor launches per stage over ~300 tensors.
//
//   k_grad_sumsq   sum of squares of every gradient element -> one fp64 scalar (fp32 per-thread
//                  partials over <= 256 elements, fp64 from the wave reduction on)
//   k_adam_ema     clip coefficient from that scalar, then per element
//                    g  = grad * clip
//                    m  = m + (1-b1)(g - m)                  (exp_avg.lerp_)
//                    v  = v*b2 + (1-b2) g g                  (mul_ + addcmul_)
//                    p  = p - (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
//                    s  = s + (1-decay)(p - s)               (EMA shadow, optional)
//                  in torch.optim.Adam's operation order (single-tensor formulas, fp32).
// A chunk table maps workgroups to (tensor, offset): tensors are visited in 16 Ki-element chunks.
#include "common.hpp"

namespace ctdd {

struct OptTensor {
  float* p; const float* g; float* m; float* v; float* shadow; int64_t n;
};
struct OptChunk { int tensor; int pad; int64_t start; };
constexpr int OPT_CHUNK = 16384;

__global__ __launch_bounds__(256) void k_grad_sumsq(const OptTensor* __restrict__ tt, const OptChunk* __restrict__ cc,
                                                   double* __restrict__ out) {
  const OptChunk c = cc[blockIdx.x];
  const OptTensor t = tt[c.tensor];
  const int64_t end = c.start + OPT_CHUNK < t.n ? c.start + OPT_CHUNK : t.n;
  float s = 0.0f;
  for (int64_t i = c.start + threadIdx.x; i < end; i += 256) { const float g = t.g[i]; s = fmaf(g, g, s); }
  double d = (double)s;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) d += __shfl_xor(d, o, WAVE);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}

struct AdamArgs {
  float lr_over_bc1, beta1, beta2, eps, bc2_sqrt;   // step_size = lr / (1 - b1^t); sqrt(1 - b2^t)
  float max_norm;                                    // <= 0: no clipping
  float ema_w;                                       // 1 - decay; < 0: no EMA
};
__global__ __launch_bounds__(256) void k_adam_ema(const OptTensor* __restrict__ tt, const OptChunk* __restrict__ cc,
                                                 const double* __restrict__ sumsq, const AdamArgs a) {
  const OptChunk c = cc[blockIdx.x];
  const OptTensor t = tt[c.tensor];
  const int64_t end = c.start + OPT_CHUNK < t.n ? c.start + OPT_CHUNK : t.n;
  float clip = 1.0f;
  if (a.max_norm > 0.0f) {                           // clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
    const float norm = (float)sqrt(*sumsq);
    clip = fminf(a.max_norm / (norm + 1e-6f), 1.0f);
  }
  const float w1 = 1.0f - a.beta1, w2 = 1.0f - a.beta2;
  for (int64_t i = c.start + threadIdx.x; i < end; i += 256) {
    const float g = t.g[i] * clip;
    float m = t.m[i], v = t.v[i], p = t.p[i];
    m = m + w1 * (g - m);
    v = v * a.beta2 + w2 * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - a.lr_over_bc1 * (m / denom);
    t.m[i] = m; t.v[i] = v; t.p[i] = p;
    if (a.ema_w >= 0.0f && t.shadow) { const float s = t.shadow[i]; t.shadow[i] = s + a.ema_w * (p - s); }
  }
}

}  // namespace ctdd

using namespace ctdd;

extern "C" int ctdd_opt_chunk_elems(void) { return OPT_CHUNK; }

// The two launches separately, for callers whose tensors fall into several tables (param groups with different
// hyper-parameters, parameters whose Adam step counts differ): ONE gradient norm over all tables, then one update per table.
extern "C" int ctdd_grad_sumsq(const void* tensors, const void* chunks, int nchunks, double* sumsq_scratch, int zero_first,
                               void* stream) {
  CTDD_REQUIRE(tensors && chunks && nchunks > 0 && sumsq_scratch, CTDD_EINVAL, "grad sumsq: null table / scratch");
  hipStream_t st = (hipStream_t)stream;
  if (zero_first && hipMemsetAsync(sumsq_scratch, 0, sizeof(double), st) != hipSuccess) return finish_launch("memset");
  hipLaunchKernelGGL(k_grad_sumsq, dim3(nchunks), dim3(256), 0, st, (const OptTensor*)tensors, (const OptChunk*)chunks,
                     sumsq_scratch);
  return finish_launch("k_grad_sumsq");
}

static int adam_apply(const void* tensors, const void* chunks, int nchunks, float lr, float beta1, float beta2, float eps,
                      int64_t step, float max_norm, float ema_decay, const double* sumsq, hipStream_t st) {
  AdamArgs a;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  a.lr_over_bc1 = (float)((double)lr / bc1);
  a.bc2_sqrt = (float)sqrt(bc2);
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.max_norm = max_norm;
  a.ema_w = ema_decay >= 0.0f ? 1.0f - ema_decay : -1.0f;
  hipLaunchKernelGGL(k_adam_ema, dim3(nchunks), dim3(256), 0, st, (const OptTensor*)tensors, (const OptChunk*)chunks, sumsq, a);
  return finish_launch("k_adam_ema");
}

extern "C" int ctdd_adam_ema_apply(const void* tensors, const void* chunks, int nchunks, float lr, float beta1, float beta2,
                                   float eps, int64_t step, float max_norm, float ema_decay, const double* sumsq,
                                   void* stream) {
  CTDD_REQUIRE(tensors && chunks && nchunks > 0 && (sumsq || max_norm <= 0.0f), CTDD_EINVAL, "adam apply: null table / norm");
  CTDD_REQUIRE(step >= 1 && beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f, CTDD_EINVAL,
               "adam apply: step=%lld beta=(%g,%g)", (long long)step, (double)beta1, (double)beta2);
  return adam_apply(tensors, chunks, nchunks, lr, beta1, beta2, eps, step, max_norm, ema_decay, sumsq, (hipStream_t)stream);
}

extern "C" int ctdd_adam_ema_step(const void* tensors, const void* chunks, int nchunks, float lr, float beta1, float beta2,
                                  float eps, int64_t step, float max_norm, float ema_decay, double* sumsq_scratch,
                                  void* stream) {
  CTDD_REQUIRE(tensors && chunks && nchunks > 0 && sumsq_scratch, CTDD_EINVAL, "adam step: null table / scratch");
  CTDD_REQUIRE(step >= 1 && beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f, CTDD_EINVAL,
               "adam step: step=%lld beta=(%g,%g)", (long long)step, (double)beta1, (double)beta2);
  hipStream_t st = (hipStream_t)stream;
  if (max_norm > 0.0f) {
    if (hipMemsetAsync(sumsq_scratch, 0, sizeof(double), st) != hipSuccess) return finish_launch("memset");
    hipLaunchKernelGGL(k_grad_sumsq, dim3(nchunks), dim3(256), 0, st, (const OptTensor*)tensors, (const OptChunk*)chunks,
                       sumsq_scratch);
    if (int rc = finish_launch("k_grad_sumsq")) return rc;
  }
  return adam_apply(tensors, chunks, nchunks, lr, beta1, beta2, eps, step, max_norm, ema_decay, sumsq_scratch, st);
}
