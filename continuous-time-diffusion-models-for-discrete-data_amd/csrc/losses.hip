// losses.hip -- K12: categorical-ratio-matching objectives, forward value and gradient w.r.t. the
// logits in one pass over the (B, D, S) tensor (reference lib/losses/losses.py: CatRM._comp_loss
// 794-836, CatRMNLL 1146-1242, with get_logprob_with_logits 'direct', model_utils.py:30-38).
//
// Per row (b,d): ll = log_softmax(l), x = x_t, p = e^ll
//   rm    loss = -ll[x]                                              g = dloss/dll = -onehot(x)
//   mle   loss = -((S-1) ll[x] + sum_s log1mexp(ll[s]) - log1mexp(ll[x]))
//                                                                     g[s] = p/(1-p) (s != x), g[x] = -(S-1)
//   elbo  loss = sum_{s != x} e^{ll[s]-ll[x]} q[s,x] + (ll[s]-ll[x]) q[x,s]
//                                                                     g[s] = e^{d} q[s,x] + q[x,s], g[x] = -sum g
//   dloss/dl[j] = g[j] - p[j] sum_s g[s]
// and optionally + nll_scale * CE(l, x0):  (p[j] - onehot(x0)[j]).  The objective is
// scale * sum_rows loss + nll_scale * sum_rows -ll[x0]; rows' values go to a fp64 buffer summed by a
// second launch (one scalar, no atomics on a single address).  One wave per row, lanes stride over s.
#include "common.hpp"

namespace ctdd {

struct CrmArgs {
  const float* logits; const int32_t* xt; const int32_t* x0; const float* qt0;   // qt0 (B,S,S) for elbo, x0 for the CE term
  int64_t rows; int D, S, loss_type;
  float scale, nll_scale;
  float* grad; double* row_loss;
};

__device__ inline float lwave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
__device__ inline float lwave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
  return v;
}
// log(1 - e^y), y <= 0, and h = -d/dy = e^y / (1 - e^y), both branches as lib/utils/utils.py log1mexp
__device__ inline void log1mexp_h(float y, float& f, float& h) {
  y = -fabsf(y);
  if (y > -0.693f) { const float em = -expm1f(y); f = logf(em); h = expf(y) / em; }
  else { const float e = expf(y); f = log1pf(-e); h = e / (1.0f - e); }
}

__global__ __launch_bounds__(256) void k_crm_rows(const CrmArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int S = a.S, b = (int)(row / a.D);
  const float* l = a.logits + (size_t)row * S;
  float* gr = a.grad + (size_t)row * S;
  const int x = min(max(a.xt[row], 0), S - 1);
  const int x0 = a.x0 ? min(max(a.x0[row], 0), S - 1) : -1;
  float m = -INFINITY;
  for (int s = lane; s < S; s += 64) m = fmaxf(m, l[s]);
  m = lwave_max(m);
  float z = 0.0f;
  for (int s = lane; s < S; s += 64) z += expf(l[s] - m);
  const float L = m + logf(lwave_sum(z));
  const float llx = l[x] - L;
  // pass 3: row loss and sum_s g[s]
  float lsum = 0.0f, gsum = 0.0f;
  const float* q_col = a.qt0 ? a.qt0 + (size_t)b * S * S + x : nullptr;          // q[s,x], stride S
  const float* q_row = a.qt0 ? a.qt0 + ((size_t)b * S + x) * S : nullptr;        // q[x,s]
  if (a.loss_type == 1) {
    for (int s = lane; s < S; s += 64) {
      float f, h;
      log1mexp_h(l[s] - L, f, h);
      lsum += f;
      if (s != x) gsum += h;
    }
    float fx, hx;
    log1mexp_h(llx, fx, hx);
    lsum = lwave_sum(lsum);
    gsum = lwave_sum(gsum) - (float)(S - 1);
    lsum = -((float)(S - 1) * llx + lsum - fx);
  } else if (a.loss_type == 2) {
    for (int s = lane; s < S; s += 64)
      if (s != x) {
        const float d = (l[s] - L) - llx;
        lsum += expf(d) * q_col[(size_t)s * S] + d * q_row[s];
      }
    lsum = lwave_sum(lsum);
    gsum = 0.0f;
  } else {
    lsum = -llx;
    gsum = -1.0f;
  }
  // pass 4: gradient
  float gx_acc = 0.0f;                                  // elbo: g[x] = -sum_{s != x} g[s]
  if (a.loss_type == 2) {
    for (int s = lane; s < S; s += 64)
      if (s != x) gx_acc += expf((l[s] - L) - llx) * q_col[(size_t)s * S] + q_row[s];
    gx_acc = -lwave_sum(gx_acc);
  }
  for (int s = lane; s < S; s += 64) {
    const float ll = l[s] - L, p = expf(ll);
    float g;
    if (a.loss_type == 1) {
      float f, h;
      log1mexp_h(ll, f, h);
      g = s == x ? -(float)(S - 1) : h;
    } else if (a.loss_type == 2) {
      g = s == x ? gx_acc : expf(ll - llx) * q_col[(size_t)s * S] + q_row[s];
    } else {
      g = s == x ? -1.0f : 0.0f;
    }
    float v = a.scale * (g - p * gsum);
    if (x0 >= 0) v += a.nll_scale * (p - (s == x0 ? 1.0f : 0.0f));
    gr[s] = v;
  }
  if (lane == 0) {
    double v = (double)a.scale * (double)lsum;
    if (x0 >= 0) v += (double)a.nll_scale * (double)(-(l[x0] - L));
    a.row_loss[row] = v;
  }
}

__global__ __launch_bounds__(256) void k_sum_rows(const double* __restrict__ v, int64_t n, float* __restrict__ out) {
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += v[i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, WAVE);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((part[0] + part[1]) + (part[2] + part[3]));
}

}  // namespace ctdd

using namespace ctdd;

extern "C" int ctdd_crm_loss(const float* logits, const int32_t* xt, const int32_t* x0, const float* qt0, int B, int D, int S,
                             int loss_type, float scale, float nll_scale, float* grad_logits, double* row_scratch,
                             float* out_loss, void* stream) {
  CTDD_REQUIRE(logits && xt && grad_logits && row_scratch && out_loss, CTDD_EINVAL, "crm loss: null buffer");
  CTDD_REQUIRE(B > 0 && D > 0 && S >= 2, CTDD_EINVAL, "crm loss: B=%d D=%d S=%d", B, D, S);
  CTDD_REQUIRE(loss_type >= 0 && loss_type <= 2, CTDD_EINVAL, "crm loss: loss_type %d (0 rm, 1 mle, 2 elbo)", loss_type);
  CTDD_REQUIRE(loss_type != 2 || qt0, CTDD_EINVAL, "crm loss: elbo needs q_{t|0}");
  CrmArgs a = {logits, xt, x0, qt0, (int64_t)B * D, D, S, loss_type, scale, nll_scale, grad_logits, row_scratch};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_crm_rows, dim3((unsigned)((a.rows + 3) / 4)), dim3(256), 0, st, a);
  if (int rc = finish_launch("k_crm_rows")) return rc;
  hipLaunchKernelGGL(k_sum_rows, dim3(1), dim3(256), 0, st, (const double*)row_scratch, a.rows, out_loss);
  return finish_launch("k_sum_rows");
}
