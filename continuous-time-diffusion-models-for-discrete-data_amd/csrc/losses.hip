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
  int ll_in;     // `logits` already holds ll_all = log p_t(. | x^{\d}) (reverse_prob / reverse_logscale logit types, from
                 // ctdd_logprob): no log-softmax, grad = d loss / d ll_all (ctdd_logprob_bwd chains it to the logits)
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
  float L = 0.0f;
  if (!a.ll_in) {
    float m = -INFINITY;
    for (int s = lane; s < S; s += 64) m = fmaxf(m, l[s]);
    m = lwave_max(m);
    float z = 0.0f;
    for (int s = lane; s < S; s += 64) z += expf(l[s] - m);
    L = m + logf(lwave_sum(z));
  }
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
    float v = a.ll_in ? a.scale * g : a.scale * (g - p * gsum);
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
  CrmArgs a = {logits, xt, x0, qt0, (int64_t)B * D, D, S, loss_type, scale, nll_scale, grad_logits, row_scratch, 0};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_crm_rows, dim3((unsigned)((a.rows + 3) / 4)), dim3(256), 0, st, a);
  if (int rc = finish_launch("k_crm_rows")) return rc;
  hipLaunchKernelGGL(k_sum_rows, dim3(1), dim3(256), 0, st, (const double*)row_scratch, a.rows, out_loss);
  return finish_launch("k_sum_rows");
}

// the same objective on ll_all (reverse_prob / reverse_logscale logit types): value and d loss / d ll_all
extern "C" int ctdd_crm_loss_ll(const float* ll_all, const int32_t* xt, const float* qt0, int B, int D, int S, int loss_type, float scale,
                                float* grad_ll, double* row_scratch, float* out_loss, void* stream) {
  CTDD_REQUIRE(ll_all && xt && grad_ll && row_scratch && out_loss, CTDD_EINVAL, "crm loss (ll): null buffer");
  CTDD_REQUIRE(B > 0 && D > 0 && S >= 2, CTDD_EINVAL, "crm loss (ll): B=%d D=%d S=%d", B, D, S);
  CTDD_REQUIRE(loss_type >= 0 && loss_type <= 2 && (loss_type != 2 || qt0), CTDD_EINVAL, "crm loss (ll): loss_type %d / elbo needs q_{t|0}", loss_type);
  CrmArgs a = {ll_all, xt, nullptr, qt0, (int64_t)B * D, D, S, loss_type, scale, 0.0f, grad_ll, row_scratch, 1};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_crm_rows, dim3((unsigned)((a.rows + 3) / 4)), dim3(256), 0, st, a);
  if (int rc = finish_launch("k_crm_rows")) return rc;
  hipLaunchKernelGGL(k_sum_rows, dim3(1), dim3(256), 0, st, (const double*)row_scratch, a.rows, out_loss);
  return finish_launch("k_sum_rows");
}

// ================================================================== K11: tauLDR CT-ELBO, value + d/dlogits
// Reference lib/losses/losses.py:106-286 (CTElbo.calc_loss; NLL / CTElboLambda share the body) with
// one_forward_pass = True: logits = model(x_t), reg_x = x~.  Per sample b with tables q = q_{t|0}
// (B,S,S), qT = its transpose, R = rate(t), and per row (b,d) with x = x~_bd, p = softmax(l):
//   reg_row   = sum_s0 p[s0] A[x][s0],     A[x][s0] = (sum_{s != x} R[s,x] q[s0,s]) / (q[s0,x] + eps)
//   u[s]      = sum_s0 (p[s0] / (q[s0,x] + eps)) q[s0,s],   inner = log(u + eps)
//   Wt[s]     = [s != x] R[s,x] q[x0,s] / (q[x0,x] + eps)
//   outer_row = sum_s Wt[s] inner[s]
//   norm_row  = sum_s [s != x] R[s,x] q[x0,s] / (Z[s] (q[x0,x] + eps)),  Z[s] = sum_d' rs[x~_bd'] - rs[x] + rs[s], rs = -diag R
//   loss = elbo_scale * mean_b( -sum_d outer / sum_d norm ) + reg_scale * mean_b sum_d reg + nll_scale * sum_{b,d} -log p[x0]
//   (reg_scale = elbo_scale for the one-forward-pass objective; ctdd_ctelbo_loss_terms exposes the two weights so that the
//    two-forward-pass objective is two launches, one per network output)
// Backward: G[s] = c_b Wt[s] / (u[s] + eps), c_b = -elbo_scale / (B norm_b);  dr[s0] = sum_s G[s] q[s0,s];
//   dp[s0] = dr[s0] / (q[s0,x] + eps) + (reg_scale / B) A[x][s0];  dl[j] = p[j] (dp[j] - sum p dp) + nll_scale (p[j] - [j = x0]).
// Workgroup = 8 rows of one sample, thread s <-> state s (S <= 256); the two S x S contractions stream q / qT
// once per 8 rows with the row vectors in LDS.  Correctness-first (fp32 FMA chains, no matrix cores yet).
namespace ctdd {

constexpr int LRB = 8;   // rows per workgroup

struct ElboArgs {
  const float* logits; const int32_t* x0; const int32_t* xt;     // xt = x~ (= reg_x)
  const float* q; const float* qT; const float* R;               // (B,S,S) each
  int B, D, S; float eps, elbo_scale, nll_scale;
  float reg_scale;    // weight of the regulariser term (= elbo_scale in the one-forward-pass objective; the two-pass objective runs
                      // the kernel once per network output with one of the two term weights at zero)
  float* RT;          // (B,S,S) or null: RT[b][x][s] = R[b][s][x], written by k_elbo_atab (all column blocks) so that a row's forward
                      // rates into its state x are ONE coalesced table row (the column gather R[s][x] per (row, state) was a cache
                      // line per lane: the outer / G passes ran at a third of their memory time)
  float* Atab;        // (B,S,S): A[b][x][s0]
  float* base_sum;    // (B)
  float* u;           // (B,D,S)
  double* rows;       // (B*D,4): outer, norm, reg, nll
  float* cb;          // (B)
  float* grad;        // (B,D,S)
  float* out_loss;    // (1)
  int ll_in;          // ScoreElbo only: `logits` holds ll_all (reverse logit types); grad = d loss / d ll_all
};

// block reduction of LRB values per thread; result broadcast through red[]
__device__ inline void block_sum8(float (&v)[LRB], float* red /* [4][LRB] */, float (&out)[LRB]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    const float s = lwave_sum(v[r]);
    if (lane == 0) red[w * LRB + r] = s;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < LRB; ++r) out[r] = (red[r] + red[LRB + r]) + (red[2 * LRB + r] + red[3 * LRB + r]);
  __syncthreads();
}
__device__ inline void block_max8(float (&v)[LRB], float* red, float (&out)[LRB]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    const float s = lwave_max(v[r]);
    if (lane == 0) red[w * LRB + r] = s;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < LRB; ++r) out[r] = fmaxf(fmaxf(red[r], red[LRB + r]), fmaxf(red[2 * LRB + r], red[3 * LRB + r]));
  __syncthreads();
}

// A table + base_sum.  grid (ceil(S/LRB), B): rows s0 of sample b.
__global__ __launch_bounds__(256) void k_elbo_atab(const ElboArgs a) {
  __shared__ float qrow[LRB][256];
  const int S = a.S, b = blockIdx.y, s00 = blockIdx.x * LRB, t = threadIdx.x;
  const float* q = a.q + (size_t)b * S * S;
  const float* R = a.R + (size_t)b * S * S;
  for (int i = t; i < LRB * S; i += 256) {
    const int r = i / S, s = i % S;
    qrow[r][s] = s00 + r < S ? q[(size_t)(s00 + r) * S + s] : 0.0f;
  }
  __syncthreads();
  if (t < S) {                                      // thread = x
    float acc[LRB];
#pragma unroll
    for (int r = 0; r < LRB; ++r) acc[r] = 0.0f;
    for (int s = 0; s < S; ++s) {
      const float rv = s == t ? 0.0f : R[(size_t)s * S + t];
#pragma unroll
      for (int r = 0; r < LRB; ++r) acc[r] = fmaf(qrow[r][s], rv, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < LRB; ++r)
      if (s00 + r < S) a.Atab[((size_t)b * S + t) * S + s00 + r] = acc[r] / (qrow[r][t] + a.eps);
    if (a.RT) {
#pragma unroll
      for (int r = 0; r < LRB; ++r)
        if (s00 + r < S) a.RT[((size_t)b * S + t) * S + s00 + r] = R[(size_t)(s00 + r) * S + t];
    }
  }
  if (blockIdx.x == 0) {                            // base_sum[b] = sum_d rs[x~_bd]
    float s = 0.0f;
    for (int d = t; d < a.D; d += 256) {
      const int x = min(max(a.xt[(size_t)b * a.D + d], 0), S - 1);
      s -= R[(size_t)x * S + x];
    }
    __shared__ float red[4];
    s = lwave_sum(s);
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    if (t == 0) a.base_sum[b] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// forward rows.  grid (ceil(D/LRB), B).  PHASE 0: everything, the S x S contraction as fp32 FMA chains (any S <= 256);
// S % 32 == 0 (MNIST / CIFAR: S = 256) splits it around the matrix-core GEMM k_bgemm_f32:  PHASE 1 writes the left operand
// rvec = p / (q[., x] + eps) (into the grad buffer, free until backward) and the reg / nll row sums,  PHASE 2 reads u = rvec @ q
// back and forms the outer / norm row sums.
template <int PHASE>
__global__ __launch_bounds__(256) void k_elbo_fwd(const ElboArgs a) {
  __shared__ float rvec[LRB][256];
  __shared__ float red[4 * LRB];
  const int S = a.S, D = a.D, b = blockIdx.y, d0 = blockIdx.x * LRB, t = threadIdx.x;
  const bool act = t < S;
  const float* q = a.q + (size_t)b * S * S;
  const float* qT = a.qT + (size_t)b * S * S;
  const float* R = a.R + (size_t)b * S * S;
  int x[LRB], x0[LRB];
  bool ok[LRB];
  float l[LRB], mx[LRB], tmp[LRB], p[LRB];
  auto row_of = [&](int r) { return (size_t)b * D + d0 + r; };
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    ok[r] = d0 + r < D;
    const size_t row = (size_t)b * D + (ok[r] ? d0 + r : D - 1);
    x[r] = min(max(a.xt[row], 0), S - 1);
    x0[r] = min(max(a.x0[row], 0), S - 1);
    l[r] = act ? a.logits[row * S + t] : -INFINITY;
    tmp[r] = l[r];
  }
  block_max8(tmp, red, mx);
#pragma unroll
  for (int r = 0; r < LRB; ++r) tmp[r] = act ? expf(l[r] - mx[r]) : 0.0f;
  float zs[LRB];
  block_sum8(tmp, red, zs);
  float regp[LRB], nllp[LRB];
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    const float L = mx[r] + logf(zs[r]);
    p[r] = act ? expf(l[r] - L) : 0.0f;
    const float den = act ? qT[(size_t)x[r] * S + t] + a.eps : 1.0f;
    if (PHASE == 1) { if (act && ok[r]) a.grad[row_of(r) * S + t] = p[r] / den; }
    else rvec[r][t] = p[r] / den;
    regp[r] = act ? p[r] * a.Atab[((size_t)b * S + x[r]) * S + t] : 0.0f;
    nllp[r] = (act && t == x0[r]) ? -(l[r] - L) : 0.0f;
  }
  if (PHASE == 1) {
    float o3[LRB], o4[LRB];
    block_sum8(regp, red, o3);
    block_sum8(nllp, red, o4);
    if (t < LRB && d0 + t < D) {
      double* dst = a.rows + ((size_t)b * D + d0 + t) * 4;
      float v3 = 0, v4 = 0;
#pragma unroll
      for (int r = 0; r < LRB; ++r)
        if (r == t) { v3 = o3[r]; v4 = o4[r]; }
      dst[2] = v3; dst[3] = v4;
    }
    return;
  }
  __syncthreads();
  float acc[LRB];
#pragma unroll
  for (int r = 0; r < LRB; ++r) acc[r] = 0.0f;
  if (PHASE == 2) {
#pragma unroll
    for (int r = 0; r < LRB; ++r) acc[r] = (act && ok[r]) ? a.u[row_of(r) * S + t] : 0.0f;
  } else if (act)
    for (int s0 = 0; s0 < S; ++s0) {
      const float qv = q[(size_t)s0 * S + t];
#pragma unroll
      for (int r = 0; r < LRB; ++r) acc[r] = fmaf(rvec[r][s0], qv, acc[r]);
    }
  const float bsum = a.base_sum[b];
  float outp[LRB], normp[LRB];
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    outp[r] = 0.0f; normp[r] = 0.0f;
    if (act) {
      const size_t row = (size_t)b * D + d0 + r;
      if (PHASE == 0 && ok[r]) a.u[row * S + t] = acc[r];
      const float orate = t == x[r] ? 0.0f : (a.RT ? a.RT[((size_t)b * S + x[r]) * S + t] : R[(size_t)t * S + x[r]]);
      const float qx0 = q[(size_t)x0[r] * S + t];
      const float qx0xt = q[(size_t)x0[r] * S + x[r]] + a.eps;
      const float Z = bsum + R[(size_t)x[r] * S + x[r]] - R[(size_t)t * S + t];      // base_sum - rs[x] + rs[s]
      outp[r] = orate * (qx0 / qx0xt) * logf(acc[r] + a.eps);
      normp[r] = orate * qx0 / (Z * qx0xt);
    }
  }
  float o1[LRB], o2[LRB], o3[LRB], o4[LRB];
  block_sum8(outp, red, o1);
  block_sum8(normp, red, o2);
  if (PHASE == 0) { block_sum8(regp, red, o3); block_sum8(nllp, red, o4); }
  if (t < LRB && d0 + t < D) {
    double* dst = a.rows + ((size_t)b * D + d0 + t) * 4;
    float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
#pragma unroll
    for (int r = 0; r < LRB; ++r)
      if (r == t) { v1 = o1[r]; v2 = o2[r]; if (PHASE == 0) { v3 = o3[r]; v4 = o4[r]; } }
    dst[0] = v1; dst[1] = v2;
    if (PHASE == 0) { dst[2] = v3; dst[3] = v4; }
  }
}


// ---- C_b = A_b W_b^T on the exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32): A (B, M, K), W (B, N, K), C (B, M, N)
// with K = N = 32 NT (S = 256: NT = 8).  Workgroup = 128 rows x all N of one sample, a wave = 32 rows x N (NT accumulator
// tiles); K in chunks of 16 through LDS (rows padded to 20 floats: the 16-byte fragment reads of 16 lanes cover all 64
// banks), next chunk's global loads in registers behind the current chunk's 16 NT matrix instructions per wave.  A lane half
// g contracts k = 8 g .. 8 g + 7 of the chunk: each 16-byte read feeds four instructions.
constexpr int GK = 16, GLD = GK + 4;
template <int NT>
__global__ __launch_bounds__(256) void k_bgemm_f32(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ Cm, int M) {
  constexpr int N = 32 * NT, K = N, WV = (NT + 1) / 2;              // WV: float4 of W per thread and chunk (N * 16 / 4 / 256, rounded up)
  __shared__ __attribute__((aligned(16))) float As[2][128 * GLD];
  __shared__ __attribute__((aligned(16))) float Ws[2][N * GLD];
  using f32x16l = __attribute__((ext_vector_type(16))) float;
  const int b = blockIdx.y, m0 = blockIdx.x * 128, t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, g = lane >> 5;
  const float* Ab = A + (size_t)b * M * K;
  const float* Wb = W + (size_t)b * N * K;
  float4 pa[2], pw[WV];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = t + 256 * u, row = idx >> 2, c4 = (idx & 3) * 4;
      pa[u] = m0 + row < M ? *(const float4*)(Ab + (size_t)(m0 + row) * K + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < WV; ++u) {
      const int idx = t + 256 * u, n = idx >> 2, c4 = (idx & 3) * 4;
      pw[u] = n < N ? *(const float4*)(Wb + (size_t)n * K + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = t + 256 * u, row = idx >> 2, c4 = (idx & 3) * 4;
      *(float4*)(&As[buf][row * GLD + c4]) = pa[u];
    }
#pragma unroll
    for (int u = 0; u < WV; ++u) {
      const int idx = t + 256 * u, n = idx >> 2, c4 = (idx & 3) * 4;
      if (n < N) *(float4*)(&Ws[buf][n * GLD + c4]) = pw[u];
    }
  };
  f32x16l acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  fetch(0);
  stash(0);
  __syncthreads();
  for (int c = 0; c < K / GK; ++c) {
    const int buf = c & 1;
    if (c + 1 < K / GK) fetch((c + 1) * GK);
    const float* Aw = &As[buf][(wave * 32 + li) * GLD + 8 * g];
    const float* Ww = &Ws[buf][li * GLD + 8 * g];
#pragma unroll
    for (int s4 = 0; s4 < 2; ++s4) {
      const float4 av = *(const float4*)(Aw + 4 * s4);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const float4 bv = *(const float4*)(Ww + (size_t)i * 32 * GLD + 4 * s4);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[i], 0, 0, 0);
      }
    }
    if (c + 1 < K / GK) stash(buf ^ 1);          // (the other buffer: its readers passed the barrier of the previous chunk)
    __syncthreads();
  }
  // accumulator tile i: column 32 i + li, rows (r & 3) + 8 (r >> 2) + 4 g of the wave's 32
  float* Cb = Cm + (size_t)b * M * N;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
      if (row < M) Cb[(size_t)row * N + 32 * i + li] = acc[i][r];
    }
}

// per-sample sums of the four row quantities: one workgroup per sample -> sums (B, 4) fp64
__global__ __launch_bounds__(256) void k_elbo_sample_sums(const ElboArgs a, double* __restrict__ sums) {
  __shared__ double red[4][256];
  const int b = blockIdx.x, t = threadIdx.x;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int d = t; d < a.D; d += 256) {
    const double* r = a.rows + ((size_t)b * a.D + d) * 4;
    s[0] += r[0]; s[1] += r[1]; s[2] += r[2]; s[3] += r[3];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][t] = s[k];
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (t < o) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][t] += red[k][t + o];
    }
    __syncthreads();
  }
  if (t < 4) sums[(size_t)b * 4 + t] = red[t][0];
}

// per-sample sums -> c_b and the scalar loss.  one workgroup.
__global__ __launch_bounds__(256) void k_elbo_reduce(const ElboArgs a, const double* __restrict__ sums) {
  __shared__ double acc[256];
  double tot = 0.0;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    const double so = sums[(size_t)b * 4], sn = sums[(size_t)b * 4 + 1], sr = sums[(size_t)b * 4 + 2], sl = sums[(size_t)b * 4 + 3];
    a.cb[b] = (float)(-(double)a.elbo_scale / ((double)a.B * sn));
    tot += ((double)a.elbo_scale * (-so / sn) + (double)a.reg_scale * sr) / (double)a.B + (double)a.nll_scale * sl;
  }
  acc[threadIdx.x] = tot;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (threadIdx.x < o) acc[threadIdx.x] += acc[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.out_loss[0] = (float)acc[0];
}

// backward rows.  grid (ceil(D/LRB), B).  PHASE 0: everything (fp32 FMA contraction);  PHASE 1: G rows -> the grad buffer (the
// matrix-core GEMM then writes dr = G @ q^T over u);  PHASE 2: dr -> d/dlogits.
template <int PHASE>
__global__ __launch_bounds__(256) void k_elbo_bwd(const ElboArgs a) {
  __shared__ float gvec[LRB][256];
  __shared__ float red[4 * LRB];
  const int S = a.S, D = a.D, b = blockIdx.y, d0 = blockIdx.x * LRB, t = threadIdx.x;
  const bool act = t < S;
  const float* q = a.q + (size_t)b * S * S;
  const float* qT = a.qT + (size_t)b * S * S;
  const float* R = a.R + (size_t)b * S * S;
  const float cb = a.cb[b];
  int x[LRB], x0[LRB];
  bool ok[LRB];
  float l[LRB], mx[LRB], tmp[LRB], p[LRB];
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    ok[r] = d0 + r < D;
    const size_t row = (size_t)b * D + (ok[r] ? d0 + r : D - 1);
    x[r] = min(max(a.xt[row], 0), S - 1);
    x0[r] = min(max(a.x0[row], 0), S - 1);
    l[r] = act ? a.logits[row * S + t] : -INFINITY;
    tmp[r] = l[r];
    // G[s] = c_b Wt[s] / (u[s] + eps)
    float G = 0.0f;
    if (act && PHASE != 2) {
      const float orate = t == x[r] ? 0.0f : (a.RT ? a.RT[((size_t)b * S + x[r]) * S + t] : R[(size_t)t * S + x[r]]);
      const float qx0 = q[(size_t)x0[r] * S + t];
      const float qx0xt = q[(size_t)x0[r] * S + x[r]] + a.eps;
      G = cb * orate * (qx0 / qx0xt) / (a.u[row * S + t] + a.eps);
    }
    if (PHASE == 1) { if (act && ok[r]) a.grad[row * S + t] = G; }
    else if (PHASE == 0) gvec[r][t] = G;
  }
  if (PHASE == 1) return;
  block_max8(tmp, red, mx);                          // (contains the barrier that publishes gvec)
#pragma unroll
  for (int r = 0; r < LRB; ++r) tmp[r] = act ? expf(l[r] - mx[r]) : 0.0f;
  float zs[LRB];
  block_sum8(tmp, red, zs);
  float acc[LRB];
#pragma unroll
  for (int r = 0; r < LRB; ++r) { acc[r] = 0.0f; p[r] = act ? expf(l[r] - (mx[r] + logf(zs[r]))) : 0.0f; }
  if (PHASE == 2) {
#pragma unroll
    for (int r = 0; r < LRB; ++r) acc[r] = (act && ok[r]) ? a.u[((size_t)b * D + d0 + r) * S + t] : 0.0f;
  } else if (act)
    for (int s = 0; s < S; ++s) {                    // dr[s0 = t] = sum_s G[s] q[s0, s] = sum_s G[s] qT[s, s0]
      const float qv = qT[(size_t)s * S + t];
#pragma unroll
      for (int r = 0; r < LRB; ++r) acc[r] = fmaf(gvec[r][s], qv, acc[r]);
    }
  float dp[LRB], pd[LRB];
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    dp[r] = 0.0f;
    if (act) dp[r] = acc[r] / (qT[(size_t)x[r] * S + t] + a.eps) + (a.reg_scale / (float)a.B) * a.Atab[((size_t)b * S + x[r]) * S + t];
    pd[r] = p[r] * dp[r];
  }
  float pdot[LRB];
  block_sum8(pd, red, pdot);
#pragma unroll
  for (int r = 0; r < LRB; ++r)
    if (act && ok[r])
      a.grad[((size_t)b * D + d0 + r) * S + t] = p[r] * (dp[r] - pdot[r]) + a.nll_scale * (p[r] - (t == x0[r] ? 1.0f : 0.0f));
}

// ---- the regulariser table on the matrix cores (S % 32 == 0): A[b][x][s0] = sum_{s != x} q[s0][s] R[s][x] / (q[s0][x] + eps)
//   k_elbo_rt:     RT[b][x][s] = (s == x) ? 0 : R[b][s][x]      (32 x 32 tiles through LDS; + base_sum[b])
//   k_bgemm_f32:   Atab[b][x][s0] = sum_s RT[b][x][s] q[b][s0][s]   (the zeroed diagonal: no cancellation against the -R[x][x] term)
//   k_elbo_afix:   Atab[b][x][s0] /= qT[b][x][s0] + eps
// (k_elbo_atab's thread-per-state fp32 FMA chains: 115 us at 64 samples.)
__global__ __launch_bounds__(256) void k_elbo_rt(const ElboArgs a) {
  __shared__ float tile[32][33];
  const int S = a.S, b = blockIdx.z, x0 = blockIdx.x * 32, s0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* R = a.R + (size_t)b * S * S;
#pragma unroll
  for (int k = 0; k < 4; ++k) tile[ty + 8 * k][tx] = R[(size_t)(s0 + ty + 8 * k) * S + x0 + tx];      // [s][x]
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = x0 + ty + 8 * k, s = s0 + tx;
    a.RT[((size_t)b * S + x) * S + s] = s == x ? 0.0f : tile[tx][ty + 8 * k];
  }
  if (blockIdx.x == 0 && blockIdx.y == 0) {           // base_sum[b] = sum_d rs[x~_bd]  (as k_elbo_atab)
    const int t = threadIdx.x;
    float sm = 0.0f;
    for (int d = t; d < a.D; d += 256) {
      const int x = min(max(a.xt[(size_t)b * a.D + d], 0), S - 1);
      sm -= R[(size_t)x * S + x];
    }
    __shared__ float red[4];
    sm = lwave_sum(sm);
    if ((t & 63) == 0) red[t >> 6] = sm;
    __syncthreads();
    if (t == 0) a.base_sum[b] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}
__global__ __launch_bounds__(256) void k_elbo_afix(const ElboArgs a) {
  const size_t n4 = (size_t)a.B * a.S * a.S / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 v = ((const float4*)a.Atab)[i];
    const float4 qv = ((const float4*)a.qT)[i];
    v.x /= qv.x + a.eps; v.y /= qv.y + a.eps; v.z /= qv.z + a.eps; v.w /= qv.w + a.eps;
    ((float4*)a.Atab)[i] = v;
  }
}

// ---- S = 256: the four row passes around the two matrix-core GEMMs with a WAVE per row (a lane owns four consecutive states:
// one 16-byte load per table row and lane, reductions inside the wave).  The workgroup-per-eight-rows kernels above (thread =
// state) spend their time in block reductions -- two barriers each, six of them in a forward pass -- and the second forward
// pass re-formed a softmax it does not use: 85 / 100 / 33 / 68 us for the four passes at 64 x 784 rows.  MODE 0: fwd<1> (rvec,
// reg / nll sums), 1: fwd<2> (outer / norm sums from u), 2: bwd<1> (G), 3: bwd<2> (d/dlogits from dr).  RT must be set.
constexpr int RV_ROWS = 4;                                     // rows per wave, two at a time
template <int MODE>
__global__ __launch_bounds__(256) void k_elbo_rowsv(const ElboArgs a) {
  constexpr int S = 256;
  const int D = a.D, b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, s0 = 4 * lane;
  const int d0 = (blockIdx.x * 4 + wv) * RV_ROWS;
  const float* q = a.q + (size_t)b * S * S;
  const float* qT = a.qT + (size_t)b * S * S;
  const float* R = a.R + (size_t)b * S * S;
  const float* RT = a.RT + (size_t)b * S * S;
  const float* At = a.Atab + (size_t)b * S * S;
  float rdiag[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float bsum = 0.0f, cb = 0.0f;
  if (MODE == 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) rdiag[c] = R[(size_t)(s0 + c) * S + s0 + c];
    bsum = a.base_sum[b];
  }
  if (MODE == 2) cb = a.cb[b];
#pragma unroll
  for (int r2 = 0; r2 < RV_ROWS; r2 += 2) {
    float4 lg[2], t1[2], t2[2], t3[2];
    int x[2], x0[2];
    bool ok[2];
    size_t row[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {                                // everything the two rows need, requested together
      const int d = d0 + r2 + e;
      ok[e] = d < D;
      row[e] = (size_t)b * D + (ok[e] ? d : D - 1);
      x[e] = min(max(a.xt[row[e]], 0), S - 1);
      x0[e] = min(max(a.x0[row[e]], 0), S - 1);
      if (MODE == 0 || MODE == 3) lg[e] = *(const float4*)(a.logits + row[e] * S + s0);
      if (MODE == 0 || MODE == 3) { t1[e] = *(const float4*)(qT + (size_t)x[e] * S + s0); t2[e] = *(const float4*)(At + (size_t)x[e] * S + s0); }
      if (MODE == 1 || MODE == 2) { t1[e] = *(const float4*)(RT + (size_t)x[e] * S + s0); t2[e] = *(const float4*)(q + (size_t)x0[e] * S + s0); }
      if (MODE != 0) t3[e] = *(const float4*)(a.u + row[e] * S + s0);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (MODE == 0 || MODE == 3) {
        const float l4[4] = {lg[e].x, lg[e].y, lg[e].z, lg[e].w};
        const float mx = lwave_max(fmaxf(fmaxf(l4[0], l4[1]), fmaxf(l4[2], l4[3])));
        float ex[4], zs = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) { ex[c] = expf(l4[c] - mx); zs += ex[c]; }
        const float L = mx + logf(lwave_sum(zs));
        float p[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) p[c] = expf(l4[c] - L);
        const float den[4] = {t1[e].x + a.eps, t1[e].y + a.eps, t1[e].z + a.eps, t1[e].w + a.eps};
        const float at[4] = {t2[e].x, t2[e].y, t2[e].z, t2[e].w};
        if (MODE == 0) {
          float reg = 0.0f, nll = 0.0f;
#pragma unroll
          for (int c = 0; c < 4; ++c) { reg = fmaf(p[c], at[c], reg); nll += (s0 + c == x0[e]) ? -(l4[c] - L) : 0.0f; }
          if (ok[e]) *(float4*)(a.grad + row[e] * S + s0) = make_float4(p[0] / den[0], p[1] / den[1], p[2] / den[2], p[3] / den[3]);
          reg = lwave_sum(reg); nll = lwave_sum(nll);
          if (lane == 0 && ok[e]) { double* dst = a.rows + row[e] * 4; dst[2] = reg; dst[3] = nll; }
        } else {
          const float dr[4] = {t3[e].x, t3[e].y, t3[e].z, t3[e].w};
          float dp[4], pd = 0.0f;
#pragma unroll
          for (int c = 0; c < 4; ++c) { dp[c] = dr[c] / den[c] + (a.reg_scale / (float)a.B) * at[c]; pd = fmaf(p[c], dp[c], pd); }
          pd = lwave_sum(pd);
          float o[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = p[c] * (dp[c] - pd) + a.nll_scale * (p[c] - (s0 + c == x0[e] ? 1.0f : 0.0f));
          if (ok[e]) *(float4*)(a.grad + row[e] * S + s0) = make_float4(o[0], o[1], o[2], o[3]);
        }
      } else {
        const float rt[4] = {t1[e].x, t1[e].y, t1[e].z, t1[e].w}, qx[4] = {t2[e].x, t2[e].y, t2[e].z, t2[e].w};
        const float uu[4] = {t3[e].x, t3[e].y, t3[e].z, t3[e].w};
        const float qx0xt = q[(size_t)x0[e] * S + x[e]] + a.eps;
        if (MODE == 1) {
          const float rxx = R[(size_t)x[e] * S + x[e]];
          float outp = 0.0f, normp = 0.0f;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float orate = (s0 + c == x[e]) ? 0.0f : rt[c];
            const float Z = bsum + rxx - rdiag[c];                                    // base_sum - rs[x] + rs[s]
            outp += orate * (qx[c] / qx0xt) * logf(uu[c] + a.eps);
            normp += orate * qx[c] / (Z * qx0xt);
          }
          outp = lwave_sum(outp); normp = lwave_sum(normp);
          if (lane == 0 && ok[e]) { double* dst = a.rows + row[e] * 4; dst[0] = outp; dst[1] = normp; }
        } else {
          float G[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float orate = (s0 + c == x[e]) ? 0.0f : rt[c];
            G[c] = cb * orate * (qx[c] / qx0xt) / (uu[c] + a.eps);
          }
          if (ok[e]) *(float4*)(a.grad + row[e] * S + s0) = make_float4(G[0], G[1], G[2], G[3]);
        }
      }
    }
  }
}

}  // namespace ctdd

extern "C" int64_t ctdd_ctelbo_scratch_bytes(int B, int D, int S) {
  // Atab (B,S,S) f32 | u (B,D,S) f32 | rows (B*D,4) f64 | base_sum (B) f32 | cb (B) f32 | sums (B,4) f64 | RT (B,S,S) f32   (each 256-B aligned)
  auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
  return 2 * al((int64_t)B * S * S * 4) + al((int64_t)B * D * S * 4) + al((int64_t)B * D * 32) + 2 * al((int64_t)B * 4) + al((int64_t)B * 32);
}

extern "C" int ctdd_ctelbo_loss_terms(const float* logits, const int32_t* x0, const int32_t* x_tilde, const float* qt0, const float* qt0T,
                                      const float* rate, int B, int D, int S, float eps, float sig_scale, float reg_scale, float nll_scale,
                                      void* scratch, float* grad_logits, float* out_loss, void* stream);
extern "C" int ctdd_ctelbo_loss(const float* logits, const int32_t* x0, const int32_t* x_tilde, const float* qt0, const float* qt0T,
                                const float* rate, int B, int D, int S, float eps, float elbo_scale, float nll_scale,
                                void* scratch, float* grad_logits, float* out_loss, void* stream) {
  return ctdd_ctelbo_loss_terms(logits, x0, x_tilde, qt0, qt0T, rate, B, D, S, eps, elbo_scale, elbo_scale, nll_scale, scratch, grad_logits,
                                out_loss, stream);
}
extern "C" int ctdd_ctelbo_loss_terms(const float* logits, const int32_t* x0, const int32_t* x_tilde, const float* qt0, const float* qt0T,
                                      const float* rate, int B, int D, int S, float eps, float sig_scale, float reg_scale, float nll_scale,
                                      void* scratch, float* grad_logits, float* out_loss, void* stream) {
  const float elbo_scale = sig_scale;
  CTDD_REQUIRE(logits && x0 && x_tilde && qt0 && qt0T && rate && scratch && grad_logits && out_loss, CTDD_EINVAL, "ct-elbo: null buffer");
  CTDD_REQUIRE(B > 0 && D > 0 && S >= 2 && S <= 256, CTDD_ERANGE, "ct-elbo: B=%d D=%d S=%d (S <= 256)", B, D, S);
  auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
  unsigned char* sp = (unsigned char*)scratch;
  ElboArgs a;
  a.logits = logits; a.x0 = x0; a.xt = x_tilde; a.q = qt0; a.qT = qt0T; a.R = rate;
  a.B = B; a.D = D; a.S = S; a.eps = eps; a.elbo_scale = elbo_scale; a.nll_scale = nll_scale; a.reg_scale = reg_scale;
  a.Atab = (float*)sp; sp += al((int64_t)B * S * S * 4);
  a.u = (float*)sp; sp += al((int64_t)B * D * S * 4);
  a.rows = (double*)sp; sp += al((int64_t)B * D * 32);
  a.base_sum = (float*)sp; sp += al((int64_t)B * 4);
  a.cb = (float*)sp; sp += al((int64_t)B * 4);
  double* sums = (double*)sp; sp += al((int64_t)B * 32);
  a.RT = (float*)sp;
  a.grad = grad_logits; a.out_loss = out_loss; a.ll_in = 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 rg((D + LRB - 1) / LRB, B), gg((D + 127) / 128, B);
  const bool mfma = S % 32 == 0;                     // the S x S contractions on the exact-fp32 matrix instruction
  auto gemm_m = [&](const float* Am, const float* Wm, float* Cm, int Mrows) {
    const dim3 g2((Mrows + 127) / 128, B);
    switch (S / 32) {
      case 1: hipLaunchKernelGGL(k_bgemm_f32<1>, g2, dim3(256), 0, st, Am, Wm, Cm, Mrows); break;
      case 2: hipLaunchKernelGGL(k_bgemm_f32<2>, g2, dim3(256), 0, st, Am, Wm, Cm, Mrows); break;
      case 3: hipLaunchKernelGGL(k_bgemm_f32<3>, g2, dim3(256), 0, st, Am, Wm, Cm, Mrows); break;
      case 4: hipLaunchKernelGGL(k_bgemm_f32<4>, g2, dim3(256), 0, st, Am, Wm, Cm, Mrows); break;
      case 5: hipLaunchKernelGGL(k_bgemm_f32<5>, g2, dim3(256), 0, st, Am, Wm, Cm, Mrows); break;
      case 6: hipLaunchKernelGGL(k_bgemm_f32<6>, g2, dim3(256), 0, st, Am, Wm, Cm, Mrows); break;
      case 7: hipLaunchKernelGGL(k_bgemm_f32<7>, g2, dim3(256), 0, st, Am, Wm, Cm, Mrows); break;
      default: hipLaunchKernelGGL(k_bgemm_f32<8>, g2, dim3(256), 0, st, Am, Wm, Cm, Mrows); break;
    }
    return finish_launch("k_bgemm_f32");
  };
  auto gemm = [&](const float* Am, const float* Wm, float* Cm) { return gemm_m(Am, Wm, Cm, D); };
  (void)gg;
  static const bool atab_fma = [] { const char* e = getenv("CTDD_ELBO_ATAB_FMA"); return e && e[0] == '1'; }();   // (A/B: the FMA-chain table kernel)
  if (mfma && !atab_fma) {
    hipLaunchKernelGGL(k_elbo_rt, dim3(S / 32, S / 32, B), dim3(256), 0, st, a);
    if (int rc = finish_launch("k_elbo_rt")) return rc;
    if (int rc = gemm_m(a.RT, a.q, a.Atab, S)) return rc;
    hipLaunchKernelGGL(k_elbo_afix, dim3(1024), dim3(256), 0, st, a);
    if (int rc = finish_launch("k_elbo_afix")) return rc;
  } else {
    hipLaunchKernelGGL(k_elbo_atab, dim3((S + LRB - 1) / LRB, B), dim3(256), 0, st, a);
    if (int rc = finish_launch("k_elbo_atab")) return rc;
  }
  static const bool rows8 = [] { const char* e = getenv("CTDD_ELBO_ROWS8"); return e && e[0] == '1'; }();     // (A/B: the workgroup-per-8-rows passes)
  const bool wave_rows = mfma && S == 256 && !rows8;
  const dim3 wg((D + 4 * RV_ROWS - 1) / (4 * RV_ROWS), B);
  if (wave_rows) {
    hipLaunchKernelGGL(k_elbo_rowsv<0>, wg, dim3(256), 0, st, a);               // rvec -> grad buffer; reg / nll row sums
    if (int rc = finish_launch("k_elbo_rowsv<0>")) return rc;
    if (int rc = gemm(a.grad, a.qT, a.u)) return rc;
    hipLaunchKernelGGL(k_elbo_rowsv<1>, wg, dim3(256), 0, st, a);
    if (int rc = finish_launch("k_elbo_rowsv<1>")) return rc;
  } else if (mfma) {
    hipLaunchKernelGGL(k_elbo_fwd<1>, rg, dim3(256), 0, st, a);                 // rvec -> grad buffer; reg / nll row sums
    if (int rc = finish_launch("k_elbo_fwd<1>")) return rc;
    if (int rc = gemm(a.grad, a.qT, a.u)) return rc;                            // u[row][s] = sum_s0 rvec[row][s0] qT[s][s0]
    hipLaunchKernelGGL(k_elbo_fwd<2>, rg, dim3(256), 0, st, a);
    if (int rc = finish_launch("k_elbo_fwd<2>")) return rc;
  } else {
    hipLaunchKernelGGL(k_elbo_fwd<0>, rg, dim3(256), 0, st, a);
    if (int rc = finish_launch("k_elbo_fwd")) return rc;
  }
  hipLaunchKernelGGL(k_elbo_sample_sums, dim3(B), dim3(256), 0, st, a, sums);
  if (int rc = finish_launch("k_elbo_sample_sums")) return rc;
  hipLaunchKernelGGL(k_elbo_reduce, dim3(1), dim3(256), 0, st, a, (const double*)sums);
  if (int rc = finish_launch("k_elbo_reduce")) return rc;
  if (wave_rows) {
    hipLaunchKernelGGL(k_elbo_rowsv<2>, wg, dim3(256), 0, st, a);               // G -> grad buffer
    if (int rc = finish_launch("k_elbo_rowsv<2>")) return rc;
    if (int rc = gemm(a.grad, a.q, a.u)) return rc;                             // dr = G q^T (over u)
    hipLaunchKernelGGL(k_elbo_rowsv<3>, wg, dim3(256), 0, st, a);
    return finish_launch("k_elbo_rowsv<3>");
  }
  if (mfma) {
    hipLaunchKernelGGL(k_elbo_bwd<1>, rg, dim3(256), 0, st, a);                 // G -> grad buffer
    if (int rc = finish_launch("k_elbo_bwd<1>")) return rc;
    if (int rc = gemm(a.grad, a.q, a.u)) return rc;                             // dr[row][s0] = sum_s G[row][s] q[s0][s]   (over u)
    hipLaunchKernelGGL(k_elbo_bwd<2>, rg, dim3(256), 0, st, a);
    return finish_launch("k_elbo_bwd<2>");
  }
  hipLaunchKernelGGL(k_elbo_bwd<0>, rg, dim3(256), 0, st, a);
  return finish_launch("k_elbo_bwd");
}

// ================================================================== K12b: ScoreElbo (direct logits), value + d/dlogits
// Reference lib/losses/losses.py:1255-1500: the CT-ELBO with SDDM ratios exp(ll_all - ll_xt).  Per row (b,d),
// ll = log_softmax(l), x = x~_bd, xr = reg_x_bd (= x~ with one forward pass, x_t with two), dd[s] = ll[s] - ll[x]:
//   reg_row   = sum_s e^{dd[s]} [s != xr] R[s,xr]
//   outer_row = sum_s Wt[s] dd[s],  Wt[s] = [s != x] R[s,x] q[x0,s] / (q[x0,x] + eps);  norm_row as in K11
//   loss = mean_b(-sum_d outer / sum_d norm) + mean_b sum_d reg + (nll_weight / B) sum_{b,d} -ll[x]
// g[s] = d loss / d dd[s] = e^{dd[s]} [s != xr] R[s,xr] / B + c_b Wt[s]  (s != x),  c_b = -1 / (B norm_b);
// g[x] = -sum_{s != x} g[s] - nll_weight / B;  dl[j] = g[j] - p[j] sum_s g[s].
// One wave per row; forward rows -> per-sample normalisers (k_elbo_reduce) -> backward rows.
namespace ctdd {

__global__ __launch_bounds__(256) void k_selbo_fwd(const ElboArgs a, const int32_t* __restrict__ regx) {
  const int lane = threadIdx.x & 63, S = a.S, D = a.D;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)a.B * D) return;
  const int b = (int)(row / D);
  const float* l = a.logits + (size_t)row * S;
  const float* q = a.q + (size_t)b * S * S;
  const float* R = a.R + (size_t)b * S * S;
  const int x = min(max(a.xt[row], 0), S - 1), x0 = min(max(a.x0[row], 0), S - 1), xr = min(max(regx[row], 0), S - 1);
  float L = 0.0f;
  if (!a.ll_in) {
    float m = -INFINITY;
    for (int s = lane; s < S; s += 64) m = fmaxf(m, l[s]);
    m = lwave_max(m);
    float z = 0.0f;
    for (int s = lane; s < S; s += 64) z += expf(l[s] - m);
    L = m + logf(lwave_sum(z));
  }
  const float llx = l[x] - L;
  const float qx0xt = q[(size_t)x0 * S + x] + a.eps, bsum = a.base_sum[b], rsx = -R[(size_t)x * S + x];
  float reg = 0.0f, outer = 0.0f, norm = 0.0f;
  for (int s = lane; s < S; s += 64) {
    const float dd = (l[s] - L) - llx;
    if (s != xr) reg += expf(dd) * R[(size_t)s * S + xr];
    if (s != x) {
      const float orate = R[(size_t)s * S + x], qx0 = q[(size_t)x0 * S + s];
      const float Z = bsum - rsx - R[(size_t)s * S + s];
      outer += orate * (qx0 / qx0xt) * dd;
      norm += orate * qx0 / (Z * qx0xt);
    }
  }
  reg = lwave_sum(reg); outer = lwave_sum(outer); norm = lwave_sum(norm);
  if (lane == 0) {
    double* dst = a.rows + (size_t)row * 4;
    dst[0] = outer; dst[1] = norm; dst[2] = reg; dst[3] = -llx;
  }
}

__global__ __launch_bounds__(256) void k_selbo_bwd(const ElboArgs a, const int32_t* __restrict__ regx) {
  const int lane = threadIdx.x & 63, S = a.S, D = a.D;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)a.B * D) return;
  const int b = (int)(row / D);
  const float* l = a.logits + (size_t)row * S;
  const float* q = a.q + (size_t)b * S * S;
  const float* R = a.R + (size_t)b * S * S;
  float* gr = a.grad + (size_t)row * S;
  const int x = min(max(a.xt[row], 0), S - 1), x0 = min(max(a.x0[row], 0), S - 1), xr = min(max(regx[row], 0), S - 1);
  const float cb = a.cb[b], invB = a.elbo_scale / (float)a.B;
  float L = 0.0f;
  if (!a.ll_in) {
    float m = -INFINITY;
    for (int s = lane; s < S; s += 64) m = fmaxf(m, l[s]);
    m = lwave_max(m);
    float z = 0.0f;
    for (int s = lane; s < S; s += 64) z += expf(l[s] - m);
    L = m + logf(lwave_sum(z));
  }
  const float llx = l[x] - L;
  const float qx0xt = q[(size_t)x0 * S + x] + a.eps;
  auto gd = [&](int s) {                              // d loss / d dd[s], s != x
    const float dd = (l[s] - L) - llx;
    float g = cb * R[(size_t)s * S + x] * (q[(size_t)x0 * S + s] / qx0xt);
    if (s != xr) g += invB * expf(dd) * R[(size_t)s * S + xr];
    return g;
  };
  float gs = 0.0f;
  for (int s = lane; s < S; s += 64)
    if (s != x) gs += gd(s);
  gs = lwave_sum(gs);
  const float gx = -gs - a.nll_scale;                  // nll_scale = nll_weight / B here
  const float gsum = gs + gx;
  for (int s = lane; s < S; s += 64) {
    const float p = expf(l[s] - L);
    gr[s] = (s == x ? gx : gd(s)) - (a.ll_in ? 0.0f : p * gsum);
  }
}

}  // namespace ctdd

static int score_elbo_impl(const float* logits, const int32_t* x0, const int32_t* x_tilde, const int32_t* reg_x, const float* qt0,
                           const float* rate, int B, int D, int S, float eps, float nll_scale, void* scratch, float* grad_logits,
                           float* out_loss, int ll_in, void* stream);
extern "C" int ctdd_score_elbo_loss(const float* logits, const int32_t* x0, const int32_t* x_tilde, const int32_t* reg_x,
                                    const float* qt0, const float* rate, int B, int D, int S, float eps, float nll_scale,
                                    void* scratch, float* grad_logits, float* out_loss, void* stream) {
  return score_elbo_impl(logits, x0, x_tilde, reg_x, qt0, rate, B, D, S, eps, nll_scale, scratch, grad_logits, out_loss, 0, stream);
}
// the same objective on ll_all (reverse logit types): value and d loss / d ll_all
extern "C" int ctdd_score_elbo_loss_ll(const float* ll_all, const int32_t* x0, const int32_t* x_tilde, const int32_t* reg_x,
                                       const float* qt0, const float* rate, int B, int D, int S, float eps, float nll_scale,
                                       void* scratch, float* grad_ll, float* out_loss, void* stream) {
  return score_elbo_impl(ll_all, x0, x_tilde, reg_x, qt0, rate, B, D, S, eps, nll_scale, scratch, grad_ll, out_loss, 1, stream);
}
static int score_elbo_impl(const float* logits, const int32_t* x0, const int32_t* x_tilde, const int32_t* reg_x, const float* qt0,
                           const float* rate, int B, int D, int S, float eps, float nll_scale, void* scratch, float* grad_logits,
                           float* out_loss, int ll_in, void* stream) {
  CTDD_REQUIRE(logits && x0 && x_tilde && reg_x && qt0 && rate && scratch && grad_logits && out_loss, CTDD_EINVAL, "score-elbo: null buffer");
  CTDD_REQUIRE(B > 0 && D > 0 && S >= 2 && S <= 256, CTDD_ERANGE, "score-elbo: B=%d D=%d S=%d (S <= 256)", B, D, S);
  auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
  unsigned char* sp = (unsigned char*)scratch;          // same layout as ctdd_ctelbo_scratch_bytes
  ElboArgs a;
  a.logits = logits; a.x0 = x0; a.xt = x_tilde; a.q = qt0; a.qT = qt0; a.R = rate;
  a.B = B; a.D = D; a.S = S; a.eps = eps; a.elbo_scale = 1.0f; a.nll_scale = nll_scale; a.reg_scale = 1.0f;
  a.Atab = (float*)sp; sp += al((int64_t)B * S * S * 4);
  a.u = (float*)sp; sp += al((int64_t)B * D * S * 4);
  a.rows = (double*)sp; sp += al((int64_t)B * D * 32);
  a.base_sum = (float*)sp; sp += al((int64_t)B * 4);
  a.cb = (float*)sp; sp += al((int64_t)B * 4);
  double* sums = (double*)sp;
  a.RT = nullptr;                                       // (k_elbo_atab runs its first column block only here)
  a.grad = grad_logits; a.out_loss = out_loss; a.ll_in = ll_in;
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * D;
  hipLaunchKernelGGL(k_elbo_atab, dim3(1, B), dim3(256), 0, st, a);          // (first column block only: base_sum; its A rows are unused scratch)
  if (int rc = finish_launch("k_elbo_atab")) return rc;
  hipLaunchKernelGGL(k_selbo_fwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a, reg_x);
  if (int rc = finish_launch("k_selbo_fwd")) return rc;
  hipLaunchKernelGGL(k_elbo_sample_sums, dim3(B), dim3(256), 0, st, a, sums);
  if (int rc = finish_launch("k_elbo_sample_sums")) return rc;
  hipLaunchKernelGGL(k_elbo_reduce, dim3(1), dim3(256), 0, st, a, (const double*)sums);
  if (int rc = finish_launch("k_elbo_reduce")) return rc;
  hipLaunchKernelGGL(k_selbo_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a, reg_x);
  return finish_launch("k_selbo_bwd");
}

// ================================================================== backward of get_logprob_with_logits for the reverse logit types
// (reference lib/models/model_utils.py:42-56 under autograd): with p = softmax(l) and the sample's q = q_{t|0},
//   reverse_prob      ll[s] = log(acc[s] + 1e-35),  acc = p @ q
//   reverse_logscale  ll[s] = logsumexp_s0(log p[s0] + log q[s0][s], q <= 1e-35 -> -1e9) = log(p @ q'), q' = q [q > 1e-35]
// Given g = d loss / d ll:  ga[s] = g[s] / (acc[s] + 1e-35)   (logscale: g / acc' where acc' > 0, else 0)
//   dp[s0] = sum_s ga[s] q[s0][s]  (= ga @ q^T: the transposed table qT is streamed so that both contractions read rows),
//   dl[j]  = p[j] (dp[j] - sum_s0 p[s0] dp[s0])  [+ nll_scale (p[j] - [j = x0]): the cross-entropy term of CatRMNLL, which is
//   on the raw logits, losses.py:1240-1242; its value -log p[x0] goes to ce_rows].
// Workgroup = 8 rows (dimensions) of one sample, thread s <-> state (S <= 256): each streamed table element feeds 8 FMAs.
namespace ctdd {

struct LpbArgs {
  const float* logits; const float* q; const float* qT; const float* dll; const int32_t* x0;
  int B, D, S, logit_type; float nll_scale;
  float* grad; double* ce_rows;
};
__global__ __launch_bounds__(256) void k_logprob_bwd(const LpbArgs a) {
  __shared__ __attribute__((aligned(16))) float pv[256 * LRB];      // [s0][r] p, then [s][r] ga
  __shared__ float red[4 * LRB];
  const int S = a.S, b = blockIdx.y, d0 = blockIdx.x * LRB, s = threadIdx.x;
  const bool on = s < S;
  const float* q = a.q + (size_t)b * S * S;
  const float* qT = a.qT + (size_t)b * S * S;
  const bool logscale = a.logit_type == 2;
  float l[LRB], p[LRB], m[LRB], z[LRB];
  size_t rowoff[LRB];
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    const int d = min(d0 + r, a.D - 1);
    rowoff[r] = ((size_t)b * a.D + d) * S;
    l[r] = on ? a.logits[rowoff[r] + s] : -INFINITY;
  }
  block_max8(l, red, m);
#pragma unroll
  for (int r = 0; r < LRB; ++r) p[r] = on ? expf(l[r] - m[r]) : 0.0f;
  block_sum8(p, red, z);
#pragma unroll
  for (int r = 0; r < LRB; ++r) { p[r] = p[r] / z[r]; pv[s * LRB + r] = p[r]; }
  __syncthreads();
  // acc[r] = sum_s0 p_r[s0] q[s0][s]
  float acc[LRB];
#pragma unroll
  for (int r = 0; r < LRB; ++r) acc[r] = 0.0f;
  if (on) {
    for (int s0 = 0; s0 < S; ++s0) {
      float qv = q[(size_t)s0 * S + s];
      if (logscale && qv <= 1e-35f) qv = 0.0f;
      const float4 p0 = *(const float4*)(pv + s0 * LRB), p1 = *(const float4*)(pv + s0 * LRB + 4);
      acc[0] = fmaf(p0.x, qv, acc[0]); acc[1] = fmaf(p0.y, qv, acc[1]); acc[2] = fmaf(p0.z, qv, acc[2]); acc[3] = fmaf(p0.w, qv, acc[3]);
      acc[4] = fmaf(p1.x, qv, acc[4]); acc[5] = fmaf(p1.y, qv, acc[5]); acc[6] = fmaf(p1.z, qv, acc[6]); acc[7] = fmaf(p1.w, qv, acc[7]);
    }
  }
  __syncthreads();                                                    // pv is read: reuse it for ga
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    float ga = 0.0f;
    if (on && d0 + r < a.D) {
      const float g = a.dll[rowoff[r] + s];
      ga = logscale ? (acc[r] > 0.0f ? g / acc[r] : 0.0f) : g / (acc[r] + 1e-35f);
    }
    pv[s * LRB + r] = ga;
  }
  __syncthreads();
  // dp[r] (this thread: s0 = s) = sum_s' ga_r[s'] qT[s'][s0]
  float dp[LRB];
#pragma unroll
  for (int r = 0; r < LRB; ++r) dp[r] = 0.0f;
  if (on) {
    for (int sp = 0; sp < S; ++sp) {
      float qv = qT[(size_t)sp * S + s];
      if (logscale && qv <= 1e-35f) qv = 0.0f;
      const float4 g0 = *(const float4*)(pv + sp * LRB), g1 = *(const float4*)(pv + sp * LRB + 4);
      dp[0] = fmaf(g0.x, qv, dp[0]); dp[1] = fmaf(g0.y, qv, dp[1]); dp[2] = fmaf(g0.z, qv, dp[2]); dp[3] = fmaf(g0.w, qv, dp[3]);
      dp[4] = fmaf(g1.x, qv, dp[4]); dp[5] = fmaf(g1.y, qv, dp[5]); dp[6] = fmaf(g1.z, qv, dp[6]); dp[7] = fmaf(g1.w, qv, dp[7]);
    }
  }
  float pd[LRB], dot[LRB];
#pragma unroll
  for (int r = 0; r < LRB; ++r) pd[r] = p[r] * dp[r];
  block_sum8(pd, red, dot);
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    if (!(on && d0 + r < a.D)) continue;
    float v = p[r] * (dp[r] - dot[r]);
    if (a.x0) {
      const int x0 = min(max(a.x0[(size_t)b * a.D + d0 + r], 0), S - 1);
      v += a.nll_scale * (p[r] - (s == x0 ? 1.0f : 0.0f));
      if (s == x0 && a.ce_rows) a.ce_rows[(size_t)b * a.D + d0 + r] = (double)a.nll_scale * (double)(-(l[r] - m[r] - logf(z[r])));
    }
    a.grad[rowoff[r] + s] = v;
  }
}


// ---- reverse_prob log-probabilities and their backward with the S x S contractions on the exact-fp32 matrix instruction
// (S % 32 == 0, per-sample tables: the training objectives of the MNIST / CIFAR hollow configs).  The generic row kernel
// (ctdd_logprob: k_rows) runs them as fp32 FMA chains with the table streamed per row: 2.5 ms for 32 x 784 rows at S = 256.
//   forward   P = softmax(l)  ->  acc = P q  (k_bgemm_f32, W = q^T)  ->  ll_all = log(acc + 1e-35), ll_xt = ll_all[x]
//   backward  ga = dll / (acc + 1e-35) = dll exp(-ll_all)  ->  dp = ga q^T  (k_bgemm_f32, W = q)  ->
//             dl[j] = p[j] (dp[j] - sum p dp)  [+ nll_scale (p[j] - [j = x0])]
// Row kernels: 8 rows per workgroup, thread <-> state.
__global__ __launch_bounds__(256) void k_lp_softmax(const float* __restrict__ logits, int64_t rows, int S, float* __restrict__ P) {
  __shared__ float red[4 * LRB];
  const int t = threadIdx.x;
  const bool on = t < S;
  float l[LRB], m[LRB], e[LRB], z[LRB];
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    const int64_t row = min((int64_t)blockIdx.x * LRB + r, rows - 1);
    l[r] = on ? logits[(size_t)row * S + t] : -INFINITY;
  }
  block_max8(l, red, m);
#pragma unroll
  for (int r = 0; r < LRB; ++r) e[r] = on ? expf(l[r] - m[r]) : 0.0f;
  block_sum8(e, red, z);
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    const int64_t row = (int64_t)blockIdx.x * LRB + r;
    if (on && row < rows) P[(size_t)row * S + t] = e[r] / z[r];
  }
}
// in place: acc -> ll_all = log(acc + 1e-35); ll_xt[row] = ll_all[row][x]
__global__ __launch_bounds__(256) void k_lp_log(float* __restrict__ ll, const int32_t* __restrict__ x, int64_t rows, int S, float* __restrict__ ll_xt) {
  const int t = threadIdx.x;
  if (t >= S) return;
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    const int64_t row = (int64_t)blockIdx.x * LRB + r;
    if (row >= rows) break;
    const float v = logf(ll[(size_t)row * S + t] + 1e-35f);
    ll[(size_t)row * S + t] = v;
    if (ll_xt && t == min(max(x[row], 0), S - 1)) ll_xt[row] = v;
  }
}
__global__ __launch_bounds__(256) void k_lp_ga(const float* __restrict__ dll, const float* __restrict__ ll, int64_t n4, float* __restrict__ ga) {
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n4; v += (int64_t)gridDim.x * 256) {
    const float4 g = *(const float4*)(dll + v * 4), l = *(const float4*)(ll + v * 4);
    *(float4*)(ga + v * 4) = make_float4(g.x * expf(-l.x), g.y * expf(-l.y), g.z * expf(-l.z), g.w * expf(-l.w));
  }
}
__global__ __launch_bounds__(256) void k_lp_final(const float* __restrict__ logits, const float* __restrict__ dp, const int32_t* __restrict__ x0,
                                                  int64_t rows, int S, float nll_scale, float* __restrict__ grad, double* __restrict__ ce_rows) {
  __shared__ float red[4 * LRB];
  const int t = threadIdx.x;
  const bool on = t < S;
  float l[LRB], m[LRB], p[LRB], z[LRB], d[LRB], pd[LRB], dot[LRB];
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    const int64_t row = min((int64_t)blockIdx.x * LRB + r, rows - 1);
    l[r] = on ? logits[(size_t)row * S + t] : -INFINITY;
    d[r] = on ? dp[(size_t)row * S + t] : 0.0f;
  }
  block_max8(l, red, m);
#pragma unroll
  for (int r = 0; r < LRB; ++r) p[r] = on ? expf(l[r] - m[r]) : 0.0f;
  block_sum8(p, red, z);
#pragma unroll
  for (int r = 0; r < LRB; ++r) { p[r] = p[r] / z[r]; pd[r] = p[r] * d[r]; }
  block_sum8(pd, red, dot);
#pragma unroll
  for (int r = 0; r < LRB; ++r) {
    const int64_t row = (int64_t)blockIdx.x * LRB + r;
    if (!on || row >= rows) continue;
    float v = p[r] * (d[r] - dot[r]);
    if (x0) {
      const int xz = min(max(x0[row], 0), S - 1);
      v += nll_scale * (p[r] - (t == xz ? 1.0f : 0.0f));
      if (t == xz && ce_rows) ce_rows[row] = (double)nll_scale * (double)(-(l[r] - m[r] - logf(z[r])));
    }
    grad[(size_t)row * S + t] = v;
  }
}
template <typename F>
static int launch_bgemm(int S, dim3 gg, hipStream_t st, const float* Am, const float* Wm, float* Cm, int M) {
  switch (S / 32) {
    case 1: hipLaunchKernelGGL(k_bgemm_f32<1>, gg, dim3(256), 0, st, Am, Wm, Cm, M); break;
    case 2: hipLaunchKernelGGL(k_bgemm_f32<2>, gg, dim3(256), 0, st, Am, Wm, Cm, M); break;
    case 3: hipLaunchKernelGGL(k_bgemm_f32<3>, gg, dim3(256), 0, st, Am, Wm, Cm, M); break;
    case 4: hipLaunchKernelGGL(k_bgemm_f32<4>, gg, dim3(256), 0, st, Am, Wm, Cm, M); break;
    case 5: hipLaunchKernelGGL(k_bgemm_f32<5>, gg, dim3(256), 0, st, Am, Wm, Cm, M); break;
    case 6: hipLaunchKernelGGL(k_bgemm_f32<6>, gg, dim3(256), 0, st, Am, Wm, Cm, M); break;
    case 7: hipLaunchKernelGGL(k_bgemm_f32<7>, gg, dim3(256), 0, st, Am, Wm, Cm, M); break;
    default: hipLaunchKernelGGL(k_bgemm_f32<8>, gg, dim3(256), 0, st, Am, Wm, Cm, M); break;
  }
  return finish_launch("k_bgemm_f32");
}

}  // namespace ctdd

// scratch: (B, D, S) fp32 (softmax)
extern "C" int ctdd_logprob_rp_mfma(const float* logits, const int32_t* x, const float* qt0T, int B, int D, int S, float* scratch,
                                    float* out_ll_all, float* out_ll_xt, void* stream) {
  CTDD_REQUIRE(logits && qt0T && scratch && out_ll_all && (x || !out_ll_xt), CTDD_EINVAL, "logprob (matrix cores): null buffer");
  CTDD_REQUIRE(B > 0 && D > 0 && S >= 32 && S <= 256 && S % 32 == 0, CTDD_ERANGE, "logprob (matrix cores): B=%d D=%d S=%d (S %% 32 == 0, <= 256)", B, D, S);
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * D;
  const unsigned rb = (unsigned)((rows + LRB - 1) / LRB);
  hipLaunchKernelGGL(k_lp_softmax, dim3(rb), dim3(256), 0, st, logits, rows, S, scratch);
  if (int rc = finish_launch("k_lp_softmax")) return rc;
  if (int rc = launch_bgemm<void>(S, dim3((D + 127) / 128, B), st, scratch, qt0T, out_ll_all, D)) return rc;   // acc[row][s] = sum_s0 P[row][s0] qT[s][s0]
  hipLaunchKernelGGL(k_lp_log, dim3(rb), dim3(256), 0, st, out_ll_all, x, rows, S, out_ll_xt);
  return finish_launch("k_lp_log");
}
// backward of the above given ll_all (its output) and dll = d loss / d ll_all; scratch: 2 x (B, D, S) fp32 (ga | dp)
extern "C" int ctdd_logprob_rp_bwd_mfma(const float* logits, const float* qt0, const float* ll_all, const float* dll, const int32_t* x0,
                                        float nll_scale, int B, int D, int S, float* scratch, float* grad_logits, double* ce_rows,
                                        float* out_ce, void* stream) {
  CTDD_REQUIRE(logits && qt0 && ll_all && dll && scratch && grad_logits, CTDD_EINVAL, "logprob bwd (matrix cores): null buffer");
  CTDD_REQUIRE(B > 0 && D > 0 && S >= 32 && S <= 256 && S % 32 == 0, CTDD_ERANGE, "logprob bwd (matrix cores): B=%d D=%d S=%d", B, D, S);
  CTDD_REQUIRE(!x0 || (ce_rows && out_ce), CTDD_EINVAL, "logprob bwd: the cross-entropy term needs ce_rows and out_ce");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * D, n = rows * S;
  float* ga = scratch;
  float* dp = scratch + n;
  int64_t g = (n / 4 + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_lp_ga, dim3((unsigned)g), dim3(256), 0, st, dll, ll_all, n / 4, ga);
  if (int rc = finish_launch("k_lp_ga")) return rc;
  if (int rc = launch_bgemm<void>(S, dim3((D + 127) / 128, B), st, ga, qt0, dp, D)) return rc;                 // dp[row][s0] = sum_s ga[row][s] q[s0][s]
  hipLaunchKernelGGL(k_lp_final, dim3((unsigned)((rows + LRB - 1) / LRB)), dim3(256), 0, st, logits, (const float*)dp, x0, rows, S, nll_scale,
                     grad_logits, ce_rows);
  if (int rc = finish_launch("k_lp_final")) return rc;
  if (x0) {
    hipLaunchKernelGGL(k_sum_rows, dim3(1), dim3(256), 0, st, (const double*)ce_rows, rows, out_ce);
    return finish_launch("k_sum_rows");
  }
  return CTDD_OK;
}

extern "C" int ctdd_logprob_bwd(int logit_type, const float* logits, const float* qt0, const float* qt0T, const float* dll,
                                const int32_t* x0, float nll_scale, int B, int D, int S, float* grad_logits, double* ce_rows,
                                float* out_ce, void* stream) {
  CTDD_REQUIRE(logits && qt0 && qt0T && dll && grad_logits, CTDD_EINVAL, "logprob bwd: null buffer");
  CTDD_REQUIRE(logit_type == 1 || logit_type == 2, CTDD_EINVAL, "logprob bwd: logit_type %d (1 reverse_prob, 2 reverse_logscale)", logit_type);
  CTDD_REQUIRE(B > 0 && D > 0 && S >= 2 && S <= 256, CTDD_ERANGE, "logprob bwd: B=%d D=%d S=%d (S <= 256)", B, D, S);
  CTDD_REQUIRE(!x0 || (ce_rows && out_ce), CTDD_EINVAL, "logprob bwd: the cross-entropy term needs ce_rows and out_ce");
  LpbArgs a = {logits, qt0, qt0T, dll, x0, B, D, S, logit_type, nll_scale, grad_logits, ce_rows};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_logprob_bwd, dim3((D + LRB - 1) / LRB, B), dim3(256), 0, st, a);
  if (int rc = finish_launch("k_logprob_bwd")) return rc;
  if (x0) {
    hipLaunchKernelGGL(k_sum_rows, dim3(1), dim3(256), 0, st, (const double*)ce_rows, (int64_t)B * D, out_ce);
    return finish_launch("k_sum_rows");
  }
  return CTDD_OK;
}
