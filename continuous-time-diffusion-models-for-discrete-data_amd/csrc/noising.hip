// noising.hip -- K2 x_t ~ q_{t|0}(.|x0) and K3 the one-jump neighbour x~ (forward noising of the
// losses: lib/losses/losses.py:34-101 CTElbo, 859-874 CatRM, 1211-1225 CatRMNLL, 1526-1593 NLL,
// 1807-1874 CTElboLambda).  Draws are ATen's one-sample multinomial: argmax_s(p_s / E_s), E~Exp(1),
// first index on ties -- bit-exact given (probs, E): IEEE fp32 division, no fast-math.
#include "common.hpp"

namespace ctdd {

// G lanes per (b,d) row, element s = li + k*G.
__global__ __launch_bounds__(256) void k_noise_categorical(const float* __restrict__ probs,
                                                           const int32_t* __restrict__ tidx,
                                                           const int32_t* __restrict__ x0,
                                                           const float* __restrict__ E, uint64_t seed,
                                                           uint64_t offset, int B, int D, int S, int G,
                                                           int32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & (G - 1), gi = lane / G;
  const int64_t R = (int64_t)B * D;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * (WAVE / G) + gi;
  const bool live = row < R;
  const int64_t rowc = live ? row : R - 1;
  const int b = (int)(rowc / D);
  const int tbl = tidx ? tidx[b] : b;
  const int xv = min(max(x0[rowc], 0), S - 1);
  const float* prow = probs + ((size_t)tbl * S + xv) * S;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int s = li; s < S; s += G) {
    float Ev;
    if (E) Ev = E[(size_t)rowc * S + s];
    else Ev = -logf(u01(philox_row(seed, offset, (uint64_t)rowc, (uint32_t)s).x));
    const float v = prow[s] / Ev;
    if (v > best) { best = v; bi = s; }   // s ascending: strict > keeps the first maximum
  }
  for (int m = G >> 1; m >= 1; m >>= 1) {
    const float ov = __shfl_xor(best, m, WAVE);
    const int oi = __shfl_xor(bi, m, WAVE);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (live && li == 0) out[row] = bi;
}

// block-wide (value,index) argmax, first index on ties; result valid in every thread
__device__ inline void block_argmax(float& v, int& i, float* sv, int* si) {
  for (int m = 32; m >= 1; m >>= 1) {
    const float ov = __shfl_xor(v, m, WAVE);
    const int oi = __shfl_xor(i, m, WAVE);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sv[wave] = v; si[wave] = i; }
  __syncthreads();
  v = sv[0]; i = si[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
    if (sv[w] > v || (sv[w] == v && si[w] < i)) { v = sv[w]; i = si[w]; }
}
__device__ inline float block_sum(float v, float* sv) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.0f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sv[w];
  return t;
}
__device__ inline float block_max(float v, float* sv) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = sv[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = fmaxf(t, sv[w]);
  return t;
}

// one workgroup per batch row b
__global__ __launch_bounds__(256) void k_xtilde(const float* __restrict__ rate, const int32_t* __restrict__ tidx,
                                                const int32_t* __restrict__ x_t, const float* __restrict__ E_dim,
                                                const float* __restrict__ E_val, uint64_t seed, uint64_t offset,
                                                int B, int D, int S, int32_t* __restrict__ out_dims,
                                                int32_t* __restrict__ out_newval, int32_t* __restrict__ out_xt) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int b = blockIdx.x;
  const int tbl = tidx ? tidx[b] : b;
  const float* R = rate + (size_t)tbl * S * S;
  const int32_t* xr = x_t + (size_t)b * D;
  // weights w_d = sum_{s != x_d} R[x_d][s]; Categorical(probs=w) normalises by sum_d w_d.  The sum depends on the state only:
  // one table rs[x] per sample (thread = state, its row streamed in s order -- the same fp32 chain the per-dimension loop
  // ran 2 D / 256 times per thread: 231 us -> ~20 us at D = 784, S = 256), looked up per dimension.
  __shared__ float rs[CTDD_MAX_S];
  for (int xs = threadIdx.x; xs < S; xs += 256) {
    const float* rr = R + (size_t)xs * S;
    float w = 0.0f;
    if ((S & 3) == 0) {
      for (int s = 0; s < S; s += 4) {
        const float4 v = *(const float4*)(rr + s);
        w += (s == xs) ? 0.0f : v.x; w += (s + 1 == xs) ? 0.0f : v.y; w += (s + 2 == xs) ? 0.0f : v.z; w += (s + 3 == xs) ? 0.0f : v.w;
      }
    } else {
      for (int s = 0; s < S; ++s) w += (s == xs) ? 0.0f : rr[s];
    }
    rs[xs] = w;
  }
  __syncthreads();
  float wsum = 0.0f;
  for (int d = threadIdx.x; d < D; d += 256) wsum += rs[min(max(xr[d], 0), S - 1)];
  const float W = block_sum(wsum, sv);
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float w = rs[min(max(xr[d], 0), S - 1)];
    float Ev;
    if (E_dim) Ev = E_dim[(size_t)b * D + d];
    else Ev = -logf(u01(philox_row(seed, offset, (uint64_t)b, (uint32_t)d).x));
    const float v = (w / W) / Ev;
    if (v > best) { best = v; bi = d; }
  }
  block_argmax(best, bi, sv, si);
  const int dim = bi;
  const int xv = min(max(xr[dim], 0), S - 1);
  // new value ~ Categorical(logits = where(row<=0,-1e9,log row)), row = R[xv][:] with own state 0
  const float* row = R + (size_t)xv * S;
  float mx = -INFINITY;
  for (int s = threadIdx.x; s < S; s += 256) {
    const float r = (s == xv) ? 0.0f : row[s];
    mx = fmaxf(mx, r <= 0.0f ? -1e9f : logf(r));
  }
  mx = block_max(mx, sv);
  float se = 0.0f;
  for (int s = threadIdx.x; s < S; s += 256) {
    const float r = (s == xv) ? 0.0f : row[s];
    se += expf((r <= 0.0f ? -1e9f : logf(r)) - mx);
  }
  se = block_sum(se, sv);
  const float lse = mx + logf(se);
  float m2 = -INFINITY;
  for (int s = threadIdx.x; s < S; s += 256) {
    const float r = (s == xv) ? 0.0f : row[s];
    m2 = fmaxf(m2, (r <= 0.0f ? -1e9f : logf(r)) - lse);
  }
  m2 = block_max(m2, sv);
  float s2 = 0.0f;
  for (int s = threadIdx.x; s < S; s += 256) {
    const float r = (s == xv) ? 0.0f : row[s];
    s2 += expf((r <= 0.0f ? -1e9f : logf(r)) - lse - m2);
  }
  s2 = block_sum(s2, sv);
  best = -INFINITY;
  bi = 0x7fffffff;
  for (int s = threadIdx.x; s < S; s += 256) {
    const float r = (s == xv) ? 0.0f : row[s];
    const float p = expf((r <= 0.0f ? -1e9f : logf(r)) - lse - m2) / s2;
    float Ev;
    if (E_val) Ev = E_val[(size_t)b * S + s];
    else Ev = -logf(u01(philox_row(seed, offset, (uint64_t)b, 0x40000000u + (uint32_t)s).x));
    const float v = p / Ev;
    if (v > best) { best = v; bi = s; }
  }
  block_argmax(best, bi, sv, si);
  for (int d = threadIdx.x; d < D; d += 256) out_xt[(size_t)b * D + d] = (d == dim) ? bi : xr[d];
  if (threadIdx.x == 0) {
    if (out_dims) out_dims[b] = dim;
    if (out_newval) out_newval[b] = bi;
  }
}

}  // namespace ctdd
using namespace ctdd;

extern "C" int ctdd_noise_categorical(const float* probs, const int32_t* tidx, const int32_t* x0,
                                      const float* E, uint64_t seed, uint64_t offset, int B, int D, int S,
                                      int32_t* out_xt, void* stream) {
  CTDD_REQUIRE(probs && x0 && out_xt, CTDD_EINVAL, "null probs/x0/out");
  CTDD_REQUIRE(B > 0 && D > 0, CTDD_EINVAL, "B=%d D=%d must be positive", B, D);
  CTDD_REQUIRE(S >= 2 && S <= CTDD_MAX_S, CTDD_ERANGE, "S=%d outside [2,%d]", S, CTDD_MAX_S);
  int G = 1;
  while (G < S && G < 64) G <<= 1;
  const int rows_per_wg = 4 * (64 / G);
  const int64_t R = (int64_t)B * D, grid = (R + rows_per_wg - 1) / rows_per_wg;
  CTDD_REQUIRE(grid < (1ll << 31), CTDD_ERANGE, "too many rows");
  hipLaunchKernelGGL(k_noise_categorical, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, probs, tidx,
                     x0, E, seed, offset, B, D, S, G, out_xt);
  return finish_launch("k_noise_categorical");
}

extern "C" int ctdd_xtilde_sample(const float* rate, const int32_t* tidx, const int32_t* x_t,
                                  const float* E_dim, const float* E_val, uint64_t seed, uint64_t offset,
                                  int B, int D, int S, int32_t* out_dims, int32_t* out_newval,
                                  int32_t* out_xtilde, void* stream) {
  CTDD_REQUIRE(rate && x_t && out_xtilde, CTDD_EINVAL, "null rate/x_t/out");
  CTDD_REQUIRE((E_dim == nullptr) == (E_val == nullptr), CTDD_EINVAL, "E_dim and E_val must both be given or both null");
  CTDD_REQUIRE(B > 0 && D > 0, CTDD_EINVAL, "B=%d D=%d must be positive", B, D);
  CTDD_REQUIRE(S >= 2 && S <= CTDD_MAX_S, CTDD_ERANGE, "S=%d outside [2,%d]", S, CTDD_MAX_S);
  hipLaunchKernelGGL(k_xtilde, dim3(B), dim3(256), 0, (hipStream_t)stream, rate, tidx, x_t, E_dim, E_val, seed,
                     offset, B, D, S, out_dims, out_newval, out_xtilde);
  return finish_launch("k_xtilde");
}
