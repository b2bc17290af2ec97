// rate_table.hip -- K1: q_{t|0} = V diag(exp(c_t * lambda)) W, row-normalise, clamp; R_t = beta_t R.
// Replaces forward_model.py:43-75, 95-129, 166-204, 252-306 (one launch for all nT times; the
// reference materialises N identical S x S tables per sampler step).
//
// Grid (row-blocks, nT); a 256-thread workgroup owns ROWS=16 rows x all S columns of one table
// so the row sums of the normalisation stay inside the workgroup.  K-ordered fp32 fma chain per
// output (same arithmetic as an fp32 matmul; summation order differs from MKL's).
#include "common.hpp"

namespace ctdd {

constexpr int RT_ROWS = 16;

__global__ __launch_bounds__(256) void k_rate_table(const float* __restrict__ V, const float* __restrict__ W,
                                                    const float* __restrict__ lam,
                                                    const float* __restrict__ integral, int S, int normalise,
                                                    float clamp_below, float* __restrict__ out_qt0,
                                                    float* __restrict__ out_qt0T,
                                                    float* __restrict__ out_probs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* A = smem;                      // [RT_ROWS][S]  V[r][k]*exp(c*lam[k])
  float* red = smem + RT_ROWS * S;      // [RT_ROWS][16] partial row sums
  const int t = blockIdx.y, r0 = blockIdx.x * RT_ROWS;
  const float c = integral[t];
  for (int i = threadIdx.x; i < RT_ROWS * S; i += 256) {
    const int r = i / S, k = i % S;
    A[i] = (r0 + r < S) ? V[(size_t)(r0 + r) * S + k] * expf(c * lam[k]) : 0.0f;
  }
  __syncthreads();
  const int r = threadIdx.x >> 4, cg = threadIdx.x & 15;   // 16 rows x 16 column groups
  constexpr int MAXC = CTDD_MAX_S / 16;
  float acc[MAXC];
  const int nc = (S + 15) / 16;
  float rsum = 0.0f;
  for (int j = 0; j < nc; ++j) {
    const int col = cg + 16 * j;
    float s = 0.0f;
    if (col < S)
      for (int k = 0; k < S; ++k) s = fmaf(A[r * S + k], W[(size_t)k * S + col], s);
    acc[j] = s;
    rsum += s;
  }
  red[r * 16 + cg] = rsum;
  __syncthreads();
  float tot = 0.0f;
  for (int i = 0; i < 16; ++i) tot += red[r * 16 + i];
  const int row = r0 + r;
  if (row >= S) return;
  float* q = out_qt0 ? out_qt0 + (size_t)t * S * S : nullptr;
  float* qT = out_qt0T ? out_qt0T + (size_t)t * S * S : nullptr;
  // second pass for the noising table needs the final (clamped) row: keep values in acc
  float lsum = 0.0f, lmax = -INFINITY;
  for (int j = 0; j < nc; ++j) {
    const int col = cg + 16 * j;
    if (col >= S) continue;
    float v = acc[j];
    if (normalise) v = v / tot;
    if (v < clamp_below) v = 0.0f;
    acc[j] = v;
    if (q) q[(size_t)row * S + col] = v;
    if (qT) qT[(size_t)col * S + row] = v;
    lmax = fmaxf(lmax, v <= 0.0f ? -1e9f : logf(v));
  }
  if (!out_probs) return;
  // Categorical(logits=where(row<=0,-1e9,log row)): probs = softmax(lg - logsumexp(lg))
  // (every thread that reaches here belongs to a valid row; the 16 threads of a row sit in one wave)
  for (int m = 8; m >= 1; m >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, m, WAVE));
  for (int j = 0; j < nc; ++j) {
    const int col = cg + 16 * j;
    if (col < S) lsum += expf((acc[j] <= 0.0f ? -1e9f : logf(acc[j])) - lmax);
  }
  for (int m = 8; m >= 1; m >>= 1) lsum += __shfl_xor(lsum, m, WAVE);
  const float lse = lmax + logf(lsum);
  float m2 = -INFINITY;
  for (int j = 0; j < nc; ++j) {
    const int col = cg + 16 * j;
    if (col < S) m2 = fmaxf(m2, (acc[j] <= 0.0f ? -1e9f : logf(acc[j])) - lse);
  }
  for (int m = 8; m >= 1; m >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, m, WAVE));
  float s2 = 0.0f;
  for (int j = 0; j < nc; ++j) {
    const int col = cg + 16 * j;
    if (col < S) s2 += expf((acc[j] <= 0.0f ? -1e9f : logf(acc[j])) - lse - m2);
  }
  for (int m = 8; m >= 1; m >>= 1) s2 += __shfl_xor(s2, m, WAVE);
  float* P = out_probs + (size_t)t * S * S;
  for (int j = 0; j < nc; ++j) {
    const int col = cg + 16 * j;
    if (col < S) P[(size_t)row * S + col] = expf((acc[j] <= 0.0f ? -1e9f : logf(acc[j])) - lse - m2) / s2;
  }
}


// ---- the same tables with the V diag W product on the exact-fp32 matrix instruction (S % 32 == 0: MNIST / CIFAR S = 256).
// Workgroup = 32 rows x all S columns of one table (row sums stay inside it); wave w owns the 32-column tiles i = w, w + 4, ..;
// A = V diag(exp(c lambda)) rows in LDS (row stride S + 4: the one-float fragment reads of 32 lanes hit 32 banks), the W
// columns of a tile straight from L2 (coalesced over the tile's 32 columns).  One batch (B = 64 training tables of
// 256 x 256): 327 us on fp32 FMA chains -> ~40 us.
template <int NT>
__global__ __launch_bounds__(256) void k_rate_table_mfma(const float* __restrict__ V, const float* __restrict__ W,
                                                         const float* __restrict__ lam, const float* __restrict__ integral,
                                                         int normalise, float clamp_below, float* __restrict__ out_qt0,
                                                         float* __restrict__ out_qt0T, float* __restrict__ out_probs) {
  constexpr int S = 32 * NT, LDA = S + 4, TPW = (NT + 3) / 4;       // TPW: tiles per wave
  using f32x16r = __attribute__((ext_vector_type(16))) float;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* A = smem;                          // [32][LDA]
  float* red = smem + 32 * LDA;             // [4 waves][32 rows]
  float* tr = red + 4 * 32;                 // [4 waves][32][33]: transposing buffer for the qT store
  const int t = blockIdx.y, r0 = blockIdx.x * 32, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, g = lane >> 5;
  const float c = integral[t];
  for (int i = threadIdx.x; i < 32 * S; i += 256) {
    const int r = i / S, k = i % S;
    A[r * LDA + k] = V[(size_t)(r0 + r) * S + k] * expf(c * lam[k]);
  }
  __syncthreads();
  f32x16r acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  const float* Ar = A + li * LDA + g;
#pragma unroll 4
  for (int kk = 0; kk < S / 2; ++kk) {
    const float av = Ar[2 * kk];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int i = wave + 4 * j;
      if (i < NT) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, W[(size_t)(2 * kk + g) * S + 32 * i + li], acc[j], 0, 0, 0);
    }
  }
  // a reduction over the S columns of every row: registers r of a lane are 16 rows, the 32 lanes of a half their columns
  auto row_reduce = [&](float (&v)[16], bool is_max) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(v[r], o, WAVE);
        v[r] = is_max ? fmaxf(v[r], ov) : v[r] + ov;
      }
    __syncthreads();
    if (li == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * g] = v[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
      v[r] = is_max ? fmaxf(fmaxf(red[row], red[32 + row]), fmaxf(red[64 + row], red[96 + row]))
                    : (red[row] + red[32 + row]) + (red[64 + row] + red[96 + row]);
    }
  };
  float part[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    part[r] = 0.0f;
#pragma unroll
    for (int j = 0; j < TPW; ++j)
      if (wave + 4 * j < NT) part[r] += acc[j][r];
  }
  row_reduce(part, false);                               // part[r] = the row's total
  float* q = out_qt0 ? out_qt0 + (size_t)t * S * S : nullptr;
  float* qT = out_qt0T ? out_qt0T + (size_t)t * S * S : nullptr;
  float* myT = tr + wave * 32 * 33;
  float lmax[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) lmax[r] = -INFINITY;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int i = wave + 4 * j;
    if (i >= NT) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
      float v = acc[j][r];
      if (normalise) v = v / part[r];
      if (v < clamp_below) v = 0.0f;
      acc[j][r] = v;
      if (q) q[(size_t)(r0 + row) * S + 32 * i + li] = v;
      myT[li * 33 + row] = v;                              // [column][row]
      lmax[r] = fmaxf(lmax[r], v <= 0.0f ? -1e9f : logf(v));
    }
    if (qT) {                                              // (wave-private buffer: lanes of one wave only)
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) {
        const int col = 2 * cc + g;
        qT[(size_t)(32 * i + col) * S + r0 + li] = myT[col * 33 + li];
      }
    }
  }
  if (!out_probs) return;
  // Categorical(logits = where(row <= 0, -1e9, log row)): probs = softmax(lg - logsumexp(lg)), the reductions over the whole row
  row_reduce(lmax, true);
  float lsum[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    lsum[r] = 0.0f;
#pragma unroll
    for (int j = 0; j < TPW; ++j)
      if (wave + 4 * j < NT) lsum[r] += expf((acc[j][r] <= 0.0f ? -1e9f : logf(acc[j][r])) - lmax[r]);
  }
  row_reduce(lsum, false);
  float m2[16], lse[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    lse[r] = lmax[r] + logf(lsum[r]);
    m2[r] = -INFINITY;
#pragma unroll
    for (int j = 0; j < TPW; ++j)
      if (wave + 4 * j < NT) m2[r] = fmaxf(m2[r], (acc[j][r] <= 0.0f ? -1e9f : logf(acc[j][r])) - lse[r]);
  }
  row_reduce(m2, true);
  float s2[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    s2[r] = 0.0f;
#pragma unroll
    for (int j = 0; j < TPW; ++j)
      if (wave + 4 * j < NT) s2[r] += expf((acc[j][r] <= 0.0f ? -1e9f : logf(acc[j][r])) - lse[r] - m2[r]);
  }
  row_reduce(s2, false);
  float* P = out_probs + (size_t)t * S * S;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int i = wave + 4 * j;
    if (i >= NT) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
      P[(size_t)(r0 + row) * S + 32 * i + li] = expf((acc[j][r] <= 0.0f ? -1e9f : logf(acc[j][r])) - lse[r] - m2[r]) / s2[r];
    }
  }
}

__global__ void k_scale_rate(const float* __restrict__ base, const float* __restrict__ beta, int S,
                             float* __restrict__ out) {
  const int t = blockIdx.y;
  const float b = beta[t];
  const size_t n = (size_t)S * S;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[(size_t)t * n + i] = base[i] * b;
}

}  // namespace ctdd
using namespace ctdd;

extern "C" int ctdd_rate_table(const float* eigvecs, const float* right, const float* eigvals,
                               const float* base_rate, const float* integral, const float* beta, int nT,
                               int S, int normalise, float clamp_below, float* out_qt0, float* out_qt0T,
                               float* out_rate, float* out_noise_probs, void* stream) {
  CTDD_REQUIRE(nT > 0 && nT <= 65535, CTDD_ERANGE, "nT=%d outside [1,65535]", nT);
  CTDD_REQUIRE(S >= 2 && S <= CTDD_MAX_S, CTDD_ERANGE, "S=%d outside [2,%d]", S, CTDD_MAX_S);
  hipStream_t st = (hipStream_t)stream;
  if (out_qt0 || out_qt0T || out_noise_probs) {
    CTDD_REQUIRE(eigvecs && right && eigvals && integral, CTDD_EINVAL, "null eigen-decomposition input");
    if (S % 32 == 0 && S <= 256) {                    // matrix-core variant
      const size_t lds = (size_t)(32 * (S + 4) + 4 * 32 + 4 * 32 * 33) * sizeof(float);
      dim3 g(S / 32, nT), b(256);
#define RT_MFMA(NT_)                                                                                                    \
  case NT_: {                                                                                                           \
    static bool done[16] = {};                                                                                          \
    ensure_lds_ceiling((const void*)k_rate_table_mfma<NT_>, done);                                                      \
    hipLaunchKernelGGL(k_rate_table_mfma<NT_>, g, b, lds, st, eigvecs, right, eigvals, integral, normalise, clamp_below, \
                       out_qt0, out_qt0T, out_noise_probs);                                                             \
  } break;
      switch (S / 32) { RT_MFMA(1) RT_MFMA(2) RT_MFMA(3) RT_MFMA(4) RT_MFMA(5) RT_MFMA(6) RT_MFMA(7) RT_MFMA(8) }
#undef RT_MFMA
      if (int rc = finish_launch("k_rate_table_mfma")) return rc;
    } else {
      const size_t lds = (size_t)(RT_ROWS * S + RT_ROWS * 16) * sizeof(float);
      dim3 g((S + RT_ROWS - 1) / RT_ROWS, nT), b(256);
      if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)k_rate_table, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k_rate_table, g, b, lds, st, eigvecs, right, eigvals, integral, S, normalise,
                         clamp_below, out_qt0, out_qt0T, out_noise_probs);
      if (int rc = finish_launch("k_rate_table")) return rc;
    }
  }
  if (out_rate) {
    CTDD_REQUIRE(base_rate && beta, CTDD_EINVAL, "null base_rate/beta");
    const int nb = (int)(((size_t)S * S + 255) / 256);
    dim3 g(nb > 1024 ? 1024 : nb, nT), b(256);
    hipLaunchKernelGGL(k_scale_rate, g, b, 0, st, base_rate, beta, S, out_rate);
    if (int rc = finish_launch("k_scale_rate")) return rc;
  }
  return CTDD_OK;
}
