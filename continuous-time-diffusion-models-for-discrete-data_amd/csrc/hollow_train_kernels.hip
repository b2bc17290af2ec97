// hollow_train_kernels.hip -- training-side kernels of the SDDM hollow transformer (reference: what `l.backward()`,
// lib/training/training.py:27, runs through lib/networks/hollow_networks.py:311-447 (attention / MLP blocks), 204-308 (readout),
// 90-132 (FiLM residual readout), 668-755 (embedding)), plus the training-mode forward pieces the inference kernels of
// hollow_kernels.hip do not have: dropout inside attention and after activations.
//
//   LayerNorm (+ add, + FiLM) backward      k_hollow_ln_bwd       one wave per run of rows of one sample; column sums in registers
//   attention forward with dropout          k_hollow_attn_train   thread = query (fp32 FMA, online softmax), writes (max, sum) per query
//   attention backward                      k_hollow_attn_bwd_q   thread = query: D_i = dO.O, dQ;   k_hollow_attn_bwd_kv  thread = key:
//                                           dK, dV over the queries that see it -- no atomics, the scores are recomputed in both
//   ReLU / GELU (+ dropout) forward, backward   k_hollow_act
//   embedding backward                      k_hollow_embed_bwd
// The linear layers' gradients run on the U-Net kernels (ctdd_unet_conv* with transposed weights, ctdd_unet_wgrad kind 1x1).
// Dropout masks are Philox(seed, step * 4096 + layer, element) (common.hpp), regenerated in backward.
#include "common.hpp"

namespace ctdd {

__device__ inline float hwave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
__device__ inline unsigned short ht_bf16(float a) {
  using v2f = __attribute__((ext_vector_type(2))) float;
  using v2b = __attribute__((ext_vector_type(2))) __bf16;
  v2f v = {a, 0.0f};
  return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(v, v2b)) & 0xFFFFu);
}
// keep flags of elements 4e .. 4e+3 of a tensor under dropout rate p
__device__ inline unsigned keep4(const uint64_t* rng, uint64_t layer, uint64_t quad, float p) {
  const u4 r = philox_row(rng[0], rng[1] * 4096u + layer, quad, 0x44524F50u);
  return (u01(r.x) >= p ? 1u : 0u) | (u01(r.y) >= p ? 2u : 0u) | (u01(r.z) >= p ? 4u : 0u) | (u01(r.w) >= p ? 8u : 0u);
}

// ============================================================================ LayerNorm (+ add, + FiLM) backward
// forward (hollow_kernels.hip: k_hollow_layernorm): h = x (+ y); xhat = (h - mean) rstd; z = gamma xhat + beta; out = a_b z + b_b (FiLM)
// backward: dz = dout a_b; dxhat = dz gamma; dh = rstd (dxhat - mean_E(dxhat) - xhat mean_E(dxhat xhat)) -> dx (and dy)
//           dgamma[e] += sum_rows dz xhat; dbeta[e] += sum_rows dz; da[b][e] += sum_rows dout z; db[b][e] += sum_rows dout
struct LnBwdArgs {
  const float* x; const float* y; int64_t x_bs, y_bs;       // forward inputs (batch strides in floats)
  const float* gamma; const float* beta; float eps;
  const float* film; int film_stride;                       // optional (B, 2E)
  const float* dout; int64_t dout_bs;                       // gradient of the output, rows (b, j) at dout + b*dout_bs + j*E
  int B, T, E, rpw;                                         // rpw: rows per wave
  float* dx; int64_t dx_bs; int acc_dx;                     // gradient w.r.t. x (acc: add); dy gets the same values when y is given
  float* dy; int64_t dy_bs; int acc_dy;
  float* dgamma; float* dbeta;                              // [E], atomically accumulated
  float* dfilm;                                             // [B][2E] (da | db), atomically accumulated, or null
};
__global__ __launch_bounds__(256) void k_hollow_ln_bwd(const LnBwdArgs a) {
  const int lane = threadIdx.x & 63, E = a.E;
  const int wps = (a.T + a.rpw - 1) / a.rpw;                                  // waves per sample
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (int64_t)a.B * wps) return;
  const int b = (int)(w / wps), j0 = (int)(w % wps) * a.rpw, j1 = min(j0 + a.rpw, a.T);
  float g[8], be[8], fa[8], fb[8], sg[8], sb[8], sa[8], sfb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int e = lane + 64 * k;
    g[k] = e < E ? a.gamma[e] : 0.0f;
    be[k] = e < E ? a.beta[e] : 0.0f;
    fa[k] = (a.film && e < E) ? a.film[(size_t)b * a.film_stride + e] : 1.0f;
    fb[k] = (a.film && e < E) ? a.film[(size_t)b * a.film_stride + E + e] : 0.0f;
    sg[k] = sb[k] = sa[k] = sfb[k] = 0.0f;
  }
  (void)fb;
  for (int j = j0; j < j1; ++j) {
    const float* x = a.x + (size_t)b * a.x_bs + (size_t)j * E;
    const float* y = a.y ? a.y + (size_t)b * a.y_bs + (size_t)j * E : nullptr;
    const float* dr = a.dout + (size_t)b * a.dout_bs + (size_t)j * E;
    float h[8], d[8];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int e = lane + 64 * k;
      h[k] = e < E ? x[e] + (y ? y[e] : 0.0f) : 0.0f;
      d[k] = e < E ? dr[e] : 0.0f;
      s += h[k];
    }
    const float mean = hwave_sum(s) / (float)E;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float t = lane + 64 * k < E ? h[k] - mean : 0.0f;
      q = fmaf(t, t, q);
    }
    const float rstd = 1.0f / sqrtf(hwave_sum(q) / (float)E + a.eps);
    float m1 = 0.0f, m2 = 0.0f, xh[8], dxh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool on = lane + 64 * k < E;
      xh[k] = on ? (h[k] - mean) * rstd : 0.0f;
      const float z = g[k] * xh[k] + be[k];
      const float dz = d[k] * fa[k];
      sa[k] += d[k] * z; sfb[k] += d[k];
      sg[k] = fmaf(dz, xh[k], sg[k]); sb[k] += dz;
      dxh[k] = on ? dz * g[k] : 0.0f;
      m1 += dxh[k]; m2 = fmaf(dxh[k], xh[k], m2);
    }
    m1 = hwave_sum(m1) / (float)E; m2 = hwave_sum(m2) / (float)E;
    float* dx = a.dx + (size_t)b * a.dx_bs + (size_t)j * E;
    float* dy = (a.dy && a.y) ? a.dy + (size_t)b * a.dy_bs + (size_t)j * E : nullptr;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int e = lane + 64 * k;
      if (e < E) {
        const float v = rstd * (dxh[k] - m1 - xh[k] * m2);
        dx[e] = a.acc_dx ? dx[e] + v : v;
        if (dy) dy[e] = a.acc_dy ? dy[e] + v : v;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int e = lane + 64 * k;
    if (e < E) {
      atomicAdd(a.dgamma + e, sg[k]);
      atomicAdd(a.dbeta + e, sb[k]);
      if (a.dfilm) {
        atomicAdd(a.dfilm + (size_t)b * 2 * E + e, sa[k]);
        atomicAdd(a.dfilm + (size_t)b * 2 * E + E + e, sfb[k]);
      }
    }
  }
}

// ============================================================================ attention, training mode
// rows as in hollow_kernels.hip: q (b, i) at q + b*q_bs + i*q_rs + h*hd etc.; mode 0 causal, 1 anti-causal, 2 readout.
// dropout (nn.MultiheadAttention's attention dropout): out_i = sum_j softmax(s)_ij keep_ij / (1 - p) v_j.
struct AttnTrainArgs {
  const float* q; const float* k; const float* v; int64_t q_bs, k_bs, v_bs; int q_rs, k_rs, v_rs;
  int B, Tq, Tk, H, hd, mode; float scale;
  float* out; int out_rs;                         // (b, i) at out + (b*Tq + i)*out_rs + h*hd
  float* stats;                                   // [B][H][Tq][4]: row max, row sum (of exp(s - max)), D_i = dO.O (backward), spare
  float drop_p; const uint64_t* rng; uint64_t layer;
  // backward
  const float* d_out; float* dq; float* dk; float* dv; int64_t dq_bs, dk_bs, dv_bs; int dq_rs, dk_rs, dv_rs;
};
__device__ inline bool attn_allowed(int mode, int Tq, int i, int j) {
  if (mode == 0) return j <= i;
  if (mode == 1) return j >= i;
  return j == 0 || (j <= Tq ? j - 1 <= i : j - Tq - 1 >= i);
}
// keep flag of probability (b, h, i, j): one Philox block per four consecutive keys
__device__ inline bool attn_keep(const AttnTrainArgs& a, int b, int h, int i, int j) {
  const uint64_t quads = (uint64_t)(a.Tk + 3) / 4;
  const uint64_t quad = (((uint64_t)b * a.H + h) * a.Tq + i) * quads + (uint64_t)(j >> 2);
  return (keep4(a.rng, a.layer, quad, a.drop_p) >> (j & 3)) & 1u;
}
constexpr int TQ = 128, TK = 32;
template <int HD>
__global__ __launch_bounds__(TQ) void k_hollow_attn_train(const AttnTrainArgs a) {
  __shared__ __attribute__((aligned(16))) float Ks[TK * HD];
  __shared__ __attribute__((aligned(16))) float Vs[TK * HD];
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * TQ, i = i0 + threadIdx.x;
  const bool qok = i < a.Tq;
  float qv[HD], acc[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) { acc[c] = 0.0f; qv[c] = 0.0f; }
  if (qok) {
    const float* qr = a.q + (size_t)b * a.q_bs + (size_t)i * a.q_rs + h * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) qv[c] = qr[c] * a.scale;
  }
  float m = -INFINITY, l = 0.0f;
  const float inv_keep = a.drop_p > 0.0f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  for (int j0 = 0; j0 < a.Tk; j0 += TK) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < TK * HD; idx += TQ) {
      const int jj = idx / HD, c = idx % HD, j = j0 + jj;
      Ks[idx] = j < a.Tk ? a.k[(size_t)b * a.k_bs + (size_t)j * a.k_rs + h * HD + c] : 0.0f;
      Vs[idx] = j < a.Tk ? a.v[(size_t)b * a.v_bs + (size_t)j * a.v_rs + h * HD + c] : 0.0f;
    }
    __syncthreads();
    if (!qok) continue;
    const int nj = min(TK, a.Tk - j0);
    for (int jj = 0; jj < nj; ++jj) {
      const int j = j0 + jj;
      if (!attn_allowed(a.mode, a.Tq, i, j)) continue;
      float s = 0.0f;
#pragma unroll
      for (int c = 0; c < HD; ++c) s = fmaf(qv[c], Ks[jj * HD + c], s);
      const float mn = fmaxf(m, s), corr = expf(m - mn), p = expf(s - mn);
      l = l * corr + p;
      float pd = p;
      if (a.drop_p > 0.0f) pd = attn_keep(a, b, h, i, j) ? p * inv_keep : 0.0f;
#pragma unroll
      for (int c = 0; c < HD; ++c) acc[c] = fmaf(pd, Vs[jj * HD + c], acc[c] * corr);
      m = mn;
    }
  }
  if (qok) {
    const float inv = 1.0f / l;
    float* o = a.out + ((size_t)b * a.Tq + i) * a.out_rs + h * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = acc[c] * inv;
    float* st = a.stats + (((size_t)b * a.H + h) * a.Tq + i) * 4;
    st[0] = m; st[1] = l;
  }
}
// thread = query: D_i = dO_i . O_i, dQ_i = scale sum_j dS_ij k_j,  dS_ij = p_ij (keep_ij / (1-p) dO_i.v_j - D_i)
template <int HD>
__global__ __launch_bounds__(TQ) void k_hollow_attn_bwd_q(const AttnTrainArgs a) {
  __shared__ __attribute__((aligned(16))) float Ks[TK * HD];
  __shared__ __attribute__((aligned(16))) float Vs[TK * HD];
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * TQ, i = i0 + threadIdx.x;
  const bool qok = i < a.Tq;
  float qv[HD], dO[HD], dq[HD];
  float m = 0.0f, il = 0.0f, Di = 0.0f;
#pragma unroll
  for (int c = 0; c < HD; ++c) { qv[c] = 0.0f; dO[c] = 0.0f; dq[c] = 0.0f; }
  if (qok) {
    const float* qr = a.q + (size_t)b * a.q_bs + (size_t)i * a.q_rs + h * HD;
    const float* dr = a.d_out + ((size_t)b * a.Tq + i) * a.out_rs + h * HD;
    const float* orow = a.out + ((size_t)b * a.Tq + i) * a.out_rs + h * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) { qv[c] = qr[c] * a.scale; dO[c] = dr[c]; Di = fmaf(dr[c], orow[c], Di); }
    float* st = a.stats + (((size_t)b * a.H + h) * a.Tq + i) * 4;
    m = st[0]; il = 1.0f / st[1];
    st[2] = Di;
  }
  const float inv_keep = a.drop_p > 0.0f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  for (int j0 = 0; j0 < a.Tk; j0 += TK) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < TK * HD; idx += TQ) {
      const int jj = idx / HD, c = idx % HD, j = j0 + jj;
      Ks[idx] = j < a.Tk ? a.k[(size_t)b * a.k_bs + (size_t)j * a.k_rs + h * HD + c] : 0.0f;
      Vs[idx] = j < a.Tk ? a.v[(size_t)b * a.v_bs + (size_t)j * a.v_rs + h * HD + c] : 0.0f;
    }
    __syncthreads();
    if (!qok) continue;
    const int nj = min(TK, a.Tk - j0);
    for (int jj = 0; jj < nj; ++jj) {
      const int j = j0 + jj;
      if (!attn_allowed(a.mode, a.Tq, i, j)) continue;
      float s = 0.0f, dp = 0.0f;
#pragma unroll
      for (int c = 0; c < HD; ++c) { s = fmaf(qv[c], Ks[jj * HD + c], s); dp = fmaf(dO[c], Vs[jj * HD + c], dp); }
      const float p = expf(s - m) * il;
      if (a.drop_p > 0.0f) dp = attn_keep(a, b, h, i, j) ? dp * inv_keep : 0.0f;
      const float ds = p * (dp - Di) * a.scale;
#pragma unroll
      for (int c = 0; c < HD; ++c) dq[c] = fmaf(ds, Ks[jj * HD + c], dq[c]);
    }
  }
  if (qok) {
    float* o = a.dq + (size_t)b * a.dq_bs + (size_t)i * a.dq_rs + h * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = dq[c];
  }
}
// thread = key: dV_j = sum_i p_ij keep/(1-p) dO_i ; dK_j = scale sum_i dS_ij q_i  over the queries that see key j
template <int HD>
__global__ __launch_bounds__(TQ) void k_hollow_attn_bwd_kv(const AttnTrainArgs a) {
  __shared__ __attribute__((aligned(16))) float Qs[TK * HD];
  __shared__ __attribute__((aligned(16))) float Ds[TK * HD];
  __shared__ float St[TK * 3];
  const int b = blockIdx.z, h = blockIdx.y, j0 = blockIdx.x * TQ, j = j0 + threadIdx.x;
  const bool kok = j < a.Tk;
  float kv[HD], vv[HD], dk[HD], dv[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) { kv[c] = 0.0f; vv[c] = 0.0f; dk[c] = 0.0f; dv[c] = 0.0f; }
  if (kok) {
    const float* kr = a.k + (size_t)b * a.k_bs + (size_t)j * a.k_rs + h * HD;
    const float* vr = a.v + (size_t)b * a.v_bs + (size_t)j * a.v_rs + h * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) { kv[c] = kr[c]; vv[c] = vr[c]; }
  }
  const float inv_keep = a.drop_p > 0.0f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  for (int i0 = 0; i0 < a.Tq; i0 += TK) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < TK * HD; idx += TQ) {
      const int ii = idx / HD, c = idx % HD, i = i0 + ii;
      Qs[idx] = i < a.Tq ? a.q[(size_t)b * a.q_bs + (size_t)i * a.q_rs + h * HD + c] * a.scale : 0.0f;
      Ds[idx] = i < a.Tq ? a.d_out[((size_t)b * a.Tq + i) * a.out_rs + h * HD + c] : 0.0f;
    }
    for (int idx = threadIdx.x; idx < TK; idx += TQ) {
      const int i = i0 + idx;
      const float* st = a.stats + (((size_t)b * a.H + h) * a.Tq + min(i, a.Tq - 1)) * 4;
      St[idx * 3] = st[0]; St[idx * 3 + 1] = 1.0f / st[1]; St[idx * 3 + 2] = st[2];
    }
    __syncthreads();
    if (!kok) continue;
    const int ni = min(TK, a.Tq - i0);
    for (int ii = 0; ii < ni; ++ii) {
      const int i = i0 + ii;
      if (!attn_allowed(a.mode, a.Tq, i, j)) continue;
      float s = 0.0f, dp = 0.0f;
#pragma unroll
      for (int c = 0; c < HD; ++c) { s = fmaf(Qs[ii * HD + c], kv[c], s); dp = fmaf(Ds[ii * HD + c], vv[c], dp); }
      const float p = expf(s - St[ii * 3]) * St[ii * 3 + 1];
      float pk = p;
      if (a.drop_p > 0.0f) { const bool keep = attn_keep(a, b, h, i, j); pk = keep ? p * inv_keep : 0.0f; dp = keep ? dp * inv_keep : 0.0f; }
      const float ds = p * (dp - St[ii * 3 + 2]);
#pragma unroll
      for (int c = 0; c < HD; ++c) { dv[c] = fmaf(pk, Ds[ii * HD + c], dv[c]); dk[c] = fmaf(ds, Qs[ii * HD + c], dk[c]); }
    }
  }
  if (kok) {
    float* ok_ = a.dk + (size_t)b * a.dk_bs + (size_t)j * a.dk_rs + h * HD;
    float* ov = a.dv + (size_t)b * a.dv_bs + (size_t)j * a.dv_rs + h * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) { ok_[c] = dk[c]; ov[c] = dv[c]; }          // (Qs holds scale q: dK = sum dS (scale q))
  }
}

// ============================================================================ activation (+ dropout), forward and backward
// forward: out = drop(act(pre)); backward: dpre = drop(dout) act'(pre).  act 0 none, 1 ReLU, 2 GELU (erf).  n % 4 == 0.
__global__ __launch_bounds__(256) void k_hollow_act(const float* __restrict__ pre, const float* __restrict__ dout, float* __restrict__ out,
                                                   unsigned short* __restrict__ out_bf16, int64_t nquad, int act, float drop_p,
                                                   const uint64_t* rng, uint64_t layer) {
  const float inv_keep = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nquad; v += (int64_t)gridDim.x * 256) {
    const float4 p4 = *(const float4*)(pre + v * 4);
    const float p[4] = {p4.x, p4.y, p4.z, p4.w};
    float g[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (dout) { const float4 d4 = *(const float4*)(dout + v * 4); g[0] = d4.x; g[1] = d4.y; g[2] = d4.z; g[3] = d4.w; }
    const unsigned keep = drop_p > 0.0f ? keep4(rng, layer, (uint64_t)v, drop_p) : 0xFu;
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float f, df;                                             // act(pre), act'(pre)
      if (act == 1) { f = fmaxf(p[k], 0.0f); df = p[k] > 0.0f ? 1.0f : 0.0f; }
      else if (act == 2) {
        const float cdf = 0.5f * (1.0f + erff(p[k] * 0.70710678118654752f));
        f = p[k] * cdf; df = cdf + p[k] * 0.3989422804014327f * expf(-0.5f * p[k] * p[k]);
      } else { f = p[k]; df = 1.0f; }
      const float kf = (keep >> k) & 1u ? inv_keep : 0.0f;
      r[k] = dout ? g[k] * kf * df : f * kf;
    }
    *(float4*)(out + v * 4) = make_float4(r[0], r[1], r[2], r[3]);
    if (out_bf16) {
      unsigned short* o = out_bf16 + v * 4;
      o[0] = ht_bf16(r[0]); o[1] = ht_bf16(r[1]); o[2] = ht_bf16(r[2]); o[3] = ht_bf16(r[3]);
    }
  }
}

// ============================================================================ embedding backward (hollow_networks.py:729-753)
// x_embed[b][d] = w_in xn[b][d] + b_in feeds l2r[b][d+1] and r2l[b][d-1]:  dxe[b][d] = dl2r[b][d+1] (d <= D-2) + dr2l[b][d-1] (d >= 1)
// dw_in[e] = sum dxe xn ; db_in[e] = sum dxe
struct EmbedBwdArgs { const int64_t* x64; const int32_t* x32; const float* dl2r; const float* dr2l; int B, D, E, S; float* dw; float* db; };
__global__ __launch_bounds__(256) void k_hollow_embed_bwd(const EmbedBwdArgs a) {
  const int b = blockIdx.y, E = a.E, D = a.D;
  for (int e = threadIdx.x; e < E; e += 256) {
    float sw = 0.0f, sb = 0.0f;
    for (int d = blockIdx.x; d < D; d += gridDim.x) {
      const float xr = a.x64 ? (float)a.x64[(size_t)b * D + d] : (float)a.x32[(size_t)b * D + d];
      const float xn = (xr / (float)(a.S - 1)) * 2.0f - 1.0f;
      float g = 0.0f;
      if (d <= D - 2) g += a.dl2r[((size_t)b * D + d + 1) * E + e];
      if (d >= 1) g += a.dr2l[((size_t)b * D + d - 1) * E + e];
      sw = fmaf(g, xn, sw); sb += g;
    }
    atomicAdd(a.dw + e, sw);
    atomicAdd(a.db + e, sb);
  }
}

}  // namespace ctdd
using namespace ctdd;

extern "C" int ctdd_hollow_layernorm_bwd(const void* args_, void* stream) {
  LnBwdArgs a = *(const LnBwdArgs*)args_;
  CTDD_REQUIRE(a.x && a.gamma && a.beta && a.dout && a.dx && a.dgamma && a.dbeta, CTDD_EINVAL, "layernorm bwd: null buffer");
  CTDD_REQUIRE(a.E >= 1 && a.E <= 512 && a.B > 0 && a.T > 0, CTDD_ERANGE, "layernorm bwd: E=%d (<= 512) B=%d T=%d", a.E, a.B, a.T);
  if (a.rpw <= 0) a.rpw = 8;
  const int64_t waves = (int64_t)a.B * ((a.T + a.rpw - 1) / a.rpw);
  hipLaunchKernelGGL(k_hollow_ln_bwd, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return finish_launch("k_hollow_ln_bwd");
}

#define HD_DISPATCH(KERNEL, GRIDX)                                                                          \
  switch (a.hd) {                                                                                            \
    case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(GRIDX, a.H, a.B), dim3(TQ), 0, (hipStream_t)stream, a); break;   \
    case 8: hipLaunchKernelGGL(KERNEL<8>, dim3(GRIDX, a.H, a.B), dim3(TQ), 0, (hipStream_t)stream, a); break;   \
    case 16: hipLaunchKernelGGL(KERNEL<16>, dim3(GRIDX, a.H, a.B), dim3(TQ), 0, (hipStream_t)stream, a); break; \
    case 32: hipLaunchKernelGGL(KERNEL<32>, dim3(GRIDX, a.H, a.B), dim3(TQ), 0, (hipStream_t)stream, a); break; \
    default: CTDD_REQUIRE(false, CTDD_ERANGE, "attention (training): head dimension %d (4, 8, 16 or 32)", a.hd);   \
  }
static int attn_check(const AttnTrainArgs& a) {
  CTDD_REQUIRE(a.q && a.k && a.v && a.out && a.stats, CTDD_EINVAL, "attention (training): null buffer");
  CTDD_REQUIRE(a.mode >= 0 && a.mode <= 2 && (a.mode != 2 || a.Tk == 2 * a.Tq + 1), CTDD_EINVAL, "attention (training): mode %d Tq=%d Tk=%d", a.mode, a.Tq, a.Tk);
  CTDD_REQUIRE(a.drop_p >= 0.0f && a.drop_p < 1.0f && (a.drop_p == 0.0f || a.rng), CTDD_EINVAL, "attention (training): dropout %g", (double)a.drop_p);
  return CTDD_OK;
}
extern "C" int ctdd_hollow_attention_train(const void* args_, void* stream) {
  const AttnTrainArgs& a = *(const AttnTrainArgs*)args_;
  if (int rc = attn_check(a)) return rc;
  HD_DISPATCH(k_hollow_attn_train, (a.Tq + TQ - 1) / TQ)
  return finish_launch("k_hollow_attn_train");
}
extern "C" int ctdd_hollow_attention_bwd(const void* args_, void* stream) {
  const AttnTrainArgs& a = *(const AttnTrainArgs*)args_;
  if (int rc = attn_check(a)) return rc;
  CTDD_REQUIRE(a.d_out && a.dq && a.dk && a.dv, CTDD_EINVAL, "attention bwd: null gradient buffer");
  HD_DISPATCH(k_hollow_attn_bwd_q, (a.Tq + TQ - 1) / TQ)
  if (int rc = finish_launch("k_hollow_attn_bwd_q")) return rc;
  HD_DISPATCH(k_hollow_attn_bwd_kv, (a.Tk + TQ - 1) / TQ)
  return finish_launch("k_hollow_attn_bwd_kv");
}
#undef HD_DISPATCH

extern "C" int ctdd_hollow_act(const float* pre, const float* dout, float* out, void* out_bf16, int64_t n, int act, float drop_p,
                               const uint64_t* rng, uint64_t layer, void* stream) {
  CTDD_REQUIRE(pre && out && n > 0 && n % 4 == 0 && act >= 0 && act <= 2, CTDD_EINVAL, "act: n=%lld act=%d", (long long)n, act);
  CTDD_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f && (drop_p == 0.0f || rng), CTDD_EINVAL, "act: dropout %g", (double)drop_p);
  int64_t g = (n / 4 + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_hollow_act, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, pre, dout, out, (unsigned short*)out_bf16, n / 4, act,
                     drop_p, rng, layer);
  return finish_launch("k_hollow_act");
}

extern "C" int ctdd_hollow_embed_bwd(const void* args_, void* stream) {
  const EmbedBwdArgs& a = *(const EmbedBwdArgs*)args_;
  CTDD_REQUIRE((a.x64 || a.x32) && a.dl2r && a.dr2l && a.dw && a.db, CTDD_EINVAL, "embed bwd: null buffer");
  hipLaunchKernelGGL(k_hollow_embed_bwd, dim3(16, a.B), dim3(256), 0, (hipStream_t)stream, a);
  return finish_launch("k_hollow_embed_bwd");
}
