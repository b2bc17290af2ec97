// hollow_train_kernels.hip -- training-side kernels of the SDDM hollow transformer (reference: what `l.backward()`,
// lib/training/training.py:27, runs through lib/networks/hollow_networks.py:311-447 (attention / MLP blocks), 204-308 (readout),
// 90-132 (FiLM residual readout), 668-755 (embedding)), plus the training-mode forward pieces the inference kernels of
// hollow_kernels.hip do not have: dropout inside attention and after activations.
//
//   LayerNorm (+ add, + FiLM) backward      k_hollow_ln_bwd<KN>   four rows in flight per wave; dgamma / dbeta through LDS into replicated accumulators
//   attention, fp32 FMA (parity mode)       k_hollow_attn_train (thread = query, online softmax, writes (max, sum) per query),
//                                           k_hollow_attn_bwd_q (thread = query: D_i = dO.O, dQ), k_hollow_attn_bwd_kv (thread = key: dK, dV)
//   attention, matrix cores (bf16 mode)     k_hollow_attn_q_mfma<HD, BWD, NW> (forward / dQ), k_hollow_attn_kv_mfma<HD, NW> (dK, dV)
//                                           -- no atomics, the scores are recomputed in both backward kernels; the same dropout masks
//   ReLU / GELU (+ dropout)                 k_hollow_act (fp32), k_hollow_relu_bf16 (the bf16-only MLP hidden tensor)
//   dropout (+ residual, + bf16 copy)       k_hollow_dropout
//   bias gradients                          k_hollow_colsum (two-stage column sums)
//   embedding backward                      k_hollow_embed_bwd
// The linear layers run on gemm_kernels.hip (bf16) / the U-Net's fp32 GEMM kernel, their weight gradients on ctdd_unet_wgrad (kind 1x1).
// Dropout masks are Philox(seed, step * 4096 + layer, element) (common.hpp), regenerated in backward.
#include "common.hpp"
#ifndef CTDD_ATT_NW
#define CTDD_ATT_NW 4      // waves per workgroup of the matrix-core attention kernels
#endif

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
__device__ inline unsigned keep4v(uint64_t seed, uint64_t ctr, uint64_t quad, float p) {      // (seed, step * 4096 + layer) preloaded
  const u4 r = philox_row(seed, ctr, quad, 0x44524F50u);
  return (u01(r.x) >= p ? 1u : 0u) | (u01(r.y) >= p ? 2u : 0u) | (u01(r.z) >= p ? 4u : 0u) | (u01(r.w) >= p ? 8u : 0u);
}
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
  int nrep; int rep_stride;                                 // > 1: workgroup w adds into dgamma / dbeta + (w % nrep) * rep_stride (the caller sums the replicas)
  const float* dres; int64_t dres_bs;                       // optional: dx = dres + gradient (the residual stream's incoming gradient; out of place)
};
// KN = ceil(E / 64) columns per lane; RG rows in flight per wave (independent loads and reduction chains overlap); the
// four waves of a workgroup meet in LDS before ONE atomic per column and workgroup (thousands of waves on the same
// 2E addresses serialise in L2 otherwise: 117 us -> ~15 us at 28800 x 128).
template <int KN>
__global__ __launch_bounds__(256) void k_hollow_ln_bwd(const LnBwdArgs a) {
  constexpr int RG = KN <= 2 ? 4 : 2;
  __shared__ float red[4][2][64 * KN];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, E = a.E;
  const int wps = (a.T + a.rpw - 1) / a.rpw;                                  // waves per sample
  const int64_t w = (int64_t)blockIdx.x * 4 + wv;
  const bool live = w < (int64_t)a.B * wps;
  const int b = live ? (int)(w / wps) : 0, j0 = live ? (int)(w % wps) * a.rpw : 0, j1 = live ? min(j0 + a.rpw, a.T) : 0;
  float g[KN], be[KN], fa[KN], sg[KN], sb[KN], sa[KN], sfb[KN];
#pragma unroll
  for (int k = 0; k < KN; ++k) {
    const int e = lane + 64 * k;
    g[k] = e < E ? a.gamma[e] : 0.0f;
    be[k] = e < E ? a.beta[e] : 0.0f;
    fa[k] = (a.film && e < E) ? a.film[(size_t)b * a.film_stride + e] : 1.0f;
    sg[k] = sb[k] = sa[k] = sfb[k] = 0.0f;
  }
  for (int jb = j0; jb < j1; jb += RG) {
    float h[RG][KN], d[RG][KN], s[RG], q[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      const int j = min(jb + r, j1 - 1);
      const float* x = a.x + (size_t)b * a.x_bs + (size_t)j * E;
      const float* y = a.y ? a.y + (size_t)b * a.y_bs + (size_t)j * E : nullptr;
      const float* dr = a.dout + (size_t)b * a.dout_bs + (size_t)j * E;
      s[r] = 0.0f;
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        const int e = lane + 64 * k;
        h[r][k] = e < E ? x[e] + (y ? y[e] : 0.0f) : 0.0f;
        d[r][k] = (e < E && jb + r < j1) ? dr[e] : 0.0f;                       // rows past the run contribute nothing
        s[r] += h[r][k];
      }
    }
#pragma unroll
    for (int r = 0; r < RG; ++r) s[r] = hwave_sum(s[r]) / (float)E;
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      q[r] = 0.0f;
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        const float t = lane + 64 * k < E ? h[r][k] - s[r] : 0.0f;
        q[r] = fmaf(t, t, q[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < RG; ++r) q[r] = 1.0f / sqrtf(hwave_sum(q[r]) / (float)E + a.eps);
    float m1[RG], m2[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      m1[r] = m2[r] = 0.0f;
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        const bool on = lane + 64 * k < E;
        const float xh = on ? (h[r][k] - s[r]) * q[r] : 0.0f;
        const float z = g[k] * xh + be[k];
        const float dz = d[r][k] * fa[k];
        sa[k] = fmaf(d[r][k], z, sa[k]); sfb[k] += d[r][k];
        sg[k] = fmaf(dz, xh, sg[k]); sb[k] += dz;
        const float dxh = on ? dz * g[k] : 0.0f;
        h[r][k] = xh; d[r][k] = dxh;                                          // reuse the registers: xhat, dxhat
        m1[r] += dxh; m2[r] = fmaf(dxh, xh, m2[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < RG; ++r) { m1[r] = hwave_sum(m1[r]) / (float)E; m2[r] = hwave_sum(m2[r]) / (float)E; }
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      if (jb + r >= j1) break;
      const int j = jb + r;
      float* dx = a.dx + (size_t)b * a.dx_bs + (size_t)j * E;
      float* dy = (a.dy && a.y) ? a.dy + (size_t)b * a.dy_bs + (size_t)j * E : nullptr;
      const float* dres = a.dres ? a.dres + (size_t)b * a.dres_bs + (size_t)j * E : nullptr;
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        const int e = lane + 64 * k;
        if (e < E) {
          const float v = q[r] * (d[r][k] - m1[r] - h[r][k] * m2[r]);
          dx[e] = dres ? dres[e] + v : (a.acc_dx ? dx[e] + v : v);
          if (dy) dy[e] = a.acc_dy ? dy[e] + v : v;
        }
      }
    }
  }
  if (a.dfilm && live) {                                                       // per-sample sums: few waves per address
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const int e = lane + 64 * k;
      if (e < E) {
        atomicAdd(a.dfilm + (size_t)b * 2 * E + e, sa[k]);
        atomicAdd(a.dfilm + (size_t)b * 2 * E + E + e, sfb[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KN; ++k) { red[wv][0][lane + 64 * k] = sg[k]; red[wv][1][lane + 64 * k] = sb[k]; }
  __syncthreads();
  const int rep = a.nrep > 1 ? (int)(blockIdx.x % (unsigned)a.nrep) : 0;
  for (int idx = threadIdx.x; idx < 2 * 64 * KN; idx += 256) {
    const int which = idx / (64 * KN), e = idx % (64 * KN);
    if (e < E)
      atomicAdd((which ? a.dbeta : a.dgamma) + (size_t)rep * a.rep_stride + e, red[0][which][e] + red[1][which][e] + red[2][which][e] + red[3][which][e]);
  }
}

// The same with 16-byte vectors (E = 4 NV LPR, LPR | 64: every width the trainer runs -- 128, 256, 512): a lane owns 4 NV
// columns, LPR lanes a row, a wave instruction moves 64 / LPR rows; the reductions run over LPR lanes.  (With a column per lane
// and 4-byte loads the 28 800 x 128 call took 28.6 us for 59 MB; the rows' dependent load -> mean -> variance -> means -> store
// chains need more bytes in flight per wave than that layout gives.)
template <int NV>
__global__ __launch_bounds__(256) void k_hollow_ln_bwd_v4(const LnBwdArgs a, int LPR) {
  constexpr int RG = NV == 1 ? 4 : 2;
  __shared__ float red[4][2][512];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, E = a.E;
  const int RPP = 64 / LPR, sub = lane % LPR, rsub = lane / LPR;
  const int wps = (a.T + a.rpw - 1) / a.rpw;                                  // waves per sample
  const int64_t w = (int64_t)blockIdx.x * 4 + wv;
  const bool live = w < (int64_t)a.B * wps;
  const int b = live ? (int)(w / wps) : 0, j0 = live ? (int)(w % wps) * a.rpw : 0, j1 = live ? min(j0 + a.rpw, a.T) : 0;
  // sums over the LPR lanes of a row, all RG row groups at once (independent exchanges per step: one group at a time was a chain
  // of 5 x RG dependent LDS round trips per reduction and made this kernel slower than the column-per-lane one)
  auto lsum = [&](float (&v)[RG]) {
    for (int o = LPR >> 1; o >= 1; o >>= 1) {
#pragma unroll
      for (int r = 0; r < RG; ++r) v[r] += __shfl_xor(v[r], o, WAVE);
    }
  };
  float4 g[NV], be[NV], fa[NV], sg[NV], sb[NV], sa[NV], sfb[NV];
  const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int e = 4 * sub + 4 * LPR * v;
    g[v] = *(const float4*)(a.gamma + e);
    be[v] = *(const float4*)(a.beta + e);
    fa[v] = a.film ? *(const float4*)(a.film + (size_t)b * a.film_stride + e) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    sg[v] = sb[v] = sa[v] = sfb[v] = z4;
  }
  for (int jb = j0; jb < j1; jb += RG * RPP) {
    float h[RG][NV][4], d[RG][NV][4], s_[RG], q[RG];
    bool ok[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      const int jr = jb + r * RPP + rsub;
      ok[r] = jr < j1;
      const int j = ok[r] ? jr : j1 - 1;
      const float* x = a.x + (size_t)b * a.x_bs + (size_t)j * E;
      const float* y = a.y ? a.y + (size_t)b * a.y_bs + (size_t)j * E : nullptr;
      const float* dr = a.dout + (size_t)b * a.dout_bs + (size_t)j * E;
      s_[r] = 0.0f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int e = 4 * sub + 4 * LPR * v;
        float4 t = *(const float4*)(x + e);
        if (y) { const float4 u = *(const float4*)(y + e); t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        const float4 dd = ok[r] ? *(const float4*)(dr + e) : z4;              // rows past the run contribute nothing
        h[r][v][0] = t.x; h[r][v][1] = t.y; h[r][v][2] = t.z; h[r][v][3] = t.w;
        d[r][v][0] = dd.x; d[r][v][1] = dd.y; d[r][v][2] = dd.z; d[r][v][3] = dd.w;
        s_[r] += (t.x + t.y) + (t.z + t.w);
      }
    }
    lsum(s_);
#pragma unroll
    for (int r = 0; r < RG; ++r) s_[r] = s_[r] / (float)E;
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      q[r] = 0.0f;
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float t = h[r][v][c] - s_[r]; q[r] = fmaf(t, t, q[r]); }
    }
    lsum(q);
#pragma unroll
    for (int r = 0; r < RG; ++r) q[r] = 1.0f / sqrtf(q[r] / (float)E + a.eps);
    float m1[RG], m2[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      m1[r] = m2[r] = 0.0f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const float gv[4] = {g[v].x, g[v].y, g[v].z, g[v].w}, bv[4] = {be[v].x, be[v].y, be[v].z, be[v].w}, fv[4] = {fa[v].x, fa[v].y, fa[v].z, fa[v].w};
        float* sgp = (float*)&sg[v]; float* sbp = (float*)&sb[v]; float* sap = (float*)&sa[v]; float* sfp = (float*)&sfb[v];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float xh = (h[r][v][c] - s_[r]) * q[r];
          const float z = gv[c] * xh + bv[c];
          const float dz = d[r][v][c] * fv[c];
          sap[c] = fmaf(d[r][v][c], z, sap[c]); sfp[c] += d[r][v][c];
          sgp[c] = fmaf(dz, xh, sgp[c]); sbp[c] += dz;
          const float dxh = dz * gv[c];
          h[r][v][c] = xh; d[r][v][c] = dxh;                                  // reuse the registers: xhat, dxhat
          m1[r] += dxh; m2[r] = fmaf(dxh, xh, m2[r]);
        }
      }
    }
    lsum(m1); lsum(m2);
#pragma unroll
    for (int r = 0; r < RG; ++r) { m1[r] = m1[r] / (float)E; m2[r] = m2[r] / (float)E; }
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      if (!ok[r]) continue;
      const int j = jb + r * RPP + rsub;
      float* dx = a.dx + (size_t)b * a.dx_bs + (size_t)j * E;
      float* dy = (a.dy && a.y) ? a.dy + (size_t)b * a.dy_bs + (size_t)j * E : nullptr;
      const float* dres = a.dres ? a.dres + (size_t)b * a.dres_bs + (size_t)j * E : nullptr;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int e = 4 * sub + 4 * LPR * v;
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = q[r] * (d[r][v][c] - m1[r] - h[r][v][c] * m2[r]);
        float4 ox = make_float4(o[0], o[1], o[2], o[3]);
        if (dres) { const float4 t = *(const float4*)(dres + e); ox.x += t.x; ox.y += t.y; ox.z += t.z; ox.w += t.w; }
        else if (a.acc_dx) { const float4 t = *(const float4*)(dx + e); ox.x += t.x; ox.y += t.y; ox.z += t.z; ox.w += t.w; }
        *(float4*)(dx + e) = ox;
        if (dy) {
          float4 oy = make_float4(o[0], o[1], o[2], o[3]);
          if (a.acc_dy) { const float4 t = *(const float4*)(dy + e); oy.x += t.x; oy.y += t.y; oy.z += t.z; oy.w += t.w; }
          *(float4*)(dy + e) = oy;
        }
      }
    }
  }
  // the row sub-groups of the wave hold partial sums of the same columns
  auto rsum4 = [&](float4& v) {
    for (int o = LPR; o < 64; o <<= 1) {
      v.x += __shfl_xor(v.x, o, WAVE); v.y += __shfl_xor(v.y, o, WAVE); v.z += __shfl_xor(v.z, o, WAVE); v.w += __shfl_xor(v.w, o, WAVE);
    }
  };
#pragma unroll
  for (int v = 0; v < NV; ++v) { rsum4(sg[v]); rsum4(sb[v]); }
  if (a.dfilm) {                                                               // per-sample sums: few waves per address
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      rsum4(sa[v]); rsum4(sfb[v]);
      if (live && rsub == 0) {
        const int e = 4 * sub + 4 * LPR * v;
        float* da = a.dfilm + (size_t)b * 2 * E + e;
        atomicAdd(da, sa[v].x); atomicAdd(da + 1, sa[v].y); atomicAdd(da + 2, sa[v].z); atomicAdd(da + 3, sa[v].w);
        atomicAdd(da + E, sfb[v].x); atomicAdd(da + E + 1, sfb[v].y); atomicAdd(da + E + 2, sfb[v].z); atomicAdd(da + E + 3, sfb[v].w);
      }
    }
  }
  if (rsub == 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int e = 4 * sub + 4 * LPR * v;
      *(float4*)(&red[wv][0][e]) = sg[v];
      *(float4*)(&red[wv][1][e]) = sb[v];
    }
  }
  __syncthreads();
  const int rep = a.nrep > 1 ? (int)(blockIdx.x % (unsigned)a.nrep) : 0;
  for (int idx = threadIdx.x; idx < 2 * E; idx += 256) {
    const int which = idx / E, e = idx % E;
    atomicAdd((which ? a.dbeta : a.dgamma) + (size_t)rep * a.rep_stride + e, red[0][which][e] + red[1][which][e] + red[2][which][e] + red[3][which][e]);
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
  // optional bf16 copies (the matrix-core kernels): of out (forward), of dq / dk / dv (same strides as the fp32 ones)
  unsigned short* out_bf16; unsigned short* dq_bf16; unsigned short* dk_bf16; unsigned short* dv_bf16;
};
__device__ inline bool attn_allowed(int mode, int Tq, int i, int j) {
  if (mode == 0) return j <= i;
  if (mode == 1) return j >= i;
  return j == 0 || (j <= Tq ? j - 1 <= i : j - Tq - 1 >= i);
}
// Dropout on the attention probabilities: one Philox block per (query, EIGHT consecutive keys), its 128 bits as eight 16-bit
// uniforms (Philox4x32-10 is 20 quarter-rate 32 x 32 -> 64 multiplies: at one block per four probabilities it was ~45 % of the
// matrix-core kernels' instructions; this stream runs seven rounds, common.hpp: philox_row7).  keep_k = u16_k >= thr, thr = round(p 65536): the rate in force is thr / 65536 (|error| <
// 8e-6), and 1 / (1 - thr / 65536) is the rescale every kernel uses.
__device__ inline unsigned attn_thr(float p) { return (unsigned)(p * 65536.0f + 0.5f); }
__device__ inline float attn_inv_keep(float p) { return p > 0.0f ? 1.0f / (1.0f - (float)attn_thr(p) * (1.0f / 65536.0f)) : 1.0f; }
__device__ inline unsigned keep8v(uint64_t seed, uint64_t ctr, uint64_t oct, unsigned thr) {
  const u4 r = philox_row7(seed, ctr, oct, 0x4154544Eu);
  return ((r.x & 0xFFFFu) >= thr ? 1u : 0u) | ((r.x >> 16) >= thr ? 2u : 0u) | ((r.y & 0xFFFFu) >= thr ? 4u : 0u) | ((r.y >> 16) >= thr ? 8u : 0u) |
         ((r.z & 0xFFFFu) >= thr ? 16u : 0u) | ((r.z >> 16) >= thr ? 32u : 0u) | ((r.w & 0xFFFFu) >= thr ? 64u : 0u) | ((r.w >> 16) >= thr ? 128u : 0u);
}
// keep flag of probability (b, h, i, j)  (the fp32 kernels: one block per probability -- their mode is the parity mode)
__device__ inline bool attn_keep(const AttnTrainArgs& a, int b, int h, int i, int j) {
  const uint64_t octs = (uint64_t)(a.Tk + 7) / 8;
  const uint64_t oct = (((uint64_t)b * a.H + h) * a.Tq + i) * octs + (uint64_t)(j >> 3);
  return (keep8v(a.rng[0], a.rng[1] * 4096u + a.layer, oct, attn_thr(a.drop_p)) >> (j & 7)) & 1u;
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
  const float inv_keep = attn_inv_keep(a.drop_p);
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
  const float inv_keep = attn_inv_keep(a.drop_p);
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
  const float inv_keep = attn_inv_keep(a.drop_p);
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

// ============================================================================ training attention on the matrix cores (bf16 operands)
// Same contract and the same Philox masks as the fp32 kernels above, head dimensions 16 and 32; the products run on
// v_mfma_f32_32x32x16_bf16, softmax / dropout / dS in fp32.  Register layout as k_hollow_attention_mfma (hollow_kernels.hip):
// a 32 x 32 accumulator tile puts ONE column on a lane (two lanes, halves kh = 0 / 1, per column) and rows
// (r & 3) + 8 (r >> 2) + 4 kh in its registers r = 0..15 -- which, packed to bf16, IS the B operand of a product contracting over
// those rows (with the A operand read from LDS in the same row order), so no tile ever moves between lanes:
//   forward / dQ kernel  (wave = 32 queries, lane = query):  S^T = K Q^T,  dP^T = V dO^T,  O^T += V^T P^T,  dQ^T += K^T dS^T
//   dK / dV kernel       (wave = 32 keys,    lane = key):    S = Q K^T,    dP = dO V^T,    dV^T += dO^T P,  dK^T += Q^T dS
using bf16x8t = __attribute__((ext_vector_type(8))) __bf16;
using f32x16t = __attribute__((ext_vector_type(16))) float;
using u32x4t = __attribute__((ext_vector_type(4))) unsigned;
__device__ inline unsigned ht_pack2(float a, float b) {
  using v2f = __attribute__((ext_vector_type(2))) float;
  using v2b = __attribute__((ext_vector_type(2))) __bf16;
  v2f v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, v2b));
}
// B-operand fragments of one row of a [.][HD] fp32 matrix (the lane's query or key): dims 16 s + 8 kh .. + 7 of k-step s
template <int HD>
__device__ inline void row_frags(const float* row, bool ok, float scale, int kh, bf16x8t (&f)[HD / 16]) {
#pragma unroll
  for (int s_ = 0; s_ < HD / 16; ++s_) {
    float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0;
    if (ok) { u0 = *(const float4*)(row + 16 * s_ + 8 * kh); u1 = *(const float4*)(row + 16 * s_ + 8 * kh + 4); }
    f[s_] = __builtin_bit_cast(bf16x8t, u32x4t{ht_pack2(u0.x * scale, u0.y * scale), ht_pack2(u0.z * scale, u0.w * scale),
                                               ht_pack2(u1.x * scale, u1.y * scale), ht_pack2(u1.z * scale, u1.w * scale)});
  }
}
// does the chunk of 32 keys from jc concern the queries [lo, hi]?  /  does every (query, key) pair of the two ranges pass the mask?
__device__ inline bool chunk_any(int mode, int Tq, int Tk, int jc, int lo, int hi) {
  const int jl = min(jc + 32, Tk) - 1;
  if (hi < lo || jc >= Tk) return false;
  if (mode == 0) return jc <= hi;
  if (mode == 1) return jl >= lo;
  return jc == 0 || (jc <= Tq && jc - 1 <= hi) || (jl > Tq && jl - Tq - 1 >= lo);
}
__device__ inline bool chunk_full(int mode, int Tq, int Tk, int j0, int q0) {   // queries q0..q0+31, keys j0..j0+31, all in range
  if (q0 + 31 >= Tq || j0 + 31 >= Tk) return false;
  if (mode == 0) return j0 + 31 <= q0;
  if (mode == 1) return j0 >= q0 + 31;
  return (j0 >= 1 && j0 + 31 <= Tq && j0 + 30 <= q0) || (j0 > Tq && j0 - Tq - 1 >= q0 + 31);
}
// The attention mask of the matrix-core kernels, branch-free with the (launch-uniform) mode as a template argument: a
// per-register `if` chain on the run-time mode made the compiler build every lane mask with scalar and / or / branch
// sequences -- ~1700 instructions per 32 x 32 tile, an order of magnitude above the softmax arithmetic.
template <int MODE>
__device__ inline bool allowed_t(int Tq, int i, int j) {
  if (MODE == 0) return j <= i;
  if (MODE == 1) return j >= i;
  return (j == 0) | ((j <= Tq) & (j - 1 <= i)) | ((j > Tq) & (j - Tq - 1 >= i));
}
// query-side tile: lane = query i, register r = key jb + (r & 3) + 8 (r >> 2)  (jb = chunk start + 4 kh)
template <int MODE>
__device__ inline void mask_tile_q(f32x16t& s, int Tq, int Tk, int i, bool qok, int jb) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = jb + (r & 3) + 8 * (r >> 2);
    const bool ok = qok & (j < Tk) & allowed_t<MODE>(Tq, i, j);
    s[r] = ok ? s[r] : -INFINITY;
  }
}
// key-side tile: lane = key j, register r = query ib + (r & 3) + 8 (r >> 2); returns the 16 flags
template <int MODE>
__device__ inline unsigned mask_bits_kv(int Tq, int j, bool kok, int ib) {
  unsigned m = 0u;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = ib + (r & 3) + 8 * (r >> 2);
    const bool ok = kok & (i < Tq) & allowed_t<MODE>(Tq, i, j);
    m |= ok ? (1u << r) : 0u;
  }
  return m;
}
constexpr int RLD16 = 8, TLD = 36;                  // row-major rows: HD + 8 bf16; transposed rows: 32 + 4 -- 72-byte rows put the 8-byte
                                                    // fragment reads of 32 lanes (one row each) on 32 distinct bank pairs; with 80-byte
                                                    // rows lanes 16 apart shared theirs (half of these kernels' LDS cycles were conflicts)

// Staging of a chunk of 32 rows of TWO fp32 matrices [row][HD] as bf16, split in a fetch (global -> registers, issued a chunk
// ahead so that the loads fly under the previous chunk's products) and a store (registers -> LDS: row-major R and / or
// transposed T [dim][row]).  256 threads: item = (matrix, row, four dims); HD / 16 items per thread.
template <int HD, int NTH>
struct Stage2 {
  static constexpr int NU = (2 * 32 * HD / 4 + NTH - 1) / NTH, ITEMS = 32 * HD / 4;
  float4 v[NU];
  __device__ inline void fetch(const float* m0, int64_t rs0, const float* m1, int64_t rs1, int r0, int rmax) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int idx = threadIdx.x + NTH * u, mat = idx / ITEMS, r = idx % ITEMS;
      const int rr = r / (HD / 4), c4 = (r % (HD / 4)) * 4;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < 2 * ITEMS && r0 + rr < rmax) v[u] = *(const float4*)((mat ? m1 : m0) + (size_t)(r0 + rr) * (mat ? rs1 : rs0) + c4);
    }
  }
  __device__ inline void store(float scale0, unsigned short* R0, unsigned short* T0, unsigned short* R1, unsigned short* T1) const {
    constexpr int RLD = HD + RLD16;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int idx = threadIdx.x + NTH * u, mat = idx / ITEMS, r = idx % ITEMS;
      if (idx >= 2 * ITEMS) continue;
      const int rr = r / (HD / 4), c4 = (r % (HD / 4)) * 4;
      const float sc = mat ? 1.0f : scale0;
      const unsigned p01 = ht_pack2(v[u].x * sc, v[u].y * sc), p23 = ht_pack2(v[u].z * sc, v[u].w * sc);
      unsigned short* R = mat ? R1 : R0;
      unsigned short* T = mat ? T1 : T0;
      if (R) *(uint2*)(R + rr * RLD + c4) = make_uint2(p01, p23);
      if (T) {
        T[(c4 + 0) * TLD + rr] = (unsigned short)(p01 & 0xFFFFu); T[(c4 + 1) * TLD + rr] = (unsigned short)(p01 >> 16);
        T[(c4 + 2) * TLD + rr] = (unsigned short)(p23 & 0xFFFFu); T[(c4 + 3) * TLD + rr] = (unsigned short)(p23 >> 16);
      }
    }
  }
};
// A operand of a product contracting over the 32 staged rows, from a transposed tile: lane = dim column, rows in register order
__device__ inline bf16x8t tfrag(const unsigned short* T, int col, int kh, int s_) {
  const unsigned short* p = T + col * TLD + 16 * s_ + 4 * kh;
  const uint2 a = *(const uint2*)p, b = *(const uint2*)(p + 8);
  return __builtin_bit_cast(bf16x8t, u32x4t{a.x, a.y, b.x, b.y});
}

// MODE_BWD false: forward with dropout, writes out (+ bf16 copy) and the row statistics;  true: D_i = dO.O and dQ
template <int HD, bool BWD, int NW>      // NW waves = 32 NW queries per workgroup
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_hollow_attn_q_mfma(const AttnTrainArgs a) {
  constexpr int KS = HD / 16, RLD = HD + RLD16;
  __shared__ __attribute__((aligned(16))) unsigned short Kr[32 * RLD];          // K rows [key][dim]
  __shared__ __attribute__((aligned(16))) unsigned short Vr[BWD ? 32 * RLD : 8];   // V rows (backward: dP^T = V dO^T)
  __shared__ __attribute__((aligned(16))) unsigned short Tt[32 * TLD];          // forward: V^T;  backward: K^T   ([dim][key], dims >= HD zero)
  const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, kh = lane >> 5;
  const int q0 = blockIdx.x * (32 * NW) + wave * 32, i = q0 + col;
  const bool qok = i < a.Tq;
  bf16x8t qf[KS], dof[KS];
  row_frags<HD>(a.q + (size_t)b * a.q_bs + (size_t)(qok ? i : 0) * a.q_rs + h * HD, qok, a.scale, kh, qf);
  float m = BWD ? 0.0f : -INFINITY, l = 0.0f, inv_l = 0.0f, Di = 0.0f;
  float* st = a.stats + (((size_t)b * a.H + h) * a.Tq + (qok ? i : 0)) * 4;
  if (BWD) {
    const float* dr = a.d_out + ((size_t)b * a.Tq + (qok ? i : 0)) * a.out_rs + h * HD;
    const float* orow = a.out + ((size_t)b * a.Tq + (qok ? i : 0)) * a.out_rs + h * HD;
    row_frags<HD>(dr, qok, 1.0f, kh, dof);
    if (qok) {
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
        for (int c = 0; c < 8; ++c) Di = fmaf(dr[16 * s_ + 8 * kh + c], orow[16 * s_ + 8 * kh + c], Di);
      m = st[0]; inv_l = 1.0f / st[1];
    }
    Di += __shfl_xor(Di, 32, WAVE);
    if (qok && kh == 0) st[2] = Di;
  }
  const float inv_keep = attn_inv_keep(a.drop_p);
  const uint64_t octs = (uint64_t)(a.Tk + 7) / 8;
  const uint64_t qrow = (((uint64_t)b * a.H + h) * a.Tq + (qok ? i : 0)) * octs;
  const unsigned thr = attn_thr(a.drop_p);
  const uint64_t rseed = a.drop_p > 0.0f ? a.rng[0] : 0, rctr = a.drop_p > 0.0f ? a.rng[1] * 4096u + a.layer : 0;   // (once: not per chunk)
  f32x16t oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.0f;
  for (int idx = threadIdx.x; idx < 32 * TLD / 2; idx += 64 * NW) ((unsigned*)Tt)[idx] = 0u;   // (rows >= HD stay zero)
  const int wlo = blockIdx.x * (32 * NW), whi = min(wlo + 32 * NW, a.Tq) - 1, mylo = q0, myhi = min(q0 + 32, a.Tq) - 1;
  const float* kb = a.k + (size_t)b * a.k_bs + h * HD;
  const float* vb = a.v + (size_t)b * a.v_bs + h * HD;
  Stage2<HD, 64 * NW> stg;
  auto next_visible = [&](int jc) {                                             // (uniform over the workgroup)
    while (jc < a.Tk && !chunk_any(a.mode, a.Tq, a.Tk, jc, wlo, whi)) jc += 32;
    return jc;
  };
  int jnext = next_visible(0);
  if (jnext < a.Tk) stg.fetch(kb, a.k_rs, vb, a.v_rs, jnext, a.Tk);
  while (jnext < a.Tk) {
    const int j0 = jnext;
    __syncthreads();
    stg.store(1.0f, Kr, BWD ? Tt : nullptr, BWD ? Vr : nullptr, BWD ? nullptr : Tt);
    __syncthreads();
    jnext = next_visible(j0 + 32);
    if (jnext < a.Tk) stg.fetch(kb, a.k_rs, vb, a.v_rs, jnext, a.Tk);
    if (!chunk_any(a.mode, a.Tq, a.Tk, j0, mylo, myhi)) continue;               // (wave-uniform; no barrier below)
    f32x16t sacc, pacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.0f; pacc[r] = 0.0f; }
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      const bf16x8t kf = *(const bf16x8t*)(Kr + col * RLD + 16 * s_ + 8 * kh);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s_], sacc, 0, 0, 0);
      if (BWD) {
        const bf16x8t vf = *(const bf16x8t*)(Vr + col * RLD + 16 * s_ + 8 * kh);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[s_], pacc, 0, 0, 0);
      }
    }
    const bool full = chunk_full(a.mode, a.Tq, a.Tk, j0, q0);
    if (!full) {
      if (a.mode == 0) mask_tile_q<0>(sacc, a.Tq, a.Tk, i, qok, j0 + 4 * kh);
      else if (a.mode == 1) mask_tile_q<1>(sacc, a.Tq, a.Tk, i, qok, j0 + 4 * kh);
      else mask_tile_q<2>(sacc, a.Tq, a.Tk, i, qok, j0 + 4 * kh);
    }
    unsigned keep = 0xFFFFu;                                                     // bit r: probability r survives the dropout
    if (a.drop_p > 0.0f) {
      // registers 4 g .. 4 g + 3 are keys 8 g + 4 kh + 0..3 of the chunk: bits 4 kh .. 4 kh + 3 of octet g.  The two lanes of a
      // query split the four octets (half kh draws octets 2 kh, 2 kh + 1) and exchange them
      const unsigned mine = keep8v(rseed, rctr, qrow + (uint64_t)(j0 >> 3) + 2 * kh, thr) | (keep8v(rseed, rctr, qrow + (uint64_t)(j0 >> 3) + 2 * kh + 1, thr) << 8);
      const unsigned other = (unsigned)__shfl_xor((int)mine, 32, WAVE);
      const unsigned all = kh ? (other | (mine << 16)) : (mine | (other << 16));   // octet g at bits 8 g
      keep = 0u;
#pragma unroll
      for (int g = 0; g < 4; ++g) keep |= ((all >> (8 * g + 4 * kh)) & 0xFu) << (4 * g);
    }
    unsigned pw[8];
    if (!BWD) {
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
      const float mn = fmaxf(m, mx);
      const float msafe = mn == -INFINITY ? 0.0f : mn;
      const float corr = __expf(m - msafe);
      float rs = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __expf(sacc[r] - msafe), p1 = __expf(sacc[r + 1] - msafe);
        rs += p0 + p1;
        pw[r >> 1] = ht_pack2((keep >> r) & 1u ? p0 * inv_keep : 0.0f, (keep >> (r + 1)) & 1u ? p1 * inv_keep : 0.0f);
      }
      rs += __shfl_xor(rs, 32, WAVE);
      l = l * corr + rs;
      m = mn;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] *= corr;
    } else {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __expf(sacc[r] - m) * inv_l, p1 = __expf(sacc[r + 1] - m) * inv_l;       // exp(-inf) = 0 for masked pairs
        const float d0 = (keep >> r) & 1u ? pacc[r] * inv_keep : 0.0f, d1 = (keep >> (r + 1)) & 1u ? pacc[r + 1] * inv_keep : 0.0f;
        pw[r >> 1] = ht_pack2(p0 * (d0 - Di), p1 * (d1 - Di));
      }
    }
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const bf16x8t pf = __builtin_bit_cast(bf16x8t, u32x4t{pw[4 * s_], pw[4 * s_ + 1], pw[4 * s_ + 2], pw[4 * s_ + 3]});
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(Tt, col, kh, s_), pf, oacc, 0, 0, 0);
    }
  }
  if (!qok) return;
  if (!BWD && kh == 0) { st[0] = m; st[1] = l; }
  const float f = BWD ? a.scale : 1.0f / l;
  const size_t oo = BWD ? (size_t)b * a.dq_bs + (size_t)i * a.dq_rs + h * HD : ((size_t)b * a.Tq + i) * a.out_rs + h * HD;
  float* o32 = BWD ? a.dq : a.out;
  unsigned short* o16 = BWD ? a.dq_bf16 : a.out_bf16;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int d0 = 8 * r4 + 4 * kh;                                            // dims d0..d0+3 = registers 4 r4 .. 4 r4 + 3
    if (d0 < HD) {
      const float4 v = make_float4(oacc[4 * r4] * f, oacc[4 * r4 + 1] * f, oacc[4 * r4 + 2] * f, oacc[4 * r4 + 3] * f);
      if (o32) *(float4*)(o32 + oo + d0) = v;
      if (o16) *(uint2*)(o16 + oo + d0) = make_uint2(ht_pack2(v.x, v.y), ht_pack2(v.z, v.w));
    }
  }
}

// dK, dV: wave = 32 keys (lane = key), loop over chunks of 32 queries
// (held to 128 registers = four waves per SIMD: the unconstrained build takes 156 for three waves; the few spilled dwords cost less
//  than the lost wave -- dQ + dK/dV pair 196 -> 158 us at the maze shape, 408 -> 376 us at the MNIST hollow shape)
template <int HD, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_hollow_attn_kv_mfma(const AttnTrainArgs a) {
  constexpr int KS = HD / 16, RLD = HD + RLD16;
  __shared__ __attribute__((aligned(16))) unsigned short Qr[32 * RLD], Dr[32 * RLD];     // scale Q rows, dO rows  [query][dim]
  __shared__ __attribute__((aligned(16))) unsigned short Qt[32 * TLD], Dt[32 * TLD];     // their transposes [dim][query]
  __shared__ float Sm[32], Sl[32], Sd[32];                                                // per query: max, 1 / sum, D_i
  const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, kh = lane >> 5;
  const int k0 = blockIdx.x * (32 * NW) + wave * 32, j = k0 + col;
  const bool kok = j < a.Tk;
  bf16x8t kf[KS], vf[KS];
  row_frags<HD>(a.k + (size_t)b * a.k_bs + (size_t)(kok ? j : 0) * a.k_rs + h * HD, kok, 1.0f, kh, kf);
  row_frags<HD>(a.v + (size_t)b * a.v_bs + (size_t)(kok ? j : 0) * a.v_rs + h * HD, kok, 1.0f, kh, vf);
  const float inv_keep = attn_inv_keep(a.drop_p);
  const uint64_t octs = (uint64_t)(a.Tk + 7) / 8;
  const unsigned thr = attn_thr(a.drop_p);
  const uint64_t rseed = a.drop_p > 0.0f ? a.rng[0] : 0, rctr = a.drop_p > 0.0f ? a.rng[1] * 4096u + a.layer : 0;
  f32x16t kacc, vacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) { kacc[r] = 0.0f; vacc[r] = 0.0f; }
  for (int idx = threadIdx.x; idx < 32 * TLD / 2; idx += 64 * NW) { ((unsigned*)Qt)[idx] = 0u; ((unsigned*)Dt)[idx] = 0u; }
  const int wlo = blockIdx.x * (32 * NW), whi = min(wlo + 32 * NW, a.Tk) - 1, mylo = k0, myhi = min(k0 + 32, a.Tk) - 1;
  // does the chunk of queries i0.. see any key of [lo, hi]?  (the mask read from the key side)
  auto q_any = [&](int i0, int lo, int hi) {
    const int il = min(i0 + 32, a.Tq) - 1;
    if (hi < lo) return false;
    if (a.mode == 0) return lo <= il;                                          // j <= i for some pair
    if (a.mode == 1) return hi >= i0;
    if (lo == 0) return true;
    const bool first = lo <= a.Tq && lo - 1 <= il;                             // keys 1..Tq: j - 1 <= i
    const bool second = hi > a.Tq && hi - a.Tq - 1 >= i0;                      // keys Tq+1..: j - Tq - 1 >= i
    return first || second;
  };
  const float* qb = a.q + (size_t)b * a.q_bs + h * HD;
  const float* db = a.d_out + (size_t)b * a.Tq * a.out_rs + h * HD;
  const float* sb = a.stats + ((size_t)b * a.H + h) * a.Tq * 4;
  Stage2<HD, 64 * NW> stg;
  float pm = 0.0f, pl = 0.0f, pd = 0.0f;                                         // the chunk's row statistics, fetched ahead (threads 0..31)
  auto next_visible = [&](int ic) {                                             // (uniform over the workgroup)
    while (ic < a.Tq && !q_any(ic, wlo, whi)) ic += 32;
    return ic;
  };
  auto fetch = [&](int ic) {
    stg.fetch(qb, a.q_rs, db, a.out_rs, ic, a.Tq);
    if (threadIdx.x < 32) {
      const int i = ic + threadIdx.x;
      const bool ok = i < a.Tq;
      pm = ok ? sb[(size_t)i * 4] : 0.0f;
      pl = ok ? sb[(size_t)i * 4 + 1] : 1.0f;
      pd = ok ? sb[(size_t)i * 4 + 2] : 0.0f;
    }
  };
  // (no fetch-ahead here: the two accumulators + fragments already hold this kernel at 3 waves per SIMD, and the extra
  //  registers of a chunk in flight cost more than the exposed load latency -- measured 158 us vs 183 us per call)
  for (int i0 = next_visible(0); i0 < a.Tq; i0 = next_visible(i0 + 32)) {
    fetch(i0);
    __syncthreads();
    stg.store(a.scale, Qr, Qt, Dr, Dt);
    if (threadIdx.x < 32) {
      const bool ok = i0 + (int)threadIdx.x < a.Tq;
      Sm[threadIdx.x] = pm; Sl[threadIdx.x] = ok ? 1.0f / pl : 0.0f; Sd[threadIdx.x] = pd;
    }
    __syncthreads();
    if (!q_any(i0, mylo, myhi)) continue;                                       // (wave-uniform; no barrier below)
    f32x16t sacc, pacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.0f; pacc[r] = 0.0f; }
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      const bf16x8t qa = *(const bf16x8t*)(Qr + col * RLD + 16 * s_ + 8 * kh);
      const bf16x8t da = *(const bf16x8t*)(Dr + col * RLD + 16 * s_ + 8 * kh);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[s_], sacc, 0, 0, 0);       // S[query][key]
      pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[s_], pacc, 0, 0, 0);       // dP[query][key]
    }
    unsigned keep = 0xFFFFu;                                                     // bit r: probability (query r, this key) survives
    if (a.drop_p > 0.0f) {
      // the Philox block of (query, eight consecutive keys) serves eight neighbouring lanes: lane c of them draws the blocks of
      // the registers 2 c, 2 c + 1 and the eight exchange them
      unsigned mine = 0u;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int r = 2 * (col & 7) + e, i = i0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const uint64_t oct = (((uint64_t)b * a.H + h) * a.Tq + (uint64_t)min(i, a.Tq - 1)) * octs + (uint64_t)(min(j, a.Tk - 1) >> 3);
        mine |= keep8v(rseed, rctr, oct, thr) << (8 * e);
      }
      keep = 0u;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const unsigned w = (unsigned)__shfl((int)mine, (lane & ~7) | c, WAVE);  // blocks of the registers 2 c, 2 c + 1
        keep |= (((w >> (j & 7)) & 1u) << (2 * c)) | (((w >> (8 + (j & 7))) & 1u) << (2 * c + 1));
      }
    }
    const unsigned okm = a.mode == 0 ? mask_bits_kv<0>(a.Tq, j, kok, i0 + 4 * kh)
                       : a.mode == 1 ? mask_bits_kv<1>(a.Tq, j, kok, i0 + 4 * kh) : mask_bits_kv<2>(a.Tq, j, kok, i0 + 4 * kh);
    unsigned pw[8], dw[8];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      float p[2], ds[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int il = ((r + e) & 3) + 8 * ((r + e) >> 2) + 4 * kh;
        const bool ok = (okm >> (r + e)) & 1u;
        const float pr = ok ? __expf(sacc[r + e] - Sm[il]) * Sl[il] : 0.0f;
        const bool kp = (keep >> (r + e)) & 1u;
        p[e] = kp ? pr * inv_keep : 0.0f;
        ds[e] = pr * ((kp ? pacc[r + e] * inv_keep : 0.0f) - Sd[il]);
      }
      pw[r >> 1] = ht_pack2(p[0], p[1]);
      dw[r >> 1] = ht_pack2(ds[0], ds[1]);
    }
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const bf16x8t pf = __builtin_bit_cast(bf16x8t, u32x4t{pw[4 * s_], pw[4 * s_ + 1], pw[4 * s_ + 2], pw[4 * s_ + 3]});
      const bf16x8t df = __builtin_bit_cast(bf16x8t, u32x4t{dw[4 * s_], dw[4 * s_ + 1], dw[4 * s_ + 2], dw[4 * s_ + 3]});
      vacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(Dt, col, kh, s_), pf, vacc, 0, 0, 0);   // dV^T += dO^T P
      kacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(Qt, col, kh, s_), df, kacc, 0, 0, 0);   // dK^T += (scale Q)^T dS
    }
  }
  if (!kok) return;
  const size_t ok_ = (size_t)b * a.dk_bs + (size_t)j * a.dk_rs + h * HD, ov = (size_t)b * a.dv_bs + (size_t)j * a.dv_rs + h * HD;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int d0 = 8 * r4 + 4 * kh;
    if (d0 < HD) {
      const float4 kv = make_float4(kacc[4 * r4], kacc[4 * r4 + 1], kacc[4 * r4 + 2], kacc[4 * r4 + 3]);
      const float4 vv = make_float4(vacc[4 * r4], vacc[4 * r4 + 1], vacc[4 * r4 + 2], vacc[4 * r4 + 3]);
      if (a.dk) *(float4*)(a.dk + ok_ + d0) = kv;
      if (a.dv) *(float4*)(a.dv + ov + d0) = vv;
      if (a.dk_bf16) *(uint2*)(a.dk_bf16 + ok_ + d0) = make_uint2(ht_pack2(kv.x, kv.y), ht_pack2(kv.z, kv.w));
      if (a.dv_bf16) *(uint2*)(a.dv_bf16 + ov + d0) = make_uint2(ht_pack2(vv.x, vv.y), ht_pack2(vv.z, vv.w));
    }
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
    if (out) *(float4*)(out + v * 4) = make_float4(r[0], r[1], r[2], r[3]);
    if (out_bf16) *(uint2*)(out_bf16 + v * 4) = make_uint2(ht_pack2(r[0], r[1]), ht_pack2(r[2], r[3]));
  }
}

// bf16-only ReLU (+ dropout) for the MLP hidden tensor in the bf16 mode (the fp32 copy of a (rows, mlp_dim) tensor is the
// largest stream of a block): forward  u = dropout(relu(pre)) (in place allowed);  backward  dpre = du * [u != 0] / (1 - p)
// -- a dropped or clipped entry of u is exactly 0, so the saved output IS the mask and backward draws no random numbers.
__global__ __launch_bounds__(256) void k_hollow_relu_bf16(const unsigned short* __restrict__ src, const unsigned short* __restrict__ mask_u,
                                                         unsigned short* __restrict__ out, int64_t noct, float drop_p,
                                                         const uint64_t* rng, uint64_t layer) {
  const float inv_keep = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < noct; v += (int64_t)gridDim.x * 256) {
    const uint4 in = *(const uint4*)(src + v * 8);
    const unsigned w[4] = {in.x, in.y, in.z, in.w};
    unsigned o[4];
    if (mask_u) {                                                            // backward: src = du, mask_u = saved u
      const uint4 mu = *(const uint4*)(mask_u + v * 8);
      const unsigned m[4] = {mu.x, mu.y, mu.z, mu.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a0 = (m[k] & 0x7FFFu) ? __uint_as_float(w[k] << 16) * inv_keep : 0.0f;
        const float a1 = (m[k] & 0x7FFF0000u) ? __uint_as_float(w[k] & 0xFFFF0000u) * inv_keep : 0.0f;
        o[k] = ht_pack2(a0, a1);
      }
    } else {
      unsigned keep = 0xFFu;
      if (drop_p > 0.0f) keep = keep4(rng, layer, (uint64_t)(2 * v), drop_p) | (keep4(rng, layer, (uint64_t)(2 * v + 1), drop_p) << 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a0 = fmaxf(__uint_as_float(w[k] << 16), 0.0f) * ((keep >> (2 * k)) & 1u ? inv_keep : 0.0f);
        const float a1 = fmaxf(__uint_as_float(w[k] & 0xFFFF0000u), 0.0f) * ((keep >> (2 * k + 1)) & 1u ? inv_keep : 0.0f);
        o[k] = ht_pack2(a0, a1);
      }
    }
    *(uint4*)(out + v * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// out = dropout(x) (+ res), to fp32 and / or bf16 (p = 0: a plain add / cast).  One pass where the block Functions ran a clone, an
// in-place dropout, an add or a cast; masks Philox(seed, step * 4096 + layer, quad) as k_hollow_act.
__global__ __launch_bounds__(256) void k_hollow_dropout(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ out,
                                                       unsigned short* __restrict__ out_bf16, int64_t nquad, float drop_p,
                                                       const uint64_t* rng, uint64_t layer) {
  const float inv_keep = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const uint64_t seed = drop_p > 0.0f ? rng[0] : 0, ctr = drop_p > 0.0f ? rng[1] * 4096u + layer : 0;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nquad; v += (int64_t)gridDim.x * 256) {
    const float4 x4 = *(const float4*)(x + v * 4);
    float r[4] = {x4.x, x4.y, x4.z, x4.w};
    if (drop_p > 0.0f) {
      const unsigned keep = keep4v(seed, ctr, (uint64_t)v, drop_p);
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = (keep >> k) & 1u ? r[k] * inv_keep : 0.0f;
    }
    if (res) { const float4 q4 = *(const float4*)(res + v * 4); r[0] += q4.x; r[1] += q4.y; r[2] += q4.z; r[3] += q4.w; }
    if (out) *(float4*)(out + v * 4) = make_float4(r[0], r[1], r[2], r[3]);
    if (out_bf16) *(uint2*)(out_bf16 + v * 4) = make_uint2(ht_pack2(r[0], r[1]), ht_pack2(r[2], r[3]));
  }
}

// ============================================================================ column sums (bias gradients), two stages, no atomics
// partial[blk % nrep][n] (+)= sum over the workgroup's run of rows of x[row][n]; ctdd_unet_sum_batch adds the nrep partials.  A thread keeps
// eight consecutive columns (one 16-byte load of bf16, two of fp32) of every (256 / (N / 8))-th row of the run.
__global__ __launch_bounds__(256) void k_hollow_colsum(const float* __restrict__ f, const unsigned short* __restrict__ h, int64_t rows,
                                                      int N, int ld, float* __restrict__ partial, int nrep) {
  extern __shared__ __attribute__((aligned(16))) float sm[];                  // [rpi][N]
  const int tpr = N / 8, rpi = 256 / tpr;                                      // threads per row, rows per iteration
  const int c8 = (threadIdx.x % tpr) * 8, rl = threadIdx.x / tpr;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x, lo = (int64_t)blockIdx.x * per, hi = min(lo + per, rows);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < rpi) {
    for (int64_t r = lo + rl; r < hi; r += rpi) {
      if (h) {
        const uint4 v = *(const uint4*)(h + (size_t)r * ld + c8);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[2 * k] += __uint_as_float(w[k] << 16); acc[2 * k + 1] += __uint_as_float(w[k] & 0xFFFF0000u); }
      } else {
        const float4 v0 = *(const float4*)(f + (size_t)r * ld + c8), v1 = *(const float4*)(f + (size_t)r * ld + c8 + 4);
        acc[0] += v0.x; acc[1] += v0.y; acc[2] += v0.z; acc[3] += v0.w; acc[4] += v1.x; acc[5] += v1.y; acc[6] += v1.z; acc[7] += v1.w;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sm[rl * N + c8 + k] = acc[k];
  }
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += 256) {
    float t = 0.0f;
    for (int q = 0; q < rpi; ++q) t += sm[q * N + n];
    if (nrep >= (int)gridDim.x) partial[(size_t)blockIdx.x * N + n] = t;
    else atomicAdd(partial + (size_t)(blockIdx.x % nrep) * N + n, t);          // few workgroups per address: no serialisation to speak of
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
  const int NV = a.E > 256 ? 2 : 1, LPR = a.E / (4 * NV);
  auto al16 = [](const void* p_) { return ((uintptr_t)p_ & 15) == 0; };
  const bool vec = a.E % (4 * NV) == 0 && LPR >= 1 && LPR <= 64 && 64 % LPR == 0 && a.x_bs % 4 == 0 && a.y_bs % 4 == 0 && a.dout_bs % 4 == 0 &&
                   a.dx_bs % 4 == 0 && a.dy_bs % 4 == 0 && a.dres_bs % 4 == 0 && a.film_stride % 4 == 0 && al16(a.x) && al16(a.y) && al16(a.dout) &&
                   al16(a.dx) && al16(a.dy) && al16(a.dres) && al16(a.gamma) && al16(a.beta) && al16(a.film);
  // rows per wave (0 = the kernel's own choice): one iteration of its row groups -- (4 or 2 groups) x (64 / LPR rows) on the
  // vector path, four rows on the column-per-lane path
  if (a.rpw <= 0) a.rpw = vec ? (NV == 1 ? 4 : 2) * (64 / LPR) : 4;
  const int64_t waves = (int64_t)a.B * ((a.T + a.rpw - 1) / a.rpw);
  const dim3 grid((unsigned)((waves + 3) / 4));
  {
    if (vec) {
      if (NV == 1) hipLaunchKernelGGL(k_hollow_ln_bwd_v4<1>, grid, dim3(256), 0, (hipStream_t)stream, a, LPR);
      else hipLaunchKernelGGL(k_hollow_ln_bwd_v4<2>, grid, dim3(256), 0, (hipStream_t)stream, a, LPR);
      return finish_launch("k_hollow_ln_bwd_v4");
    }
  }
  if (a.E <= 128) hipLaunchKernelGGL(k_hollow_ln_bwd<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (a.E <= 256) hipLaunchKernelGGL(k_hollow_ln_bwd<4>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_hollow_ln_bwd<8>, grid, dim3(256), 0, (hipStream_t)stream, a);
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

constexpr int ATT_NW = CTDD_ATT_NW;
#define HD_DISPATCH_MFMA(KERNEL16, KERNEL32, GRIDX)                                                                  \
  if (a.hd == 16) hipLaunchKernelGGL(KERNEL16, dim3(GRIDX, a.H, a.B), dim3(64 * ATT_NW), 0, (hipStream_t)stream, a); \
  else hipLaunchKernelGGL(KERNEL32, dim3(GRIDX, a.H, a.B), dim3(64 * ATT_NW), 0, (hipStream_t)stream, a);
static int attn_check_mfma(const AttnTrainArgs& a) {
  if (int rc = attn_check(a)) return rc;
  CTDD_REQUIRE(a.hd == 16 || a.hd == 32, CTDD_ERANGE, "attention (training, matrix cores): head dimension %d (16 or 32)", a.hd);
  CTDD_REQUIRE(a.q_rs % 4 == 0 && a.k_rs % 4 == 0 && a.v_rs % 4 == 0 && a.out_rs % 4 == 0 && a.q_bs % 4 == 0 && a.k_bs % 4 == 0 && a.v_bs % 4 == 0,
               CTDD_EINVAL, "attention (training, matrix cores): rows must be 16-byte aligned");
  return CTDD_OK;
}
extern "C" int ctdd_hollow_attention_train_bf16(const void* args_, void* stream) {
  const AttnTrainArgs& a = *(const AttnTrainArgs*)args_;
  if (int rc = attn_check_mfma(a)) return rc;
  HD_DISPATCH_MFMA((k_hollow_attn_q_mfma<16, false, ATT_NW>), (k_hollow_attn_q_mfma<32, false, ATT_NW>), (a.Tq + 32 * ATT_NW - 1) / (32 * ATT_NW))
  return finish_launch("k_hollow_attn_q_mfma (forward)");
}
extern "C" int ctdd_hollow_attention_bwd_bf16(const void* args_, void* stream) {
  const AttnTrainArgs& a = *(const AttnTrainArgs*)args_;
  if (int rc = attn_check_mfma(a)) return rc;
  CTDD_REQUIRE(a.d_out && (a.dq || a.dq_bf16) && (a.dk || a.dk_bf16) && (a.dv || a.dv_bf16), CTDD_EINVAL, "attention bwd: null gradient buffer");
  CTDD_REQUIRE(a.dq_rs % 4 == 0 && a.dk_rs % 4 == 0 && a.dv_rs % 4 == 0 && a.dq_bs % 4 == 0 && a.dk_bs % 4 == 0 && a.dv_bs % 4 == 0, CTDD_EINVAL,
               "attention bwd (matrix cores): gradient rows must be 16-byte aligned");
  HD_DISPATCH_MFMA((k_hollow_attn_q_mfma<16, true, ATT_NW>), (k_hollow_attn_q_mfma<32, true, ATT_NW>), (a.Tq + 32 * ATT_NW - 1) / (32 * ATT_NW))
  if (int rc = finish_launch("k_hollow_attn_q_mfma (dQ)")) return rc;
  HD_DISPATCH_MFMA((k_hollow_attn_kv_mfma<16, ATT_NW>), (k_hollow_attn_kv_mfma<32, ATT_NW>), (a.Tk + 32 * ATT_NW - 1) / (32 * ATT_NW))
  return finish_launch("k_hollow_attn_kv_mfma");
}

extern "C" int ctdd_hollow_act(const float* pre, const float* dout, float* out, void* out_bf16, int64_t n, int act, float drop_p,
                               const uint64_t* rng, uint64_t layer, void* stream) {
  CTDD_REQUIRE(pre && (out || out_bf16) && n > 0 && n % 4 == 0 && act >= 0 && act <= 2, CTDD_EINVAL, "act: n=%lld act=%d", (long long)n, act);
  CTDD_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f && (drop_p == 0.0f || rng), CTDD_EINVAL, "act: dropout %g", (double)drop_p);
  int64_t g = (n / 4 + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_hollow_act, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, pre, dout, out, (unsigned short*)out_bf16, n / 4, act,
                     drop_p, rng, layer);
  return finish_launch("k_hollow_act");
}

extern "C" int ctdd_hollow_colsum(const float* x_f32, const void* x_bf16, int64_t rows, int N, int ld, float* partial, int nblk, int nrep,
                                  void* stream) {
  CTDD_REQUIRE((x_f32 || x_bf16) && partial && rows > 0 && nblk > 0 && nrep > 0 && N >= 8 && N % 8 == 0 && N <= 2048 && ld % 8 == 0 && N <= ld, CTDD_EINVAL,
               "colsum: rows=%lld N=%d ld=%d nblk=%d", (long long)rows, N, ld, nblk);
  const int rpi = 256 / (N / 8);
  const size_t lds = (size_t)rpi * N * sizeof(float);                          // <= 8 KiB
  hipLaunchKernelGGL(k_hollow_colsum, dim3(nblk), dim3(256), lds, (hipStream_t)stream, x_f32, (const unsigned short*)x_bf16, rows, N, ld, partial, nrep);
  return finish_launch("k_hollow_colsum");
}

extern "C" int ctdd_hollow_dropout(const float* x, const float* res, float* out, void* out_bf16, int64_t n, float drop_p, const uint64_t* rng,
                                   uint64_t layer, void* stream) {
  CTDD_REQUIRE(x && (out || out_bf16) && n > 0 && n % 4 == 0, CTDD_EINVAL, "dropout: n=%lld", (long long)n);
  CTDD_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f && (drop_p == 0.0f || rng), CTDD_EINVAL, "dropout: rate %g", (double)drop_p);
  int64_t g = (n / 4 + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_hollow_dropout, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, res, out, (unsigned short*)out_bf16, n / 4, drop_p,
                     rng, layer);
  return finish_launch("k_hollow_dropout");
}

extern "C" int ctdd_hollow_relu_bf16(const void* src, const void* mask_u, void* out, int64_t n, float drop_p, const uint64_t* rng,
                                     uint64_t layer, void* stream) {
  CTDD_REQUIRE(src && out && n > 0 && n % 8 == 0, CTDD_EINVAL, "relu (bf16): n=%lld", (long long)n);
  CTDD_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f && (drop_p == 0.0f || mask_u || rng), CTDD_EINVAL, "relu (bf16): dropout %g", (double)drop_p);
  int64_t g = (n / 8 + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_hollow_relu_bf16, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src,
                     (const unsigned short*)mask_u, (unsigned short*)out, n / 8, drop_p, rng, layer);
  return finish_launch("k_hollow_relu_bf16");
}

extern "C" int ctdd_hollow_embed_bwd(const void* args_, void* stream) {
  const EmbedBwdArgs& a = *(const EmbedBwdArgs*)args_;
  CTDD_REQUIRE((a.x64 || a.x32) && a.dl2r && a.dr2l && a.dw && a.db, CTDD_EINVAL, "embed bwd: null buffer");
  hipLaunchKernelGGL(k_hollow_embed_bwd, dim3(16, a.B), dim3(256), 0, (hipStream_t)stream, a);
  return finish_launch("k_hollow_embed_bwd");
}
