// hollow_kernels.hip -- inference kernels of the SDDM hollow (bidirectional-causal) transformer
// (reference lib/networks/hollow_networks.py: BidirectionalTransformer2 668-755, UniDirectionalTransformer
// 497-568, SelfAttentionBlock 311-340, CrossAttention 204-280, AttentionReadout 283-308,
// ResidualReadout 90-132, PositionalEncoding 1136-1156).
// The linear layers run on the GEMM kernels of unet_kernels.hip (a token is a "pixel", the weight a 1x1 segment; ReLU /
// GELU in the epilogue): the exact-fp32 matrix kernel in the fp32 mode, the bf16 slab kernel in the bf16 and bf16x3 modes
// (bf16x3: operands as hi + lo bf16 pairs, three products per contraction).  This file holds what is not a GEMM:
//   k_hollow_embed          state -> the two shifted token sequences [temb | x_0..x_{D-2}] and [x_1..x_{D-1} | temb]
//   k_hollow_layernorm      LayerNorm of (x [+ y]) with optional per-sample FiLM, strided destination, fp32 / bf16 hi (+ lo) outputs
//   k_hollow_add            a + b over strided batches (l2r + r2l)
//   k_hollow_attention      fp32 masked multi-head attention with an online softmax: causal (j <= i), anti-causal
//                           (j >= i) and the readout mask [temb | l2r j <= i | r2l j >= i]
//   k_hollow_attention_mfma the same on the bf16 matrix cores (single or hi + lo operands), fp32 softmax
#include "common.hpp"

namespace ctdd {

__device__ inline unsigned short hk_bf16(float a) {            // round-to-nearest-even, as the hardware conversion
  using v2f = __attribute__((ext_vector_type(2))) float;
  using v2b = __attribute__((ext_vector_type(2))) __bf16;
  v2f v = {a, 0.0f};
  return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(v, v2b)) & 0xFFFFu);
}

// second bf16 term of a value: v ~ hi + lo to ~2^-17 relative (the "split" GEMM operands)
__device__ inline unsigned short hk_lo(float v, unsigned short hi) { return hk_bf16(v - __uint_as_float((unsigned)hi << 16)); }

// ------------------------------------------------------------------ embedding (hollow_networks.py:729-753, 534-563)
struct HollowEmbedArgs {
  const int64_t* x64; const int32_t* x32;       // (B, D) states
  const float* t;                                 // (B)
  const float* w_in; const float* b_in;           // Linear(1 -> E): weight[:, 0], bias
  const float* pe;                                // (>= D, E) positional table of each direction (identical construction)
  int B, D, E, S; float temb_scale;
  float* l2r; float* r2l;                         // (B, D, E)
  float* temb;                                    // (B, E)
};
__global__ __launch_bounds__(256) void k_hollow_embed(const HollowEmbedArgs a) {
  const int b = blockIdx.y, E = a.E, D = a.D, half = E / 2;
  const float tv = a.t[b] * a.temb_scale;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < D * E; i += gridDim.x * 256) {
    const int j = i / E, e = i % E;
    const int fe = e < half ? e : e - half;
    const float freq = expf((float)fe * -(logf(10000.0f) / (float)(half - 1)));
    const float arg = tv * freq;
    const float te = e < half ? sinf(arg) : cosf(arg);
    if (j == 0 && a.temb) a.temb[(size_t)b * E + e] = te;
    auto emb = [&](int d) {
      const float xr = a.x64 ? (float)a.x64[(size_t)b * D + d] : (float)a.x32[(size_t)b * D + d];
      const float xn = (xr / (float)(a.S - 1)) * 2.0f - 1.0f;
      return xn * a.w_in[e] + a.b_in[e];
    };
    const float p = a.pe[(size_t)j * E + e];
    a.l2r[((size_t)b * D + j) * E + e] = (j == 0 ? te : emb(j - 1)) + p;
    a.r2l[((size_t)b * D + j) * E + e] = (j == D - 1 ? te : emb(j + 1)) + p;
  }
}

// ------------------------------------------------------------------ LayerNorm (+ add, + FiLM)
// rows = B * T.  in row (b, j) at x + (b * x_bs + j * E); out row at out + (b * out_bs + j * E).
struct HollowLnArgs {
  const float* x; const float* y;                 // y optional: LN(x + y)
  int64_t x_bs, y_bs, out_bs;                     // batch strides in floats
  const float* gamma; const float* beta; float eps;
  const float* film; int film_stride;             // optional (B, 2E): out = a * LN + b, a = film[b][0:E], b = film[b][E:2E]
  int B, T, E;
  float* out;
  unsigned short* out_hi; int64_t out_hi_bs;      // optional bf16 copy (GEMM operand in the bf16 mode)
  unsigned short* out_lo;                         // optional second bf16 term, bf16(v - out_hi) (split mode), stride out_hi_bs
};
__global__ __launch_bounds__(256) void k_hollow_layernorm(const HollowLnArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)a.B * a.T) return;
  const int b = (int)(row / a.T), j = (int)(row % a.T), E = a.E;
  const float* x = a.x + (size_t)b * a.x_bs + (size_t)j * E;
  const float* y = a.y ? a.y + (size_t)b * a.y_bs + (size_t)j * E : nullptr;
  float* o = a.out ? a.out + (size_t)b * a.out_bs + (size_t)j * E : nullptr;
  unsigned short* oh = a.out_hi ? a.out_hi + (size_t)b * a.out_hi_bs + (size_t)j * E : nullptr;
  unsigned short* ol = (oh && a.out_lo) ? a.out_lo + (size_t)b * a.out_hi_bs + (size_t)j * E : nullptr;
  auto emit = [&](int e, float v) {
    if (a.film) v = a.film[(size_t)b * a.film_stride + e] * v + a.film[(size_t)b * a.film_stride + E + e];
    if (o) o[e] = v;
    if (oh) {
      const unsigned short hv = hk_bf16(v);
      oh[e] = hv;
      if (ol) ol[e] = hk_lo(v, hv);
    }
  };
  if (E <= 512) {                                   // the row lives in registers: one pass over memory
    float xv[8];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int e = lane + 64 * k;
      xv[k] = e < E ? x[e] + (y ? y[e] : 0.0f) : 0.0f;
      s += xv[k];
    }
    const float mean = wave_sum(s) / (float)E;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float d = lane + 64 * k < E ? xv[k] - mean : 0.0f;
      q = fmaf(d, d, q);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + a.eps);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int e = lane + 64 * k;
      if (e < E) emit(e, (xv[k] - mean) * rstd * a.gamma[e] + a.beta[e]);
    }
    return;
  }
  float s = 0.0f;
  for (int e = lane; e < E; e += 64) s += x[e] + (y ? y[e] : 0.0f);
  const float mean = wave_sum(s) / (float)E;
  float q = 0.0f;
  for (int e = lane; e < E; e += 64) { const float d = x[e] + (y ? y[e] : 0.0f) - mean; q = fmaf(d, d, q); }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + a.eps);
  for (int e = lane; e < E; e += 64) emit(e, (x[e] + (y ? y[e] : 0.0f) - mean) * rstd * a.gamma[e] + a.beta[e]);
}

// The same for widths whose rows are whole 16-byte vectors per lane (E = 4 NV LPR, LPR | 64: E = 128, 256, 512 here): a lane
// owns 4 NV consecutive-by-four columns, LPR lanes a row, so a wave instruction moves 64 / LPR rows as 16-byte loads and 8-byte
// bf16 stores (one row per wave with 4-byte loads and 2-byte stores ran at 1.5 TB/s), PASSES row groups in flight per wave.
template <int NV, int PASSES>
__global__ __launch_bounds__(256) void k_hollow_layernorm_v4(const HollowLnArgs a, int LPR) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, E = a.E;
  const int RPP = 64 / LPR, sub = lane % LPR, rsub = lane / LPR;
  const int64_t rows = (int64_t)a.B * a.T;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wv) * (PASSES * RPP);
  float4 g[NV], be[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int e = 4 * sub + 4 * LPR * v;
    g[v] = *(const float4*)(a.gamma + e);
    be[v] = *(const float4*)(a.beta + e);
  }
  float4 xv[PASSES][NV];
  int bb[PASSES], jj[PASSES];
  bool ok[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int64_t row = row0 + p * RPP + rsub;
    ok[p] = row < rows;
    const int64_t rc = ok[p] ? row : rows - 1;
    bb[p] = (int)(rc / a.T); jj[p] = (int)(rc - (int64_t)bb[p] * a.T);
    const float* x = a.x + (size_t)bb[p] * a.x_bs + (size_t)jj[p] * E;
    const float* y = a.y ? a.y + (size_t)bb[p] * a.y_bs + (size_t)jj[p] * E : nullptr;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int e = 4 * sub + 4 * LPR * v;
      float4 t = *(const float4*)(x + e);
      if (y) { const float4 u = *(const float4*)(y + e); t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
      xv[p][v] = t;
    }
  }
  auto lsum = [&](float (&v)[PASSES]) {             // over the LPR lanes of a row, all row groups at once (independent exchanges per step)
    for (int o = LPR >> 1; o >= 1; o >>= 1) {
#pragma unroll
      for (int p = 0; p < PASSES; ++p) v[p] += __shfl_xor(v[p], o, WAVE);
    }
  };
  float mean[PASSES], rstd[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    float s_ = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v) s_ += (xv[p][v].x + xv[p][v].y) + (xv[p][v].z + xv[p][v].w);
    mean[p] = s_;
  }
  lsum(mean);
#pragma unroll
  for (int p = 0; p < PASSES; ++p) mean[p] = mean[p] / (float)E;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    float q = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const float d0 = xv[p][v].x - mean[p], d1 = xv[p][v].y - mean[p], d2 = xv[p][v].z - mean[p], d3 = xv[p][v].w - mean[p];
      q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); q = fmaf(d2, d2, q); q = fmaf(d3, d3, q);
    }
    rstd[p] = q;
  }
  lsum(rstd);
#pragma unroll
  for (int p = 0; p < PASSES; ++p) rstd[p] = 1.0f / sqrtf(rstd[p] / (float)E + a.eps);
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    if (!ok[p]) continue;
    float* o = a.out ? a.out + (size_t)bb[p] * a.out_bs + (size_t)jj[p] * E : nullptr;
    unsigned short* oh = a.out_hi ? a.out_hi + (size_t)bb[p] * a.out_hi_bs + (size_t)jj[p] * E : nullptr;
    unsigned short* ol = (oh && a.out_lo) ? a.out_lo + (size_t)bb[p] * a.out_hi_bs + (size_t)jj[p] * E : nullptr;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int e = 4 * sub + 4 * LPR * v;
      float r[4] = {(xv[p][v].x - mean[p]) * rstd[p] * g[v].x + be[v].x, (xv[p][v].y - mean[p]) * rstd[p] * g[v].y + be[v].y,
                    (xv[p][v].z - mean[p]) * rstd[p] * g[v].z + be[v].z, (xv[p][v].w - mean[p]) * rstd[p] * g[v].w + be[v].w};
      if (a.film) {
        const float4 fa = *(const float4*)(a.film + (size_t)bb[p] * a.film_stride + e), fb = *(const float4*)(a.film + (size_t)bb[p] * a.film_stride + E + e);
        r[0] = fa.x * r[0] + fb.x; r[1] = fa.y * r[1] + fb.y; r[2] = fa.z * r[2] + fb.z; r[3] = fa.w * r[3] + fb.w;
      }
      if (o) *(float4*)(o + e) = make_float4(r[0], r[1], r[2], r[3]);
      if (oh) {
        const unsigned short h0 = hk_bf16(r[0]), h1 = hk_bf16(r[1]), h2 = hk_bf16(r[2]), h3 = hk_bf16(r[3]);
        *(uint2*)(oh + e) = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
        if (ol) {
          const unsigned short l0 = hk_lo(r[0], h0), l1 = hk_lo(r[1], h1), l2 = hk_lo(r[2], h2), l3 = hk_lo(r[3], h3);
          *(uint2*)(ol + e) = make_uint2((unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16));
        }
      }
    }
  }
}

// out[b][j][:] = p[b][j][:] + q[b][j][:], strided batches
__global__ __launch_bounds__(256) void k_hollow_add(const float* __restrict__ p, int64_t p_bs, const float* __restrict__ q, int64_t q_bs,
                                                   float* __restrict__ out, unsigned short* __restrict__ out_hi,
                                                   unsigned short* __restrict__ out_lo, int64_t out_bs, int64_t per_batch) {
  const int b = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_batch; i += (int64_t)gridDim.x * 256) {
    const float v = p[(size_t)b * p_bs + i] + q[(size_t)b * q_bs + i];
    if (out) out[(size_t)b * out_bs + i] = v;
    if (out_hi) {
      const unsigned short hv = hk_bf16(v);
      out_hi[(size_t)b * out_bs + i] = hv;
      if (out_lo) out_lo[(size_t)b * out_bs + i] = hk_lo(v, hv);
    }
  }
}
// rows of a (B, E) matrix into slot 0 of a (B, T, E) buffer
__global__ __launch_bounds__(256) void k_hollow_put_rows(const float* __restrict__ src, float* __restrict__ dst,
                                                        unsigned short* __restrict__ dst_hi, unsigned short* __restrict__ dst_lo,
                                                        int64_t dst_bs, int E) {
  const int b = blockIdx.x;
  for (int e = threadIdx.x; e < E; e += 256) {
    const float v = src[(size_t)b * E + e];
    if (dst) dst[(size_t)b * dst_bs + e] = v;
    if (dst_hi) {
      const unsigned short hv = hk_bf16(v);
      dst_hi[(size_t)b * dst_bs + e] = hv;
      if (dst_lo) dst_lo[(size_t)b * dst_bs + e] = hk_lo(v, hv);
    }
  }
}

// ------------------------------------------------------------------ masked attention, online softmax
// q rows (b, i): q + b*q_bs + i*q_rs + h*hd ; k / v likewise; out (b, i): out + (b*Tq + i)*E_out + h*hd.
// mode 0: key j allowed iff j <= i ; 1: j >= i ; 2 (readout, Tk = 2 Tq + 1): j == 0 | 1 <= j <= Tq: j-1 <= i | j > Tq: j-Tq-1 >= i.
struct HollowAttnArgs {
  const float* q; const float* k; const float* v;
  int64_t q_bs, k_bs, v_bs; int q_rs, k_rs, v_rs;
  int B, Tq, Tk, H, hd, mode; float scale;
  float* out; int out_rs;
  unsigned short* out_hi;                        // optional bf16 copy, same row stride
  unsigned short* out_lo;                        // optional second bf16 term of the output (split mode)
  int split;                                     // matrix-core kernel: 1 = three bf16 products per contraction (hi hi + lo hi + hi lo)
};
// One thread = one query with the whole head dimension in registers (q[HD], acc[HD]); a workgroup = 128
// consecutive queries of one (b, head); keys / values come through LDS in chunks of 32 and are read as
// wave-uniform (broadcast) float4s, so the inner loop is 2 HD FMAs per key against HD/2 LDS reads.
// Keys are consumed four at a time: one running-max update and five exps per four keys.
constexpr int AQ = 128, AK = 32;
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ inline f32x2 pk_fma(f32x2 x, f32x2 y, f32x2 z) { return __builtin_elementwise_fma(x, y, z); }   // v_pk_fma_f32
template <int HD>
__global__ __launch_bounds__(AQ) void k_hollow_attention(const HollowAttnArgs a) {
  __shared__ __attribute__((aligned(16))) float Ks[AK * HD];
  __shared__ __attribute__((aligned(16))) float Vs[AK * HD];
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * AQ;
  const int i = i0 + threadIdx.x;
  const bool qok = i < a.Tq;
  f32x2 qv[HD / 2], acc[HD / 2];                                // pairs of head dimensions: packed fp32 FMA
#pragma unroll
  for (int c = 0; c < HD / 2; ++c) { acc[c] = f32x2{0.0f, 0.0f}; qv[c] = f32x2{0.0f, 0.0f}; }
  if (qok) {
    const float* qr = a.q + (size_t)b * a.q_bs + (size_t)i * a.q_rs + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 u = *(const float4*)(qr + c);
      qv[c / 2] = f32x2{u.x * a.scale, u.y * a.scale};
      qv[c / 2 + 1] = f32x2{u.z * a.scale, u.w * a.scale};
    }
  }
  float m = -INFINITY, l = 0.0f;
  const int ilo = i0, ihi = min(i0 + AQ, a.Tq) - 1;           // query range of the workgroup
  auto allowed = [&](int j) {
    if (a.mode == 0) return j <= i;
    if (a.mode == 1) return j >= i;
    return j == 0 || (j <= a.Tq ? j - 1 <= i : j - a.Tq - 1 >= i);
  };
  for (int j0 = 0; j0 < a.Tk; j0 += AK) {
    const int j1 = min(j0 + AK, a.Tk) - 1;
    bool any;                                                  // does any query of the workgroup see this chunk? (uniform)
    if (a.mode == 0) any = j0 <= ihi;
    else if (a.mode == 1) any = j1 >= ilo;
    else any = j0 == 0 || (j0 <= a.Tq && j0 - 1 <= ihi) || (j1 > a.Tq && j1 - a.Tq - 1 >= ilo);
    if (!any) continue;
    __syncthreads();
    for (int idx = threadIdx.x; idx < AK * HD / 4; idx += AQ) {
      const int jj = idx / (HD / 4), c4 = idx % (HD / 4), j = j0 + jj;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (j < a.Tk) {
        kv = *(const float4*)(a.k + (size_t)b * a.k_bs + (size_t)j * a.k_rs + h * HD + c4 * 4);
        vv = *(const float4*)(a.v + (size_t)b * a.v_bs + (size_t)j * a.v_rs + h * HD + c4 * 4);
      }
      *(float4*)(Ks + jj * HD + c4 * 4) = kv;
      *(float4*)(Vs + jj * HD + c4 * 4) = vv;
    }
    __syncthreads();
    if (!qok) continue;
    const int nj = j1 - j0 + 1;
    for (int jj = 0; jj < nj; jj += 4) {
      float s[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x2 d2 = {0.0f, 0.0f};
        const float* kr = Ks + (jj + u) * HD;                  // (rows past nj hold zeros or stale keys: masked below)
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
          const float4 kv = *(const float4*)(kr + c);
          d2 = pk_fma(qv[c / 2], f32x2{kv.x, kv.y}, d2);
          d2 = pk_fma(qv[c / 2 + 1], f32x2{kv.z, kv.w}, d2);
        }
        const float d = d2.x + d2.y;
        s[u] = (jj + u < nj && allowed(j0 + jj + u)) ? d : -INFINITY;
      }
      const float mn = fmaxf(fmaxf(m, fmaxf(s[0], s[1])), fmaxf(s[2], s[3]));
      if (mn == -INFINITY) continue;                           // nothing visible yet
      const float corr = expf(m - mn);
      float p[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) p[u] = expf(s[u] - mn);       // exp(-inf) = 0 for masked keys
      l = l * corr + ((p[0] + p[1]) + (p[2] + p[3]));
      const f32x2 corr2 = {corr, corr};
#pragma unroll
      for (int c = 0; c < HD / 2; ++c) acc[c] *= corr2;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* vr = Vs + (jj + u) * HD;
        const f32x2 p2 = {p[u], p[u]};
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
          const float4 vv = *(const float4*)(vr + c);
          acc[c / 2] = pk_fma(p2, f32x2{vv.x, vv.y}, acc[c / 2]);
          acc[c / 2 + 1] = pk_fma(p2, f32x2{vv.z, vv.w}, acc[c / 2 + 1]);
        }
      }
      m = mn;
    }
  }
  if (qok) {
    const float inv = 1.0f / l;
    const size_t oo = ((size_t)b * a.Tq + i) * a.out_rs + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 v = make_float4(acc[c / 2].x * inv, acc[c / 2].y * inv, acc[c / 2 + 1].x * inv, acc[c / 2 + 1].y * inv);
      if (a.out) *(float4*)(a.out + oo + c) = v;
      if (a.out_hi) {
        unsigned short* oh = a.out_hi + oo + c;
        oh[0] = hk_bf16(v.x); oh[1] = hk_bf16(v.y); oh[2] = hk_bf16(v.z); oh[3] = hk_bf16(v.w);
        if (a.out_lo) {
          unsigned short* ol = a.out_lo + oo + c;
          ol[0] = hk_lo(v.x, oh[0]); ol[1] = hk_lo(v.y, oh[1]); ol[2] = hk_lo(v.z, oh[2]); ol[3] = hk_lo(v.w, oh[3]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------ bf16 matrix-core attention (throughput mode)
// Same contract as k_hollow_attention for head dimensions 16 and 32, with the two products on
// v_mfma_f32_32x32x16_bf16 and the softmax in fp32.  A wave owns 32 queries; per chunk of 32 keys it forms the
// TRANSPOSED score tile S^T[key][query] = K Q^T, so that a lane's 16 accumulator registers are 16 keys of ONE query
// (lane = query column): the online-softmax maximum and sum are 15 in-register steps plus one exchange with the lane
// holding the other half of the keys.  The probabilities then ARE the B operand of O^T[dim][query] += V^T P^T without
// any data movement: k-step s of that product contracts the keys held in registers 8s..8s+7 of the two lane halves
// ({16s + 4kh + 0..3} U {16s + 8 + 4kh + 0..3} for half kh), and V^T is read from LDS in that key order.
// 4 waves = 128 queries of one (b, head) per workgroup; K / V chunks are converted to bf16 while staged.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4h = __attribute__((ext_vector_type(4))) unsigned;
__device__ inline unsigned hk_pack2(float a, float b) {
  using v2f = __attribute__((ext_vector_type(2))) float;
  using v2b = __attribute__((ext_vector_type(2))) __bf16;
  v2f v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, v2b));
}
template <int MODE>
__device__ inline void hk_mask_tile(f32x16& s, int Tq, int Tk, int i, bool qok, int jb) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = jb + (r & 3) + 8 * (r >> 2);
    bool ok = qok & (j < Tk);
    if (MODE == 0) ok &= j <= i;
    else if (MODE == 1) ok &= j >= i;
    else ok &= (j == 0) | ((j <= Tq) & (j - 1 <= i)) | ((j > Tq) & (j - Tq - 1 >= i));
    s[r] = ok ? s[r] : -INFINITY;
  }
}
template <int HD, bool SPLIT>
__global__ __launch_bounds__(256) void k_hollow_attention_mfma(const HollowAttnArgs a) {
  constexpr int KS = HD / 16;                       // k-steps of the score product
  constexpr int KLD = HD + 8, VLD = 32 + 4;         // LDS row lengths (bf16 elements): K rows 16-byte padded; V^T rows of 72 bytes, so that
                                                    // the 8-byte fragment reads of 32 lanes (a row each) fall on 32 distinct bank pairs (80-byte
                                                    // rows: lanes 16 apart collided -- 54 % of this kernel's LDS cycles were conflicts)
  constexpr int NT = SPLIT ? 2 : 1;                 // terms per operand: hi (+ lo)
  __shared__ __attribute__((aligned(16))) unsigned short Ksm[NT][32 * KLD];     // [key][dim]
  __shared__ __attribute__((aligned(16))) unsigned short Vsm[NT][32 * VLD];     // [dim][key] (dims >= HD: zero rows)
  const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, kh = lane >> 5;
  // causal tiles: the LAST query tile has the longest key range -- dispatch the heavy tiles first (a launch that ends on its
  // longest workgroups idles most of the chip through its tail: 111 vs 92 us for the mirrored anti-causal layer)
  const int qtile = a.mode == 0 ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int q0 = qtile * 128 + wave * 32;           // first query of this wave
  const int i = q0 + col;                           // this lane's query
  const bool qok = i < a.Tq;
  // Q^T fragments (B operand of the score product): 8 consecutive dims of the lane's query per k-step, pre-scaled
  bf16x8 qf[NT][KS];
#pragma unroll
  for (int s_ = 0; s_ < KS; ++s_) {
    float qv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qok) {
      const float* qr = a.q + (size_t)b * a.q_bs + (size_t)i * a.q_rs + h * HD + 16 * s_ + 8 * kh;
      const float4 u0 = *(const float4*)qr, u1 = *(const float4*)(qr + 4);
      qv[0] = u0.x * a.scale; qv[1] = u0.y * a.scale; qv[2] = u0.z * a.scale; qv[3] = u0.w * a.scale;
      qv[4] = u1.x * a.scale; qv[5] = u1.y * a.scale; qv[6] = u1.z * a.scale; qv[7] = u1.w * a.scale;
    }
    unsigned w[4], wl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w[j] = hk_pack2(qv[2 * j], qv[2 * j + 1]);
      wl[j] = hk_pack2(qv[2 * j] - __uint_as_float(w[j] << 16), qv[2 * j + 1] - __uint_as_float(w[j] & 0xFFFF0000u));
    }
    qf[0][s_] = __builtin_bit_cast(bf16x8, u32x4h{w[0], w[1], w[2], w[3]});
    if (SPLIT) qf[NT - 1][s_] = __builtin_bit_cast(bf16x8, u32x4h{wl[0], wl[1], wl[2], wl[3]});
  }
  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.0f;
  float m = -INFINITY, l = 0.0f;
  for (int idx = threadIdx.x; idx < NT * 32 * VLD / 2; idx += 256) ((unsigned*)&Vsm[0][0])[idx] = 0u;   // (rows >= HD stay zero)
  const int wlo = qtile * 128, whi = min(wlo + 128, a.Tq) - 1;                 // query range of the workgroup
  const int mylo = q0, myhi = min(q0 + 32, a.Tq) - 1;                          // ... of this wave
  // chunk c (32 keys from jc) concerns the query range [lo, hi] iff some (query, key) pair is allowed
  auto range_any = [&](int jc, int lo, int hi) {
    const int jl = min(jc + 32, a.Tk) - 1;
    if (hi < lo) return false;
    if (a.mode == 0) return jc <= hi;
    if (a.mode == 1) return jl >= lo;
    return jc == 0 || (jc <= a.Tq && jc - 1 <= hi) || (jl > a.Tq && jl - a.Tq - 1 >= lo);
  };
  auto next_visible = [&](int jc) {                                            // (uniform over the workgroup)
    while (jc < a.Tk && !range_any(jc, wlo, whi)) jc += 32;
    return jc;
  };
  // one (K, V) float4 pair per thread and chunk, fetched a chunk ahead: the loads fly under the previous chunk's products
  const bool stager = threadIdx.x < 32 * HD / 4;
  const int sjj = threadIdx.x / (HD / 4), sc4 = (threadIdx.x % (HD / 4)) * 4;
  float4 pk = make_float4(0.f, 0.f, 0.f, 0.f), pv = pk;
  auto fetch = [&](int jc) {
    const int j = jc + sjj;
    pk = make_float4(0.f, 0.f, 0.f, 0.f);
    pv = pk;
    if (stager && j < a.Tk) {
      pk = *(const float4*)(a.k + (size_t)b * a.k_bs + (size_t)j * a.k_rs + h * HD + sc4);
      pv = *(const float4*)(a.v + (size_t)b * a.v_bs + (size_t)j * a.v_rs + h * HD + sc4);
    }
  };
  int jnext = next_visible(0);
  if (jnext < a.Tk) fetch(jnext);
  while (jnext < a.Tk) {
    const int j0 = jnext;
    __syncthreads();
    // stage the chunk: K rows as they are, V transposed, fp32 -> bf16 (hi, and the remainder lo in the split mode)
    if (stager) {
      const int jj = sjj, c4 = sc4;
      const float4 kv = pk, vv = pv;
      const unsigned k01 = hk_pack2(kv.x, kv.y), k23 = hk_pack2(kv.z, kv.w);
      *(uint2*)(&Ksm[0][jj * KLD + c4]) = make_uint2(k01, k23);
      const float ve[4] = {vv.x, vv.y, vv.z, vv.w};
      unsigned short vh[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { vh[e] = hk_bf16(ve[e]); Vsm[0][(c4 + e) * VLD + jj] = vh[e]; }
      if (SPLIT) {
        *(uint2*)(&Ksm[NT - 1][jj * KLD + c4]) =
            make_uint2(hk_pack2(kv.x - __uint_as_float(k01 << 16), kv.y - __uint_as_float(k01 & 0xFFFF0000u)),
                       hk_pack2(kv.z - __uint_as_float(k23 << 16), kv.w - __uint_as_float(k23 & 0xFFFF0000u)));
#pragma unroll
        for (int e = 0; e < 4; ++e) Vsm[NT - 1][(c4 + e) * VLD + jj] = hk_lo(ve[e], vh[e]);
      }
    }
    __syncthreads();
    jnext = next_visible(j0 + 32);
    if (jnext < a.Tk) fetch(jnext);
    if (!range_any(j0, mylo, myhi)) continue;                                  // (wave-uniform; no barrier below)
    // ---- S^T = K Q^T : A = K rows (lane = key col, 8 dims per half), B = Q^T fragments
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      const bf16x8 kf = *(const bf16x8*)(&Ksm[0][col * KLD + 16 * s_ + 8 * kh]);
      if (SPLIT) {                                                              // small terms first
        const bf16x8 kl = *(const bf16x8*)(&Ksm[NT - 1][col * KLD + 16 * s_ + 8 * kh]);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qf[0][s_], sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[NT - 1][s_], sacc, 0, 0, 0);
      }
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0][s_], sacc, 0, 0, 0);
    }
    // ---- mask + online softmax for the lane's query over its 16 keys (+ the partner lane's 16)
    float mx = -INFINITY;
    // chunks every query of the wave sees in full need no per-element mask (most of a causal triangle); wave-uniform
    bool full = q0 + 31 < a.Tq && j0 + 31 < a.Tk;
    if (a.mode == 0) full = full && j0 + 31 <= q0;
    else if (a.mode == 1) full = full && j0 >= q0 + 31;
    else full = full && ((j0 >= 1 && j0 + 31 <= a.Tq && j0 + 30 <= q0) || (j0 > a.Tq && j0 - a.Tq - 1 >= q0 + 31));
    if (full) {
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
    } else {
      // (branch-free, the launch-uniform mode hoisted: a per-register `if` chain on it compiles to scalar lane-mask and / or /
      //  branch sequences an order of magnitude longer than the softmax arithmetic)
      if (a.mode == 0) hk_mask_tile<0>(sacc, a.Tq, a.Tk, i, qok, j0 + 4 * kh);
      else if (a.mode == 1) hk_mask_tile<1>(sacc, a.Tq, a.Tk, i, qok, j0 + 4 * kh);
      else hk_mask_tile<2>(sacc, a.Tq, a.Tk, i, qok, j0 + 4 * kh);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
    const float mn = fmaxf(m, mx);
    const float msafe = mn == -INFINITY ? 0.0f : mn;                           // a query that sees nothing in this chunk
    const float corr = __expf(m - msafe);                                      // exp(-inf) = 0 on the first visible chunk
    float rs = 0.0f;
    unsigned pw[8], pl[8];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float p0 = __expf(sacc[r] - msafe), p1 = __expf(sacc[r + 1] - msafe);
      rs += p0 + p1;
      pw[r >> 1] = hk_pack2(p0, p1);
      if (SPLIT) pl[r >> 1] = hk_pack2(p0 - __uint_as_float(pw[r >> 1] << 16), p1 - __uint_as_float(pw[r >> 1] & 0xFFFF0000u));
    }
    rs += __shfl_xor(rs, 32, WAVE);
    l = l * corr + rs;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] *= corr;
    // ---- O^T += V^T P^T : A = V^T rows (lane = dim col) in the key order of the probability registers
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const unsigned short* vr = &Vsm[0][col * VLD + 16 * s_ + 4 * kh];
      const uint2 va = *(const uint2*)vr, vb = *(const uint2*)(vr + 8);
      const bf16x8 vf = __builtin_bit_cast(bf16x8, u32x4h{va.x, va.y, vb.x, vb.y});
      const bf16x8 pf = __builtin_bit_cast(bf16x8, u32x4h{pw[4 * s_], pw[4 * s_ + 1], pw[4 * s_ + 2], pw[4 * s_ + 3]});
      if (SPLIT) {
        const unsigned short* vq = &Vsm[NT - 1][col * VLD + 16 * s_ + 4 * kh];
        const uint2 vc = *(const uint2*)vq, vd = *(const uint2*)(vq + 8);
        const bf16x8 vl = __builtin_bit_cast(bf16x8, u32x4h{vc.x, vc.y, vd.x, vd.y});
        const bf16x8 pq = __builtin_bit_cast(bf16x8, u32x4h{pl[4 * s_], pl[4 * s_ + 1], pl[4 * s_ + 2], pl[4 * s_ + 3]});
        oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, pf, oacc, 0, 0, 0);
        oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pq, oacc, 0, 0, 0);
      }
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc, 0, 0, 0);
    }
  }
  if (qok) {
    const float inv = 1.0f / l;
    const size_t oo = ((size_t)b * a.Tq + i) * a.out_rs + h * HD;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int d0 = 8 * r4 + 4 * kh;                                          // dims d0..d0+3 = registers 4 r4 .. 4 r4 + 3
      if (d0 < HD) {
        const float4 v = make_float4(oacc[4 * r4] * inv, oacc[4 * r4 + 1] * inv, oacc[4 * r4 + 2] * inv, oacc[4 * r4 + 3] * inv);
        if (a.out) *(float4*)(a.out + oo + d0) = v;
        if (a.out_hi) {
          const unsigned h01 = hk_pack2(v.x, v.y), h23 = hk_pack2(v.z, v.w);
          *(uint2*)(a.out_hi + oo + d0) = make_uint2(h01, h23);
          if (a.out_lo)
            *(uint2*)(a.out_lo + oo + d0) =
                make_uint2(hk_pack2(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xFFFF0000u)),
                           hk_pack2(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xFFFF0000u)));
        }
      }
    }
  }
}

}  // namespace ctdd
using namespace ctdd;

extern "C" int ctdd_hollow_embed(const void* args_, void* stream) {
  const HollowEmbedArgs& a = *(const HollowEmbedArgs*)args_;
  CTDD_REQUIRE((a.x64 || a.x32) && a.t && a.w_in && a.b_in && a.pe && a.l2r && a.r2l, CTDD_EINVAL, "hollow embed: null buffer");
  CTDD_REQUIRE(a.E % 2 == 0 && a.E >= 4 && a.S >= 2 && a.D >= 2, CTDD_EINVAL, "hollow embed: E=%d S=%d D=%d", a.E, a.S, a.D);
  int gx = (a.D * a.E + 2047) / 2048;
  gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
  hipLaunchKernelGGL(k_hollow_embed, dim3(gx, a.B), dim3(256), 0, (hipStream_t)stream, a);
  return finish_launch("k_hollow_embed");
}

extern "C" int ctdd_hollow_layernorm(const void* args_, void* stream) {
  const HollowLnArgs& a = *(const HollowLnArgs*)args_;
  CTDD_REQUIRE(a.x && a.gamma && a.beta && (a.out || a.out_hi) && a.B > 0 && a.T > 0 && a.E > 0, CTDD_EINVAL, "hollow layernorm: bad arguments");
  const int64_t rows = (int64_t)a.B * a.T;
  // vector path: rows of whole 16-byte pieces, LPR = E / (4 NV) lanes per row with LPR | 64, 16-byte aligned rows and tables
  const int NV = a.E > 256 ? 2 : 1, LPR = a.E / (4 * NV);
  auto al16 = [](const void* p_) { return ((uintptr_t)p_ & 15) == 0; };
  const bool vec = a.E % (4 * NV) == 0 && LPR >= 1 && LPR <= 64 && 64 % LPR == 0 && a.E <= 512 && a.x_bs % 4 == 0 && a.y_bs % 4 == 0 &&
                   a.out_bs % 4 == 0 && a.out_hi_bs % 4 == 0 && a.film_stride % 4 == 0 && al16(a.x) && al16(a.y) && al16(a.out) && al16(a.gamma) &&
                   al16(a.beta) && al16(a.film) && ((uintptr_t)a.out_hi & 7) == 0 && ((uintptr_t)a.out_lo & 7) == 0;
  if (vec) {
    const int RPP = 64 / LPR;
    if (NV == 1) {
      const int64_t per_wg = 4 * 4 * RPP;
      hipLaunchKernelGGL((k_hollow_layernorm_v4<1, 4>), dim3((unsigned)((rows + per_wg - 1) / per_wg)), dim3(256), 0, (hipStream_t)stream, a, LPR);
    } else {
      const int64_t per_wg = 4 * 2 * RPP;
      hipLaunchKernelGGL((k_hollow_layernorm_v4<2, 2>), dim3((unsigned)((rows + per_wg - 1) / per_wg)), dim3(256), 0, (hipStream_t)stream, a, LPR);
    }
    return finish_launch("k_hollow_layernorm_v4");
  }
  hipLaunchKernelGGL(k_hollow_layernorm, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return finish_launch("k_hollow_layernorm");
}

// ------------------------------------------------------------------ per-sample linears (time-embedding MLP, FiLM: B rows, not B*D)
// out[b][n] = act(bias[n] + sum_k x[b][k] w[n][k]), fp32, w in the module's own [N][K] layout.  One wave per output column:
// its weight row sits in registers (K / 64 values per lane), the B input rows stream past it (L1/L2-resident) and each dot
// product is one wave reduction.  The implicit-GEMM kernel spent 50-115 us on each of these few-MFLOP layers (a 128-row tile
// for 32 rows: launch latency and an empty matrix pipe); this takes a few microseconds.
template <int KPL>     // K / 64 values per lane
__global__ __launch_bounds__(256) void k_small_linear(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      int B, int K, int N, int act, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float wr[KPL];
#pragma unroll
  for (int i = 0; i < KPL; ++i) wr[i] = w[(size_t)n * K + i * 64 + lane];
  const float bv = bias ? bias[n] : 0.0f;
  for (int b = 0; b < B; ++b) {
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int i = 0; i < KPL; i += 2) {
      s0 = fmaf(x[(size_t)b * K + i * 64 + lane], wr[i], s0);
      if (i + 1 < KPL) s1 = fmaf(x[(size_t)b * K + (i + 1) * 64 + lane], wr[i + 1], s1);
    }
    float v = wave_sum(s0 + s1) + bv;
    if (act == 1) v = fmaxf(v, 0.0f);
    else if (act == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    if (lane == 0) out[(size_t)b * N + n] = v;
  }
}

extern "C" int ctdd_hollow_small_linear(const float* x, const float* w, const float* bias, int B, int K, int N, int act, float* out,
                                        void* stream) {
  CTDD_REQUIRE(x && w && out && B > 0 && N > 0 && K > 0 && K % 64 == 0 && K <= 1024, CTDD_EINVAL, "small linear: K=%d must be a multiple of 64, <= 1024", K);
  const dim3 g((N + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  switch (K / 64) {
#define SLCASE(k_) case k_: hipLaunchKernelGGL(k_small_linear<k_>, g, dim3(256), 0, st, x, w, bias, B, K, N, act, out); break;
    SLCASE(1) SLCASE(2) SLCASE(3) SLCASE(4) SLCASE(5) SLCASE(6) SLCASE(7) SLCASE(8) SLCASE(9) SLCASE(10) SLCASE(11) SLCASE(12) SLCASE(13)
    SLCASE(14) SLCASE(15) SLCASE(16)
#undef SLCASE
  }
  return finish_launch("k_small_linear");
}

extern "C" int ctdd_hollow_add(const float* p, int64_t p_bs, const float* q, int64_t q_bs, float* out, void* out_bf16, void* out_lo,
                               int64_t out_bs, int B, int64_t per_batch, void* stream) {
  CTDD_REQUIRE(p && q && (out || out_bf16) && B > 0 && per_batch > 0, CTDD_EINVAL, "hollow add: bad arguments");
  int gx = (int)((per_batch + 2047) / 2048);
  gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
  hipLaunchKernelGGL(k_hollow_add, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, p, p_bs, q, q_bs, out, (unsigned short*)out_bf16,
                     (unsigned short*)out_lo, out_bs, per_batch);
  return finish_launch("k_hollow_add");
}

extern "C" int ctdd_hollow_put_rows(const float* src, float* dst, void* dst_bf16, void* dst_lo, int64_t dst_bs, int B, int E, void* stream) {
  CTDD_REQUIRE(src && (dst || dst_bf16) && B > 0 && E > 0, CTDD_EINVAL, "hollow put_rows: bad arguments");
  hipLaunchKernelGGL(k_hollow_put_rows, dim3(B), dim3(256), 0, (hipStream_t)stream, src, dst, (unsigned short*)dst_bf16,
                     (unsigned short*)dst_lo, dst_bs, E);
  return finish_launch("k_hollow_put_rows");
}

extern "C" int ctdd_hollow_attention_bf16(const void* args_, void* stream) {
  const HollowAttnArgs& a = *(const HollowAttnArgs*)args_;
  CTDD_REQUIRE(a.q && a.k && a.v && (a.out || a.out_hi), CTDD_EINVAL, "hollow attention: null buffer");
  CTDD_REQUIRE(a.mode >= 0 && a.mode <= 2 && (a.mode != 2 || a.Tk == 2 * a.Tq + 1) && (a.mode == 2 || a.Tk == a.Tq), CTDD_EINVAL,
               "hollow attention: mode %d with Tq=%d Tk=%d", a.mode, a.Tq, a.Tk);
  CTDD_REQUIRE(a.q_rs % 4 == 0 && a.k_rs % 4 == 0 && a.v_rs % 4 == 0 && a.out_rs % 4 == 0 && a.q_bs % 4 == 0 && a.k_bs % 4 == 0 && a.v_bs % 4 == 0,
               CTDD_EINVAL, "hollow attention: strides must be multiples of 4 floats");
  const dim3 g((a.Tq + 127) / 128, a.H, a.B);
  hipStream_t st = (hipStream_t)stream;
  switch (a.hd) {
    case 16:
      if (a.split) hipLaunchKernelGGL((k_hollow_attention_mfma<16, true>), g, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((k_hollow_attention_mfma<16, false>), g, dim3(256), 0, st, a);
      break;
    case 32:
      if (a.split) hipLaunchKernelGGL((k_hollow_attention_mfma<32, true>), g, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((k_hollow_attention_mfma<32, false>), g, dim3(256), 0, st, a);
      break;
    default: CTDD_REQUIRE(false, CTDD_ERANGE, "bf16 hollow attention: head dim %d (16 or 32)", a.hd);
  }
  return finish_launch("k_hollow_attention_mfma");
}

extern "C" int ctdd_hollow_attention(const void* args_, void* stream) {
  const HollowAttnArgs& a = *(const HollowAttnArgs*)args_;
  CTDD_REQUIRE(a.q && a.k && a.v && (a.out || a.out_hi), CTDD_EINVAL, "hollow attention: null buffer");
  CTDD_REQUIRE(a.mode >= 0 && a.mode <= 2 && (a.mode != 2 || a.Tk == 2 * a.Tq + 1) && (a.mode == 2 || a.Tk == a.Tq), CTDD_EINVAL,
               "hollow attention: mode %d with Tq=%d Tk=%d", a.mode, a.Tq, a.Tk);
  CTDD_REQUIRE(a.q_rs % 4 == 0 && a.k_rs % 4 == 0 && a.v_rs % 4 == 0 && a.out_rs % 4 == 0 && a.q_bs % 4 == 0 && a.k_bs % 4 == 0 && a.v_bs % 4 == 0,
               CTDD_EINVAL, "hollow attention: strides must be multiples of 4 floats");
  const dim3 g((a.Tq + AQ - 1) / AQ, a.H, a.B);
  hipStream_t st = (hipStream_t)stream;
  switch (a.hd) {
    case 4: hipLaunchKernelGGL(k_hollow_attention<4>, g, dim3(AQ), 0, st, a); break;
    case 8: hipLaunchKernelGGL(k_hollow_attention<8>, g, dim3(AQ), 0, st, a); break;
    case 16: hipLaunchKernelGGL(k_hollow_attention<16>, g, dim3(AQ), 0, st, a); break;
    case 32: hipLaunchKernelGGL(k_hollow_attention<32>, g, dim3(AQ), 0, st, a); break;
    case 64: hipLaunchKernelGGL(k_hollow_attention<64>, g, dim3(AQ), 0, st, a); break;
    default: CTDD_REQUIRE(false, CTDD_ERANGE, "hollow attention: head dim %d (4, 8, 16, 32 or 64)", a.hd);
  }
  return finish_launch("k_hollow_attention");
}
