// hollow_kernels.hip -- inference kernels of the SDDM hollow (bidirectional-causal) transformer
// (reference lib/networks/hollow_networks.py: BidirectionalTransformer2 668-755, UniDirectionalTransformer
// 497-568, SelfAttentionBlock 311-340, CrossAttention 204-280, AttentionReadout 283-308,
// ResidualReadout 90-132, PositionalEncoding 1136-1156).  fp32 throughout (the 1e-4 logit parity bar).
// The linear layers run on the generic fp32-MFMA implicit-GEMM kernel of unet_kernels.hip (a token is a
// "pixel", the weight a 1x1 segment; ReLU / GELU in its epilogue); this file holds what is not a GEMM:
//   k_hollow_embed     state -> the two shifted token sequences [temb | x_0..x_{D-2}] and [x_1..x_{D-1} | temb]
//   k_hollow_layernorm LayerNorm of (x [+ y]) with optional per-sample FiLM, strided destination
//   k_hollow_add       a + b over strided batches (l2r + r2l)
//   k_hollow_attention masked multi-head attention with an online softmax: causal (j <= i), anti-causal
//                      (j >= i) and the readout mask [temb | l2r j <= i | r2l j >= i]
#include "common.hpp"

namespace ctdd {

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
};
__global__ __launch_bounds__(256) void k_hollow_layernorm(const HollowLnArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)a.B * a.T) return;
  const int b = (int)(row / a.T), j = (int)(row % a.T), E = a.E;
  const float* x = a.x + (size_t)b * a.x_bs + (size_t)j * E;
  const float* y = a.y ? a.y + (size_t)b * a.y_bs + (size_t)j * E : nullptr;
  float s = 0.0f;
  for (int e = lane; e < E; e += 64) s += x[e] + (y ? y[e] : 0.0f);
  const float mean = wave_sum(s) / (float)E;
  float q = 0.0f;
  for (int e = lane; e < E; e += 64) { const float d = x[e] + (y ? y[e] : 0.0f) - mean; q = fmaf(d, d, q); }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + a.eps);
  float* o = a.out + (size_t)b * a.out_bs + (size_t)j * E;
  for (int e = lane; e < E; e += 64) {
    float v = (x[e] + (y ? y[e] : 0.0f) - mean) * rstd * a.gamma[e] + a.beta[e];
    if (a.film) v = a.film[(size_t)b * a.film_stride + e] * v + a.film[(size_t)b * a.film_stride + E + e];
    o[e] = v;
  }
}

// out[b][j][:] = p[b][j][:] + q[b][j][:], strided batches
__global__ __launch_bounds__(256) void k_hollow_add(const float* __restrict__ p, int64_t p_bs, const float* __restrict__ q, int64_t q_bs,
                                                   float* __restrict__ out, int64_t out_bs, int64_t per_batch) {
  const int b = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_batch; i += (int64_t)gridDim.x * 256)
    out[(size_t)b * out_bs + i] = p[(size_t)b * p_bs + i] + q[(size_t)b * q_bs + i];
}
// rows of a (B, E) matrix into slot 0 of a (B, T, E) buffer
__global__ __launch_bounds__(256) void k_hollow_put_rows(const float* __restrict__ src, float* __restrict__ dst, int64_t dst_bs, int E) {
  const int b = blockIdx.x;
  for (int e = threadIdx.x; e < E; e += 256) dst[(size_t)b * dst_bs + e] = src[(size_t)b * E + e];
}

// ------------------------------------------------------------------ masked attention, online softmax
// q rows (b, i): q + b*q_bs + i*q_rs + h*hd ; k / v likewise; out (b, i): out + (b*Tq + i)*E_out + h*hd.
// mode 0: key j allowed iff j <= i ; 1: j >= i ; 2 (readout, Tk = 2 Tq + 1): j == 0 | 1 <= j <= Tq: j-1 <= i | j > Tq: j-Tq-1 >= i.
// One workgroup = 64 queries of one (b, head); a query is served by 4 lanes that split the head dimension
// (hd <= 64, multiple of 4); keys come through LDS in chunks of 64.
struct HollowAttnArgs {
  const float* q; const float* k; const float* v;
  int64_t q_bs, k_bs, v_bs; int q_rs, k_rs, v_rs;
  int B, Tq, Tk, H, hd, mode; float scale;
  float* out; int out_rs;
};
constexpr int AQ = 64, AK = 64;
__global__ __launch_bounds__(256) void k_hollow_attention(const HollowAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // K chunk [AK][hd] | V chunk [AK][hd]
  const int hd = a.hd, b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * AQ;
  float* Ks = sm;
  float* Vs = sm + AK * hd;
  const int qi = threadIdx.x >> 2, part = threadIdx.x & 3, per = hd / 4;
  const int i = i0 + qi;
  const bool qok = i < a.Tq;
  float qv[16], acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) { qv[c] = 0.0f; acc[c] = 0.0f; }
  if (qok)
    for (int c = 0; c < per; ++c) qv[c] = a.q[(size_t)b * a.q_bs + (size_t)i * a.q_rs + h * hd + part * per + c] * a.scale;
  float m = -INFINITY, l = 0.0f;
  // key range that any query of this block may see
  const int ilo = i0, ihi = min(i0 + AQ, a.Tq) - 1;
  for (int j0 = 0; j0 < a.Tk; j0 += AK) {
    const int j1 = min(j0 + AK, a.Tk) - 1;
    bool any;
    if (a.mode == 0) any = j0 <= ihi;
    else if (a.mode == 1) any = j1 >= ilo;
    else any = j0 == 0 || (j0 <= a.Tq && j0 - 1 <= ihi) || (j1 > a.Tq && j1 - a.Tq - 1 >= ilo);
    if (!any) continue;                              // (uniform over the workgroup)
    __syncthreads();
    for (int idx = threadIdx.x; idx < AK * hd; idx += 256) {
      const int jj = idx / hd, c = idx % hd, j = j0 + jj;
      Ks[idx] = j < a.Tk ? a.k[(size_t)b * a.k_bs + (size_t)j * a.k_rs + h * hd + c] : 0.0f;
      Vs[idx] = j < a.Tk ? a.v[(size_t)b * a.v_bs + (size_t)j * a.v_rs + h * hd + c] : 0.0f;
    }
    __syncthreads();
    const int nj = j1 - j0 + 1;
    for (int jj = 0; jj < nj; ++jj) {
      const int j = j0 + jj;
      bool ok;
      if (a.mode == 0) ok = j <= i;
      else if (a.mode == 1) ok = j >= i;
      else ok = j == 0 || (j <= a.Tq ? j - 1 <= i : j - a.Tq - 1 >= i);
      float s = 0.0f;
      const float* kr = Ks + jj * hd + part * per;
      for (int c = 0; c < per; ++c) s = fmaf(qv[c], kr[c], s);
      s += __shfl_xor(s, 1, WAVE);
      s += __shfl_xor(s, 2, WAVE);
      if (ok && qok) {
        const float mn = fmaxf(m, s);
        const float corr = expf(m - mn), p = expf(s - mn);
        l = l * corr + p;
        const float* vr = Vs + jj * hd + part * per;
        for (int c = 0; c < per; ++c) acc[c] = fmaf(p, vr[c], acc[c] * corr);
        m = mn;
      }
    }
  }
  if (qok) {
    const float inv = 1.0f / l;
    for (int c = 0; c < per; ++c) a.out[((size_t)b * a.Tq + i) * a.out_rs + h * hd + part * per + c] = acc[c] * inv;
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
  CTDD_REQUIRE(a.x && a.gamma && a.beta && a.out && a.B > 0 && a.T > 0 && a.E > 0, CTDD_EINVAL, "hollow layernorm: bad arguments");
  const int64_t rows = (int64_t)a.B * a.T;
  hipLaunchKernelGGL(k_hollow_layernorm, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return finish_launch("k_hollow_layernorm");
}

extern "C" int ctdd_hollow_add(const float* p, int64_t p_bs, const float* q, int64_t q_bs, float* out, int64_t out_bs, int B,
                               int64_t per_batch, void* stream) {
  CTDD_REQUIRE(p && q && out && B > 0 && per_batch > 0, CTDD_EINVAL, "hollow add: bad arguments");
  int gx = (int)((per_batch + 2047) / 2048);
  gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
  hipLaunchKernelGGL(k_hollow_add, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, p, p_bs, q, q_bs, out, out_bs, per_batch);
  return finish_launch("k_hollow_add");
}

extern "C" int ctdd_hollow_put_rows(const float* src, float* dst, int64_t dst_bs, int B, int E, void* stream) {
  CTDD_REQUIRE(src && dst && B > 0 && E > 0, CTDD_EINVAL, "hollow put_rows: bad arguments");
  hipLaunchKernelGGL(k_hollow_put_rows, dim3(B), dim3(256), 0, (hipStream_t)stream, src, dst, dst_bs, E);
  return finish_launch("k_hollow_put_rows");
}

extern "C" int ctdd_hollow_attention(const void* args_, void* stream) {
  const HollowAttnArgs& a = *(const HollowAttnArgs*)args_;
  CTDD_REQUIRE(a.q && a.k && a.v && a.out, CTDD_EINVAL, "hollow attention: null buffer");
  CTDD_REQUIRE(a.hd % 4 == 0 && a.hd >= 4 && a.hd <= 64, CTDD_ERANGE, "hollow attention: head dim %d (multiple of 4, <= 64)", a.hd);
  CTDD_REQUIRE(a.mode >= 0 && a.mode <= 2 && (a.mode != 2 || a.Tk == 2 * a.Tq + 1) && (a.mode == 2 || a.Tk == a.Tq), CTDD_EINVAL,
               "hollow attention: mode %d with Tq=%d Tk=%d", a.mode, a.Tq, a.Tk);
  const size_t lds = (size_t)2 * AK * a.hd * sizeof(float);
  hipLaunchKernelGGL(k_hollow_attention, dim3((a.Tq + AQ - 1) / AQ, a.H, a.B), dim3(256), lds, (hipStream_t)stream, a);
  return finish_launch("k_hollow_attention");
}
