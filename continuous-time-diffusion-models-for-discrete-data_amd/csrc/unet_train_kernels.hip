// unet_train_kernels.hip -- hand-written BACKWARD kernels of the tauLDR U-Net score network
// (reference: the autograd graph of lib/networks/unet.py:100-140, 152-200, 303-459 walked by
// `l.backward()` in lib/training/training.py:27).  Same NHWC layouts and the same two arithmetic
// modes as the forward kernels in unet_kernels.hip (bf16 operands / fp32 accumulate on
// v_mfma_f32_32x32x16_bf16; exact fp32 on v_mfma_f32_32x32x2_f32).
//
//   data gradients (dgrad)   are convolutions themselves: the forward kernels run them on the
//                            output gradient with tap-flipped, transposed weights (ctdd_unet_pack_weights
//                            writes that layout); the stride-2 Downsample's transpose is segment kind
//                            CTDD_SEG_3x3_S2T of the generic implicit-GEMM kernel.
//   weight gradients (wgrad) k_wgrad: dW[n][tap][c] = sum_p dY[p][n] X[p + off(tap)][c].  The contraction
//                            runs over PIXELS, which are the rows of both NHWC operands: the bf16 path reads
//                            its MFMA fragments with the transposing LDS load ds_read_b64_tr_b16, the fp32
//                            path's one-float-per-lane operands need no transpose.  Border handling costs
//                            no instruction in the loop: pixels are enumerated in a zero-padded geometry
//                            (one zero column each side of a row, one zero row between images), so a tap is a
//                            constant offset and out-of-image products meet a zero.
//   GroupNorm+Swish backward two passes over (dA, X): per-(sample, channel) sums, then dX (+= existing gradient).
//   mid-block attention, first conv, upsample / bias / per-sample-bias reductions, weight re-packing
//   and gradient un-packing tables.
#include "common.hpp"

namespace ctdd {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x2v = __attribute__((ext_vector_type(2))) __bf16;
using f32x2v = __attribute__((ext_vector_type(2))) float;

__device__ inline unsigned pack2_bf16(float a, float b) {
  f32x2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
__device__ inline float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ inline float bf_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }

// load / store 8 consecutive channels of an NHWC tensor kept as fp32 or bf16
__device__ inline void load8(const float* f, const unsigned short* h, size_t off, float (&v)[8]) {
  if (f) {
    const float4 a = *(const float4*)(f + off), b = *(const float4*)(f + off + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    const uint4 u = *(const uint4*)(h + off);
    v[0] = bf_lo(u.x); v[1] = bf_hi(u.x); v[2] = bf_lo(u.y); v[3] = bf_hi(u.y);
    v[4] = bf_lo(u.z); v[5] = bf_hi(u.z); v[6] = bf_lo(u.w); v[7] = bf_hi(u.w);
  }
}
__device__ inline void store8(float* f, unsigned short* h, size_t off, const float (&v)[8]) {
  if (f) {
    *(float4*)(f + off) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(f + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
  if (h) *(uint4*)(h + off) = make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
}

// ============================================================================ weight gradient
enum { WG_3x3 = 0, WG_1x1 = 1, WG_3x3_S2 = 2 };
struct WgradArgs {
  const void* x;     // [B][Hin][Win][C]   the convolution's input activations (bf16 | fp32)
  const void* dy;    // [B][H][W][ldy]     gradient of its output; channels >= N are zero padding
  float* gw;         // [N][Ktot] fp32, this segment's columns at koff + tap*C + c; accumulated with atomics
  int B, H, W, Hin, Win, N, ldy, C, Ktot, koff;
  int kind;          // WG_3x3 (stride 1, pad 1) | WG_1x1 | WG_3x3_S2 (stride 2, input padded right/bottom)
  int nlr;           // WG_3x3: extended rows per chunk;  other kinds: pixels per chunk (multiple of 16)
  int nwn;           // waves along n (1, 2 or 4); 4 / nwn along c
  int nchunks;       // chunks in all; grid.x workgroups share them
};

// LDS image of a [positions][32 * tiles channels] operand tile.  bf16: the 64-byte pieces (one 32-channel tile) of a
// row are XOR-swizzled by the row so that the four rows a transposing read touches fall in four different bank
// quarters (unet_train_kernels.hip header; derivation in DESIGN.md); fp32: plain rows.
template <bool F32>
__device__ inline int lds_off(int pos, int tile, int within_bytes, int tiles) {
  if (F32) return (pos * tiles + tile) * 128 + within_bytes;
  const int f = tiles == 4 ? (pos & 3) : tiles == 2 ? ((pos >> 1) & 1) : 0;
  return (pos * tiles + (tile ^ f)) * 64 + within_bytes;
}

// One MFMA operand fragment of 16 consecutive positions p0 .. p0+15 (the contraction index) x 32 channels of tile
// `tile`, with the position as k.  bf16: two ds_read_b64_tr_b16 (4 positions x 16 channels per 16-lane group each).
__device__ inline bf16x8 frag_tr(const unsigned char* base, int p0, int tile, int tiles, int lane) {
  const int grp = lane >> 4, i = lane & 15, h = grp >> 1, cb = 16 * (grp & 1);
  const int q = i >> 2, pp = i & 3;
  const int r0 = p0 + 8 * h + q;
  const unsigned char* a0 = base + lds_off<false>(r0, tile, (cb + 4 * pp) * 2, tiles);
  const unsigned char* a1 = base + lds_off<false>(r0 + 4, tile, (cb + 4 * pp) * 2, tiles);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <bool F32>
__global__ __launch_bounds__(256, 2) void k_wgrad(const WgradArgs a) {
  constexpr int ESZ = F32 ? 4 : 2, EPV = 16 / ESZ, TB = F32 ? 128 : 64;    // bytes of one 32-channel tile of a row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwn = a.nwn, nwc = 4 / nwn;
  const int ngr_n = (a.N + 32 * nwn - 1) / (32 * nwn);
  const int grp_n = blockIdx.y % ngr_n, grp_c = blockIdx.y / ngr_n;
  const int nt = wave % nwn, ct = wave / nwn;
  const int n_base = grp_n * 32 * nwn, c_base = grp_c * 32 * nwc;       // first channel of the staged operand rows
  const int n0 = n_base + 32 * nt, c0 = c_base + 32 * ct;
  const bool active = n0 < a.N && c0 < a.C;
  const bool three = a.kind == WG_3x3;
  const int Wp = a.W + 2;
  const int KP = three ? ((a.nlr * Wp + 15) & ~15) : a.nlr;              // contraction positions per chunk
  const int XP = three ? KP + 2 * Wp + 2 : KP;
  unsigned char* Ys = smem;                                              // [KP][32 nwn]
  unsigned char* Xs = smem + (size_t)KP * nwn * TB;                      // [XP][32 nwc]
  const int ntaps = three ? 9 : 1;
  const int tap_s2 = a.kind == WG_3x3_S2 ? (int)blockIdx.z : 0;          // stride-2: one tap per grid.z slice

  for (int i = tid * 16; i < (KP * nwn + XP * nwc) * TB; i += 256 * 16) *(uint4*)(smem + i) = make_uint4(0, 0, 0, 0);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  const int vn = 32 * nwn / EPV, vc = 32 * nwc / EPV;                    // 16-byte vectors per staged row
  const unsigned char* xg = (const unsigned char*)a.x;
  const unsigned char* yg = (const unsigned char*)a.dy;
  const int HW = a.H * a.W;
  const float inv_W = 1.0f / (float)a.W, inv_H1 = 1.0f / (float)(a.H + 1);

  for (int chunk = blockIdx.x; chunk < a.nchunks; chunk += gridDim.x) {
    __syncthreads();                                                     // the previous chunk's fragments are read
    if (three) {
      // ---- extended rows E0 .. E0 + nlr - 1 of the list (image b, row y) = (E / (H+1), E % (H+1)); y == H is the zero row
      const int E0 = chunk * a.nlr;
      for (int v = tid; v < a.nlr * a.W * vn; v += 256) {
        const int cv = v % vn, pi = v / vn;
        const int lr = (int)(((float)pi + 0.5f) * inv_W), xx = pi - lr * a.W;
        const int E = E0 + lr;
        const int b = (int)(((float)E + 0.5f) * inv_H1), y = E - b * (a.H + 1);
        const int ch = n_base + cv * EPV;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (y < a.H && b < a.B && ch < a.ldy) val = *(const uint4*)(yg + ((size_t)((size_t)b * HW + y * a.W + xx) * a.ldy + ch) * ESZ);
        const int pos = lr * Wp + xx + 1;
        *(uint4*)(Ys + lds_off<F32>(pos, cv * EPV / 32, (cv * EPV % 32) * ESZ, nwn)) = val;
      }
      for (int v = tid; v < (a.nlr + 2) * a.W * vc; v += 256) {
        const int cv = v % vc, pi = v / vc;
        const int lr = (int)(((float)pi + 0.5f) * inv_W), xx = pi - lr * a.W;
        const int E = E0 - 1 + lr;
        const int b = E < 0 ? 0 : (int)(((float)E + 0.5f) * inv_H1), y = E < 0 ? a.H : E - b * (a.H + 1);
        const int ch = c_base + cv * EPV;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (y < a.H && b < a.B && ch < a.C) val = *(const uint4*)(xg + ((size_t)((size_t)b * HW + y * a.W + xx) * a.C + ch) * ESZ);
        const int pos = lr * Wp + xx + 2;
        *(uint4*)(Xs + lds_off<F32>(pos, cv * EPV / 32, (cv * EPV % 32) * ESZ, nwc)) = val;
      }
    } else {
      // ---- output pixels P0 .. P0 + KP - 1 in flattened (b, oy, ox) order; X gathered at the tap's input pixel
      const int64_t P0 = (int64_t)chunk * KP, M = (int64_t)a.B * HW;
      const int dyy = tap_s2 / 3, dxx = tap_s2 % 3;
      for (int v = tid; v < KP * vn; v += 256) {
        const int cv = v % vn, pi = v / vn;
        const int64_t p = P0 + pi;
        const int ch = n_base + cv * EPV;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (p < M && ch < a.ldy) val = *(const uint4*)(yg + ((size_t)p * a.ldy + ch) * ESZ);
        *(uint4*)(Ys + lds_off<F32>(pi, cv * EPV / 32, (cv * EPV % 32) * ESZ, nwn)) = val;
      }
      for (int v = tid; v < KP * vc; v += 256) {
        const int cv = v % vc, pi = v / vc;
        const int64_t p = P0 + pi;
        const int ch = c_base + cv * EPV;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (p < M && ch < a.C) {
          const int b = (int)(p / HW), r = (int)(p - (int64_t)b * HW), oy = r / a.W, ox = r - oy * a.W;
          int yy = oy, xx = ox;
          if (a.kind == WG_3x3_S2) { yy = 2 * oy + dyy; xx = 2 * ox + dxx; }
          if (yy < a.Hin && xx < a.Win) val = *(const uint4*)(xg + ((size_t)(((size_t)b * a.Hin + yy) * a.Win + xx) * a.C + ch) * ESZ);
        }
        *(uint4*)(Xs + lds_off<F32>(pi, cv * EPV / 32, (cv * EPV % 32) * ESZ, nwc)) = val;
      }
    }
    __syncthreads();
    if (!active) continue;
    if constexpr (F32) {
      const int li = lane & 31, kk = lane >> 5;
      for (int p0 = 0; p0 < KP; p0 += 2) {
        const float av = *(const float*)(Ys + lds_off<true>(p0 + kk, nt, li * 4, nwn));
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          if (t < ntaps) {
            const int off = three ? (t / 3) * Wp + (t % 3) : 0;
            const float bv = *(const float*)(Xs + lds_off<true>(p0 + kk + off, ct, li * 4, nwc));
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
          }
        }
      }
    } else {
      for (int p0 = 0; p0 < KP; p0 += 16) {
        const bf16x8 af = frag_tr(Ys, p0, nt, nwn, lane);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          if (t < ntaps) {
            const int off = three ? (t / 3) * Wp + (t % 3) : 0;
            const bf16x8 bfr = frag_tr(Xs, p0 + off, ct, nwc, lane);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[t], 0, 0, 0);
          }
        }
      }
    }
  }
  if (!active) return;
  // ---- flush: lane column c = c0 + (lane & 31), register r row n = n0 + (r&3) + 8 (r>>2) + 4 (lane>>5): one register of
  // the wave is two 128-byte row segments, the float-atomic unit that runs at the full rate
  const int col = c0 + (lane & 31), g = lane >> 5;
  if (col < a.C) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t < ntaps) {
        const int tap = a.kind == WG_3x3_S2 ? tap_s2 : t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (n < a.N) atomicAdd(a.gw + (size_t)n * a.Ktot + a.koff + tap * a.C + col, acc[t][r]);
        }
      }
    }
  }
}

// ============================================================================ GroupNorm (+Swish, +dropout) backward
// forward (unet.py:103-133): z = gamma xhat + beta, xhat = (x - mean_g) rstd_g ; a = swish(z) [* keep / (1 - p)]
// backward: dz = dA swish'(z) [* keep / (1 - p)]
//           dX = rstd_g ( gamma dz - m1_g - xhat m2_g ),  m1_g = mean_g(gamma dz), m2_g = mean_g(gamma dz xhat)
//           dgamma[c] = sum_{b,p} dz xhat ;  dbeta[c] = sum_{b,p} dz
struct GnBwdArgs {
  const float* s1_f32; const unsigned short* s1_bf16; const double* st1; int C1;     // the forward's inputs + their statistics
  const float* s2_f32; const unsigned short* s2_bf16; const double* st2; int C2;
  const float* gamma; const float* beta;
  int B, HW, G; float eps; int swish;
  const float* da_f32; const unsigned short* da_bf16;       // gradient w.r.t. the activated output [B*HW][C1+C2]
  float* sums;                                               // [B][C][2] fp32: sum dz, sum dz*xhat  (zeroed by the caller)
  float* d1_f32; unsigned short* d1_bf16; int acc1;          // gradient w.r.t. source 1 (acc: add to what is there)
  float* d2_f32; unsigned short* d2_bf16; int acc2;
  float drop_p; const uint64_t* rng; uint64_t layer;         // dropout after the activation (ResBlock, unet.py:113,132): rng = {seed, step}
};
__device__ inline void gn_bwd_channel_table(const GnBwdArgs& a, int b, float* mean, float* rstd) {
  const int C = a.C1 + a.C2, cg = C / a.G;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cg;
    double s = 0.0, q = 0.0;
    for (int j = g * cg; j < (g + 1) * cg; ++j) {
      const double* st = j < a.C1 ? a.st1 + ((size_t)b * a.C1 + j) * 2 : a.st2 + ((size_t)b * a.C2 + (j - a.C1)) * 2;
      s += st[0]; q += st[1];
    }
    const double n = (double)cg * (double)a.HW;
    const double m = s / n;
    const double var = fmax(q / n - m * m, 0.0);
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
}
// keep-mask of the 8 channels starting at element index e0 (multiple of 8): two Philox blocks
__device__ inline unsigned drop_keep8(uint64_t seed, uint64_t offset, uint64_t e0, float p) {
  const u4 r0 = philox_row(seed, offset, e0 >> 2, 0x44524F50u), r1 = philox_row(seed, offset, (e0 >> 2) + 1, 0x44524F50u);
  const uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
  unsigned m = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) m |= (u01(w[j]) >= p ? 1u : 0u) << j;
  return m;
}
// dz and xhat of one 8-channel vector
__device__ inline void gn_bwd_vec(const GnBwdArgs& a, int b, int px, int c0, const float* mean, const float* rstd, float (&dz)[8],
                                  float (&xh)[8]) {
  const int C = a.C1 + a.C2;
  const bool first = c0 < a.C1;
  const int cc = first ? c0 : c0 - a.C1, Cs = first ? a.C1 : a.C2;
  const size_t off = ((size_t)b * a.HW + px) * Cs + cc;
  float x[8], da[8];
  load8(first ? a.s1_f32 : a.s2_f32, first ? a.s1_bf16 : a.s2_bf16, off, x);
  const size_t oo = ((size_t)b * a.HW + px) * C + c0;
  load8(a.da_f32, a.da_bf16, oo, da);
  unsigned keep = 0xFFu;
  float inv_keep = 1.0f;
  if (a.drop_p > 0.0f) { keep = drop_keep8(a.rng[0], a.rng[1] * 4096u + a.layer, oo, a.drop_p); inv_keep = 1.0f / (1.0f - a.drop_p); }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    xh[j] = (x[j] - mean[c0 + j]) * rstd[c0 + j];
    float d = da[j];
    if (a.drop_p > 0.0f) d = (keep >> j) & 1u ? d * inv_keep : 0.0f;
    if (a.swish) {
      const float z = fmaf(a.gamma[c0 + j], xh[j], a.beta[c0 + j]);
      const float sg = 1.0f / (1.0f + expf(-z));
      d *= sg * (1.0f + z * (1.0f - sg));
    }
    dz[j] = d;
  }
}
// pass 1: per-(sample, channel) sums.  grid (pixel slices, B); thread = (pixel lane, 8-channel vector)
__global__ __launch_bounds__(256) void k_gn_bwd_reduce(const GnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = a.C1 + a.C2, b = blockIdx.y, vpp = C / 8;
  float* mean = sm; float* rstd = sm + C; float* part = sm + 2 * C;          // part: [2][C] block sums
  gn_bwd_channel_table(a, b, mean, rstd);
  for (int i = threadIdx.x; i < 2 * C; i += 256) part[i] = 0.0f;
  __syncthreads();
  const int per = (a.HW + gridDim.x - 1) / gridDim.x, p_lo = blockIdx.x * per, p_hi = min(p_lo + per, a.HW);
  // a thread keeps ONE channel vector and strides over the slice's pixels (256 / vpp pixel lanes; C <= 2048)
  const int lanes = 256 / vpp, cv = threadIdx.x % vpp, pl = threadIdx.x / vpp;
  if (pl < lanes) {
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.0f; s2[j] = 0.0f; }
    for (int px = p_lo + pl; px < p_hi; px += lanes) {
      float dz[8], xh[8];
      gn_bwd_vec(a, b, px, cv * 8, mean, rstd, dz, xh);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += dz[j]; s2[j] = fmaf(dz[j], xh[j], s2[j]); }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(part + cv * 8 + j, s1[j]);
      atomicAdd(part + C + cv * 8 + j, s2[j]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int c = i % C, which = i / C;
    atomicAdd(a.sums + ((size_t)b * C + c) * 2 + which, part[i]);
  }
}
// pass 2: dX.  grid (vector slices, B)
__global__ __launch_bounds__(256) void k_gn_bwd_apply(const GnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = a.C1 + a.C2, b = blockIdx.y, vpp = C / 8, cg = C / a.G;
  float* mean = sm; float* rstd = sm + C; float* m1 = sm + 2 * C; float* m2 = sm + 3 * C;
  gn_bwd_channel_table(a, b, mean, rstd);
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cg;
    float t1 = 0.0f, t2 = 0.0f;
    for (int j = g * cg; j < (g + 1) * cg; ++j) {
      t1 = fmaf(a.gamma[j], a.sums[((size_t)b * C + j) * 2], t1);
      t2 = fmaf(a.gamma[j], a.sums[((size_t)b * C + j) * 2 + 1], t2);
    }
    const float inv = 1.0f / ((float)cg * (float)a.HW);
    m1[c] = t1 * inv; m2[c] = t2 * inv;
  }
  __syncthreads();
  const int64_t nv = (int64_t)a.HW * vpp;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += (int64_t)gridDim.x * 256) {
    const int px = (int)(v / vpp), c0 = (int)(v % vpp) * 8;
    float dz[8], xh[8], dx[8];
    gn_bwd_vec(a, b, px, c0, mean, rstd, dz, xh);
#pragma unroll
    for (int j = 0; j < 8; ++j) dx[j] = rstd[c0 + j] * (a.gamma[c0 + j] * dz[j] - m1[c0 + j] - xh[j] * m2[c0 + j]);
    const bool first = c0 < a.C1;
    const int cc = first ? c0 : c0 - a.C1, Cs = first ? a.C1 : a.C2;
    const size_t off = ((size_t)b * a.HW + px) * Cs + cc;
    float* df = first ? a.d1_f32 : a.d2_f32;
    unsigned short* dh = first ? a.d1_bf16 : a.d2_bf16;
    if (first ? a.acc1 : a.acc2) {
      float old[8];
      load8(df, dh, off, old);
#pragma unroll
      for (int j = 0; j < 8; ++j) dx[j] += old[j];
    }
    store8(df, dh, off, dx);
  }
}

// forward-side dropout of an activated tensor in place (the mask backward regenerates): a *= keep / (1 - p)
__global__ __launch_bounds__(256) void k_dropout(float* f, unsigned short* h, int64_t nvec, float p, const uint64_t* rng, uint64_t layer) {
  const float inv_keep = 1.0f / (1.0f - p);
  const uint64_t seed = rng[0], offset = rng[1] * 4096u + layer;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    float x[8];
    load8(f, h, (size_t)v * 8, x);
    const unsigned keep = drop_keep8(seed, offset, (uint64_t)v * 8, p);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (keep >> j) & 1u ? x[j] * inv_keep : 0.0f;
    store8(f, h, (size_t)v * 8, x);
  }
}

// ============================================================================ small reductions / data movement
// per-(sample, channel) sums of an NHWC gradient: out_bn[b * stride + n] += sum_p dY[b,p,n] (per-sample time-bias
// gradient, unet.py:110,131) and out_n[n] += the same over all samples (bias gradients).  fp32 atomics.
__global__ __launch_bounds__(256) void k_colsum(const float* f, const unsigned short* h, int HW, int N, int ld, float* out_bn, int stride,
                                                float* out_n) {
  extern __shared__ __attribute__((aligned(16))) float sm[];                  // [N]
  const int b = blockIdx.y, vpp = N / 8;
  for (int i = threadIdx.x; i < N; i += 256) sm[i] = 0.0f;
  __syncthreads();
  const int per = (HW + gridDim.x - 1) / gridDim.x, p_lo = blockIdx.x * per, p_hi = min(p_lo + per, HW);
  const int lanes = 256 / vpp, cv = threadIdx.x % vpp, pl = threadIdx.x / vpp;          // N <= 2048
  if (pl < lanes) {
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.0f;
    for (int px = p_lo + pl; px < p_hi; px += lanes) {
      float v[8];
      load8(f, h, ((size_t)b * HW + px) * ld + cv * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(sm + cv * 8 + j, s[j]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += 256) {
    if (out_bn) atomicAdd(out_bn + (size_t)b * stride + i, sm[i]);
    if (out_n) atomicAdd(out_n + i, sm[i]);
  }
}

// out[j] (+)= sum_b in[b * bstride + j * jstride]  (GroupNorm dgamma / dbeta from the per-sample sums, ...)
__global__ __launch_bounds__(256) void k_sum_batch(const float* in, int B, int64_t bstride, int jstride, int n, float* out, int accumulate) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  float s = 0.0f;
  for (int b = 0; b < B; ++b) s += in[(size_t)b * bstride + (size_t)j * jstride];
  out[j] = accumulate ? out[j] + s : s;
}

// dst (+)= src over n elements (n % 8 == 0): the identity-skip / residual branch of a gradient
__global__ __launch_bounds__(256) void k_accumulate(const float* sf, const unsigned short* sh, float* df, unsigned short* dh, int64_t nvec,
                                                    int accumulate) {
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    float x[8];
    load8(sf, sh, (size_t)v * 8, x);
    if (accumulate) {
      float o[8];
      load8(df, dh, (size_t)v * 8, o);
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] += o[j];
    }
    store8(df, dh, (size_t)v * 8, x);
  }
}

// gradient of the nearest-2x upsampling (unet.py:79-85): d x[b,y,x,c] (+)= sum of the 2x2 block of d up
__global__ __launch_bounds__(256) void k_downsum2x(const float* uf, const unsigned short* uh, int B, int H, int W, int C, float* of,
                                                   unsigned short* oh, int accumulate) {
  const int vpp = C / 8;
  const int64_t total = (int64_t)B * H * W * vpp;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int cv = (int)(v % vpp);
    const int64_t p = v / vpp;
    const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)H * W));
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.0f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float t[8];
        load8(uf, uh, (((size_t)b * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx) * C + cv * 8, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += t[j];
      }
    const size_t o = (size_t)p * C + cv * 8;
    if (accumulate) {
      float old[8];
      load8(of, oh, o, old);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += old[j];
    }
    store8(of, oh, o, s);
  }
}
// fp32 nearest-2x upsampling (the bf16 one lives in unet_kernels.hip)
__global__ __launch_bounds__(256) void k_upsample2x_f32(const float* x, int B, int H, int W, int C, float* out) {
  const int vpp = C / 4;
  const int64_t total = (int64_t)B * 4 * H * W * vpp;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int cv = (int)(v % vpp);
    const int64_t p = v / vpp;
    const int xo = (int)(p % (2 * W)), yo = (int)((p / (2 * W)) % (2 * H)), b = (int)(p / ((int64_t)4 * H * W));
    *(float4*)(out + (size_t)p * C + cv * 4) = *(const float4*)(x + (((size_t)b * H + (yo >> 1)) * W + (xo >> 1)) * C + cv * 4);
  }
}
// fp32 -> bf16 (the loss hands over d logits in fp32), optionally from rows of `ld_in` to rows of `ld_out` with zero padding
__global__ __launch_bounds__(256) void k_cast_rows(const float* in, int64_t rows, int n, int ld_in, int ld_out, unsigned short* out_bf16,
                                                   float* out_f32) {
  const int vpr = ld_out / 8;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < rows * vpr; v += (int64_t)gridDim.x * 256) {
    const int64_t r = v / vpr;
    const int c0 = (int)(v % vpr) * 8;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = c0 + j < n ? in[(size_t)r * ld_in + c0 + j] : 0.0f;
    store8(out_f32, out_bf16, (size_t)r * ld_out + c0, x);
  }
}

// ============================================================================ mid-block attention backward (unet.py:176-200)
// one workgroup per (b, head): recompute w = softmax((q s)(k s)^T), then
//   dV = w^T dO ; dW = dO V^T ; dS = w (dW - rowsum(dW w)) ; dQ = s^2 dS K ; dK = s^2 dS^T Q      (s = ch^-1/4)
struct AttnBwdArgs { const float* qkv; const float* d_out_f32; const unsigned short* d_out_bf16; int B, T, C, heads; float* d_qkv; unsigned short* d_qkv_bf16; };
__global__ __launch_bounds__(256) void k_attn_small_bwd(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x / a.heads, hd = blockIdx.x % a.heads, ch = a.C / a.heads, T = a.T;
  float* q = sm; float* k = q + T * ch; float* v = k + T * ch; float* dO = v + T * ch;
  float* w = dO + T * ch; float* dS = w + T * T;                                       // [T][T] each
  const float sc = 1.0f / sqrtf(sqrtf((float)ch));
  for (int i = threadIdx.x; i < T * ch; i += 256) {
    const int t = i / ch, c = i % ch;
    const float* src = a.qkv + ((size_t)b * T + t) * 3 * a.C + hd * 3 * ch;
    q[i] = src[c] * sc; k[i] = src[ch + c] * sc; v[i] = src[2 * ch + c];
    const size_t oo = ((size_t)b * T + t) * a.C + hd * ch + c;
    dO[i] = a.d_out_f32 ? a.d_out_f32[oo] : bf_lo((unsigned)a.d_out_bf16[oo]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T * T; i += 256) {
    const int t = i / T, s = i % T;
    float d = 0.0f, e = 0.0f;
    for (int c = 0; c < ch; ++c) { d = fmaf(q[t * ch + c], k[s * ch + c], d); e = fmaf(dO[t * ch + c], v[s * ch + c], e); }
    w[i] = d; dS[i] = e;                                                               // scores ; dW
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 256) {
    float m = -INFINITY;
    for (int s = 0; s < T; ++s) m = fmaxf(m, w[t * T + s]);
    float z = 0.0f;
    for (int s = 0; s < T; ++s) { const float e = expf(w[t * T + s] - m); w[t * T + s] = e; z += e; }
    const float iz = 1.0f / z;
    float dot = 0.0f;
    for (int s = 0; s < T; ++s) { w[t * T + s] *= iz; dot = fmaf(dS[t * T + s], w[t * T + s], dot); }
    for (int s = 0; s < T; ++s) dS[t * T + s] = w[t * T + s] * (dS[t * T + s] - dot);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T * ch; i += 256) {
    const int t = i / ch, c = i % ch;
    float dq = 0.0f, dk = 0.0f, dv = 0.0f;
    for (int s = 0; s < T; ++s) {
      dq = fmaf(dS[t * T + s], k[s * ch + c], dq);          // dS[t][s] (k s)[s][c]
      dk = fmaf(dS[s * T + t], q[s * ch + c], dk);          // dS[s][t] (q s)[s][c]
      dv = fmaf(w[s * T + t], dO[s * ch + c], dv);
    }
    const size_t d0 = ((size_t)b * T + t) * 3 * a.C + hd * 3 * ch;
    if (a.d_qkv) { a.d_qkv[d0 + c] = dq * sc; a.d_qkv[d0 + ch + c] = dk * sc; a.d_qkv[d0 + 2 * ch + c] = dv; }
    if (a.d_qkv_bf16) {
      a.d_qkv_bf16[d0 + c] = (unsigned short)(pack2_bf16(dq * sc, 0.0f) & 0xFFFFu);
      a.d_qkv_bf16[d0 + ch + c] = (unsigned short)(pack2_bf16(dk * sc, 0.0f) & 0xFFFFu);
      a.d_qkv_bf16[d0 + 2 * ch + c] = (unsigned short)(pack2_bf16(dv, 0.0f) & 0xFFFFu);
    }
  }
}

// ============================================================================ first conv weight gradient (unet.py:343, C_in = 1..4)
// dW0[n][ci][tap] = sum_{b,p} dY[b,p,n] xc[b,ci,p + off(tap)] with xc the centred integer state; fp32 FMA, atomics
struct FirstWgradArgs { const int64_t* x64; const int32_t* x32; float lo, hi; const float* dy_f32; const unsigned short* dy_bf16;
                        int B, Cin, H, W, Cout; float* gw; float* gbias; };
__global__ __launch_bounds__(256) void k_first_conv_wgrad(const FirstWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];               // [Cout][Cin*9 + 1] block sums
  const int K = a.Cin * 9, HW = a.H * a.W, b = blockIdx.y;
  for (int i = threadIdx.x; i < a.Cout * (K + 1); i += 256) sm[i] = 0.0f;
  __syncthreads();
  const int per = (HW + gridDim.x - 1) / gridDim.x, p_lo = blockIdx.x * per, p_hi = min(p_lo + per, HW);
  // thread = (channel n, pixel lane); Cout <= 256
  const int lanes = max(256 / a.Cout, 1), n = threadIdx.x % a.Cout, pl = threadIdx.x / a.Cout;
  if (pl < lanes) {
    float acc[37];
#pragma unroll
    for (int i = 0; i < 37; ++i) acc[i] = 0.0f;
    for (int p = p_lo + pl; p < p_hi; p += lanes) {
      const size_t o = ((size_t)b * HW + p) * a.Cout + n;
      const float g = a.dy_f32 ? a.dy_f32[o] : bf_lo((unsigned)a.dy_bf16[o]);
      const int y = p / a.W, x = p % a.W;
      acc[36] += g;
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        if (ci < a.Cin) {
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            float xv = 0.0f;
            if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
              const size_t xo = (((size_t)b * a.Cin + ci) * a.H + yy) * a.W + xx;
              const float raw = a.x64 ? (float)a.x64[xo] : (float)a.x32[xo];
              xv = 2.0f * ((raw - a.lo) / (a.hi - a.lo)) - 1.0f;
            }
            acc[ci * 9 + t] = fmaf(g, xv, acc[ci * 9 + t]);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 36; ++i)
      if (i < K) atomicAdd(sm + n * (K + 1) + i, acc[i]);
    atomicAdd(sm + n * (K + 1) + K, acc[36]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.Cout * (K + 1); i += 256) {
    const int nn = i / (K + 1), kk = i % (K + 1);
    if (kk < K) atomicAdd(a.gw + (size_t)nn * K + kk, sm[i]);          // torch layout [Cout][Cin][3][3]
    else if (a.gbias) atomicAdd(a.gbias + nn, sm[i]);
  }
}

// ============================================================================ weight packing / gradient unpacking tables
// One launch converts every convolution weight of the network from the torch parameter [N][Cin_tot][k][k] (fp32 master)
// into the layouts the kernels stream, and one launch scatters every packed weight gradient back to torch layout.
struct PackEntry {
  const float* w;        // torch parameter [N][Cin_tot][k][k]
  void* fwd;             // [N][Ktot], this segment at koff + tap*C + c       (bf16 | fp32)  or null
  void* dgrad;           // [C][ntap*N]: row c, column tap'*N + n = w[n][c_off+c][ntap-1-tap'] (3x3 s1) | same tap (1x1, S2T)  or null
  float* gw;             // packed fp32 gradient [N][Ktot] (the k_wgrad target)  -> unpack source
  float* grad;           // torch-layout gradient [N][Cin_tot][k][k]              -> unpack target
  int N, Cin_tot, c_off, C, ntap, Ktot, koff, flip;
  int ldd, pad_;         // dgrad: columns per tap (N rounded up; the padding columns stay zero)
  int64_t first;         // first work item of this entry (items = N * C * ntap)
};
// entry of work item i, searched from the entry of the block's first item (one binary search per block)
__device__ inline int pack_entry_of(const PackEntry* tab, int nent, int64_t i, int64_t block_first) {
  __shared__ int e0;
  __syncthreads();
  if (threadIdx.x == 0) {
    int lo = 0, hi = nent - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].first <= block_first) lo = mid; else hi = mid - 1; }
    e0 = lo;
  }
  __syncthreads();
  int e = e0;
  while (e + 1 < nent && tab[e + 1].first <= i) ++e;
  return e;
}
__global__ __launch_bounds__(256) void k_pack_weights(const PackEntry* tab, int nent, int64_t total, int f32, uint64_t* rng_bump) {
  if (rng_bump && blockIdx.x == 0 && threadIdx.x == 0) rng_bump[1] += 1;        // one dropout stream per training forward
  for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < total; i0 += (int64_t)gridDim.x * 256) {
    const int64_t i = i0 + threadIdx.x;
    const int ei = pack_entry_of(tab, nent, i < total ? i : total - 1, i0);
    if (i >= total) continue;
    const PackEntry e = tab[ei];
    const int64_t r = i - e.first;
    // item order: n -> tap -> c (the forward layout is written contiguously)
    const int c = (int)(r % e.C), tap = (int)((r / e.C) % e.ntap), n = (int)(r / ((int64_t)e.C * e.ntap));
    const float v = e.w[((size_t)n * e.Cin_tot + e.c_off + c) * e.ntap + tap];
    const size_t fo = (size_t)n * e.Ktot + e.koff + (size_t)tap * e.C + c;
    const int tp = e.flip ? e.ntap - 1 - tap : tap;
    const size_t dg = (size_t)c * e.ntap * e.ldd + (size_t)tp * e.ldd + n;
    if (f32) {
      if (e.fwd) ((float*)e.fwd)[fo] = v;
      if (e.dgrad) ((float*)e.dgrad)[dg] = v;
    } else {
      const unsigned short hv = (unsigned short)(pack2_bf16(v, 0.0f) & 0xFFFFu);
      if (e.fwd) ((unsigned short*)e.fwd)[fo] = hv;
      if (e.dgrad) ((unsigned short*)e.dgrad)[dg] = hv;
    }
  }
}
__global__ __launch_bounds__(256) void k_unpack_grads(const PackEntry* tab, int nent, int64_t total) {
  for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < total; i0 += (int64_t)gridDim.x * 256) {
    const int64_t i = i0 + threadIdx.x;
    const int ei = pack_entry_of(tab, nent, i < total ? i : total - 1, i0);
    if (i >= total) continue;
    const PackEntry e = tab[ei];
    if (!e.gw || !e.grad) continue;
    const int64_t r = i - e.first;
    // item order: n -> c -> tap (the torch layout is written contiguously)
    const int tap = (int)(r % e.ntap), c = (int)((r / e.ntap) % e.C), n = (int)(r / ((int64_t)e.C * e.ntap));
    e.grad[((size_t)n * e.Cin_tot + e.c_off + c) * e.ntap + tap] = e.gw[(size_t)n * e.Ktot + e.koff + (size_t)tap * e.C + c];
  }
}

}  // namespace ctdd
using namespace ctdd;

// ============================================================================ C ABI (include/ctdd_unet_train.h)
static inline int grid_for(int64_t items, int cap = 4096) {
  int64_t g = (items + 255) / 256;
  return (int)(g < 1 ? 1 : g > cap ? cap : g);
}

extern "C" int ctdd_unet_wgrad(const void* args_, int f32, int grid_x, void* stream) {
  const WgradArgs& a = *(const WgradArgs*)args_;
  CTDD_REQUIRE(a.x && a.dy && a.gw, CTDD_EINVAL, "wgrad: null operand");
  CTDD_REQUIRE(a.nwn == 1 || a.nwn == 2 || a.nwn == 4, CTDD_EINVAL, "wgrad: nwn=%d", a.nwn);
  const int epv = f32 ? 4 : 8;
  CTDD_REQUIRE(a.C % epv == 0 && a.ldy % epv == 0 && a.N <= a.ldy, CTDD_EINVAL, "wgrad: C=%d ldy=%d N=%d need %d-element vectors", a.C, a.ldy, a.N, epv);
  CTDD_REQUIRE(a.kind == WG_3x3 || a.kind == WG_1x1 || a.kind == WG_3x3_S2, CTDD_EINVAL, "wgrad: kind=%d", a.kind);
  CTDD_REQUIRE(a.nlr > 0 && a.nchunks > 0 && grid_x > 0, CTDD_EINVAL, "wgrad: nlr=%d nchunks=%d grid=%d", a.nlr, a.nchunks, grid_x);
  if (a.kind == WG_3x3) CTDD_REQUIRE(a.Hin == a.H && a.Win == a.W, CTDD_EINVAL, "wgrad: 3x3 kind is stride 1");
  else CTDD_REQUIRE(a.nlr % 16 == 0, CTDD_EINVAL, "wgrad: pixels per chunk must be a multiple of 16");
  const int tb = f32 ? 128 : 64, nwc = 4 / a.nwn, Wp = a.W + 2;
  const int KP = a.kind == WG_3x3 ? ((a.nlr * Wp + 15) & ~15) : a.nlr;
  const int XP = a.kind == WG_3x3 ? KP + 2 * Wp + 2 : KP;
  const size_t lds = ((size_t)KP * a.nwn + (size_t)XP * nwc) * tb;
  CTDD_REQUIRE(lds <= 160 * 1024, CTDD_ERANGE, "wgrad: %zu bytes of LDS", lds);
  const int ngr_n = (a.N + 32 * a.nwn - 1) / (32 * a.nwn), ngr_c = (a.C + 32 * nwc - 1) / (32 * nwc);
  dim3 g((unsigned)grid_x, (unsigned)(ngr_n * ngr_c), a.kind == WG_3x3_S2 ? 9u : 1u);
  hipStream_t st = (hipStream_t)stream;
  if (f32) {
    static bool done[16] = {};
    ensure_lds_ceiling((const void*)k_wgrad<true>, done);
    hipLaunchKernelGGL(k_wgrad<true>, g, dim3(256), lds, st, a);
  } else {
    static bool done[16] = {};
    ensure_lds_ceiling((const void*)k_wgrad<false>, done);
    hipLaunchKernelGGL(k_wgrad<false>, g, dim3(256), lds, st, a);
  }
  return finish_launch("k_wgrad");
}

extern "C" int ctdd_unet_gn_bwd(const void* args_, void* stream) {
  const GnBwdArgs& a = *(const GnBwdArgs*)args_;
  const int C = a.C1 + a.C2;
  CTDD_REQUIRE(C % 8 == 0 && a.C1 % 8 == 0 && C % a.G == 0 && C <= 2048 && a.sums, CTDD_EINVAL, "gn bwd: C=%d C1=%d G=%d", C, a.C1, a.G);
  CTDD_REQUIRE((a.s1_f32 || a.s1_bf16) && (a.da_f32 || a.da_bf16) && (a.d1_f32 || a.d1_bf16), CTDD_EINVAL, "gn bwd: null tensor");
  CTDD_REQUIRE(a.drop_p >= 0.0f && a.drop_p < 1.0f && (a.drop_p == 0.0f || a.rng), CTDD_EINVAL, "gn bwd: dropout p=%g", (double)a.drop_p);
  hipStream_t st = (hipStream_t)stream;
  const int gx = max(1, min(a.HW / 16, 256 * 4 / max(a.B, 1) + 1));
  hipLaunchKernelGGL(k_gn_bwd_reduce, dim3(gx, a.B), dim3(256), (size_t)4 * C * sizeof(float), st, a);
  if (int rc = finish_launch("k_gn_bwd_reduce")) return rc;
  const int64_t nv = (int64_t)a.HW * (C / 8);
  const int ga = (int)max((int64_t)1, min((nv + 255) / 256, (int64_t)(2048 / max(a.B, 1) + 1)));
  hipLaunchKernelGGL(k_gn_bwd_apply, dim3(ga, a.B), dim3(256), (size_t)4 * C * sizeof(float), st, a);
  return finish_launch("k_gn_bwd_apply");
}

extern "C" int ctdd_unet_dropout(float* f32, void* bf16, int64_t n, float p, const uint64_t* rng, uint64_t layer, void* stream) {
  CTDD_REQUIRE((f32 || bf16) && rng && n % 8 == 0 && p > 0.0f && p < 1.0f, CTDD_EINVAL, "dropout: n=%lld p=%g", (long long)n, (double)p);
  hipLaunchKernelGGL(k_dropout, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, f32, (unsigned short*)bf16, n / 8, p, rng, layer);
  return finish_launch("k_dropout");
}

extern "C" int ctdd_unet_colsum(const float* f32, const void* bf16, int B, int HW, int N, int ld, float* out_bn, int stride, float* out_n,
                                void* stream) {
  CTDD_REQUIRE((f32 || bf16) && N % 8 == 0 && ld % 8 == 0 && N <= ld && N <= 2048 && (out_bn || out_n), CTDD_EINVAL, "colsum: N=%d ld=%d", N, ld);
  const int gx = max(1, min(HW / 16, 1024 / max(B, 1) + 1));
  hipLaunchKernelGGL(k_colsum, dim3(gx, B), dim3(256), (size_t)N * sizeof(float), (hipStream_t)stream, f32, (const unsigned short*)bf16, HW, N,
                     ld, out_bn, stride, out_n);
  return finish_launch("k_colsum");
}

extern "C" int ctdd_unet_sum_batch(const float* in, int B, int64_t bstride, int jstride, int n, float* out, int accumulate, void* stream) {
  CTDD_REQUIRE(in && out && n > 0 && B > 0, CTDD_EINVAL, "sum_batch: null / empty");
  hipLaunchKernelGGL(k_sum_batch, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, B, bstride, jstride, n, out, accumulate);
  return finish_launch("k_sum_batch");
}

extern "C" int ctdd_unet_accumulate(const float* src_f32, const void* src_bf16, float* dst_f32, void* dst_bf16, int64_t n, int accumulate,
                                    void* stream) {
  CTDD_REQUIRE((src_f32 || src_bf16) && (dst_f32 || dst_bf16) && n % 8 == 0, CTDD_EINVAL, "accumulate: n=%lld", (long long)n);
  hipLaunchKernelGGL(k_accumulate, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, src_f32, (const unsigned short*)src_bf16, dst_f32,
                     (unsigned short*)dst_bf16, n / 8, accumulate);
  return finish_launch("k_accumulate");
}

extern "C" int ctdd_unet_downsum2x(const float* up_f32, const void* up_bf16, int B, int H, int W, int C, float* out_f32, void* out_bf16,
                                   int accumulate, void* stream) {
  CTDD_REQUIRE((up_f32 || up_bf16) && (out_f32 || out_bf16) && C % 8 == 0, CTDD_EINVAL, "downsum2x: C=%d", C);
  hipLaunchKernelGGL(k_downsum2x, dim3(grid_for((int64_t)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, up_f32,
                     (const unsigned short*)up_bf16, B, H, W, C, out_f32, (unsigned short*)out_bf16, accumulate);
  return finish_launch("k_downsum2x");
}

extern "C" int ctdd_unet_upsample2x_f32(const float* x, int B, int H, int W, int C, float* out, void* stream) {
  CTDD_REQUIRE(x && out && C % 4 == 0, CTDD_EINVAL, "upsample2x_f32: C=%d", C);
  hipLaunchKernelGGL(k_upsample2x_f32, dim3(grid_for((int64_t)B * 4 * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, B, H, W, C, out);
  return finish_launch("k_upsample2x_f32");
}

extern "C" int ctdd_unet_cast_rows(const float* in, int64_t rows, int n, int ld_in, int ld_out, void* out_bf16, float* out_f32, void* stream) {
  CTDD_REQUIRE(in && (out_bf16 || out_f32) && ld_out % 8 == 0 && n <= ld_out && n <= ld_in, CTDD_EINVAL, "cast_rows: n=%d ld=%d/%d", n, ld_in, ld_out);
  hipLaunchKernelGGL(k_cast_rows, dim3(grid_for(rows * (ld_out / 8))), dim3(256), 0, (hipStream_t)stream, in, rows, n, ld_in, ld_out,
                     (unsigned short*)out_bf16, out_f32);
  return finish_launch("k_cast_rows");
}

extern "C" int ctdd_unet_attention_bwd(const void* args_, void* stream) {
  const AttnBwdArgs& a = *(const AttnBwdArgs*)args_;
  CTDD_REQUIRE(a.qkv && (a.d_out_f32 || a.d_out_bf16) && (a.d_qkv || a.d_qkv_bf16) && a.C % a.heads == 0, CTDD_EINVAL, "attention bwd: bad arguments");
  const int ch = a.C / a.heads;
  const size_t lds = (size_t)(4 * a.T * ch + 2 * a.T * a.T) * sizeof(float);
  CTDD_REQUIRE(lds <= 160 * 1024, CTDD_ERANGE, "attention bwd tile too large (T=%d)", a.T);
  static bool done[16] = {};
  ensure_lds_ceiling((const void*)k_attn_small_bwd, done);
  hipLaunchKernelGGL(k_attn_small_bwd, dim3(a.B * a.heads), dim3(256), lds, (hipStream_t)stream, a);
  return finish_launch("k_attn_small_bwd");
}

extern "C" int ctdd_unet_first_conv_wgrad(const void* args_, void* stream) {
  const FirstWgradArgs& a = *(const FirstWgradArgs*)args_;
  CTDD_REQUIRE((a.x64 || a.x32) && (a.dy_f32 || a.dy_bf16) && a.gw, CTDD_EINVAL, "first conv wgrad: null operand");
  CTDD_REQUIRE(a.Cin >= 1 && a.Cin <= 4 && a.Cout <= 256, CTDD_ERANGE, "first conv wgrad: Cin=%d Cout=%d", a.Cin, a.Cout);
  const int HW = a.H * a.W;
  const int gx = max(1, min(HW / 32, 512 / max(a.B, 1) + 1));
  hipLaunchKernelGGL(k_first_conv_wgrad, dim3(gx, a.B), dim3(256), (size_t)a.Cout * (a.Cin * 9 + 1) * sizeof(float), (hipStream_t)stream, a);
  return finish_launch("k_first_conv_wgrad");
}

extern "C" int ctdd_unet_pack_weights(const void* table_dev, int nent, int64_t total, int f32, uint64_t* rng_bump, void* stream) {
  CTDD_REQUIRE(table_dev && nent > 0 && total > 0, CTDD_EINVAL, "pack_weights: empty table");
  hipLaunchKernelGGL(k_pack_weights, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, (const PackEntry*)table_dev, nent, total, f32,
                     rng_bump);
  return finish_launch("k_pack_weights");
}
extern "C" int ctdd_unet_unpack_grads(const void* table_dev, int nent, int64_t total, void* stream) {
  CTDD_REQUIRE(table_dev && nent > 0 && total > 0, CTDD_EINVAL, "unpack_grads: empty table");
  hipLaunchKernelGGL(k_unpack_grads, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, (const PackEntry*)table_dev, nent, total);
  return finish_launch("k_unpack_grads");
}
