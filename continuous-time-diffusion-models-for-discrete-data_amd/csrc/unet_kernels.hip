// unet_kernels.hip -- hand-written inference kernels for the tauLDR U-Net score network
// (reference lib/networks/unet.py:303-459): NHWC implicit-GEMM 3x3 / 1x1 convolutions on the
// bf16 matrix cores, GroupNorm+Swish application, the first and the attention-block kernels.
//
// Tensors are NHWC.  Two arithmetic modes share every kernel:
//   bf16   bf16 activations and weights, v_mfma_f32_32x32x16_bf16, fp32 accumulate (the BASELINE
//          config's dtype; the headline throughput path);
//   fp32   fp32 activations and weights on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32
//          (bit-for-bit a k-ordered fmaf chain) -- the path that meets the 1e-4 logit parity bar.
//          (A 16-bit-mantissa bf16 hi+lo split was measured first: ~4e-3 relative on the logits
//          after 47 convolutions, not enough.)
//
// One implicit-GEMM kernel serves every convolution: M = B*H*W output pixels, N = C_out,
// K = concatenation of "segments" (source tensor x tap set): 3x3 stride 1 (pad 1), 1x1 (the
// ResBlock's linear skip folded into the same GEMM as extra K), 3x3 stride 2 with the reference's
// (0,1,0,1) padding (Downsample), 3x3 on the nearest-2x upsampled grid (Upsample, never
// materialised).  Channel concatenation of the up path is two segments, also never materialised.
// Epilogue: + bias[n] + time-projection[b][n] + residual, writes fp32 master and/or bf16 planes,
// optionally straight in the (B, D, S) logits layout, and accumulates per-(b, channel) sum /
// sum-of-squares for the next GroupNorm with one atomic per lane.
#include "common.hpp"

namespace ctdd {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x2v = __attribute__((ext_vector_type(2))) __bf16;
using f32x2v = __attribute__((ext_vector_type(2))) float;

__device__ inline unsigned pack2_bf16(float a, float b) {
  f32x2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
__device__ inline unsigned short to_bf16(float a) { return (unsigned short)(pack2_bf16(a, 0.0f) & 0xFFFFu); }
__device__ inline float from_bf16(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

enum SegKind { SEG_3x3 = 0, SEG_1x1 = 1, SEG_3x3_S2 = 2, SEG_3x3_UP = 3 };

struct ConvSeg {
  const unsigned short* hi;   // [B][Hin][Win][C] bf16   (bf16 mode)
  const float* f32;           // [B][Hin][Win][C] fp32   (fp32 mode)
  int C, kind;
};

struct ConvArgs {
  ConvSeg seg[3];
  int nseg;
  const unsigned short* w_hi; // [N][Ktot] bf16, K ordered seg -> tap -> channel
  const float* w_f32;         // [N][Ktot] fp32
  int B, H, W;                // output grid
  int Hin, Win;               // input grid of the 3x3 segments
  int N, Ktot;
  const float* bias;          // [N] or null
  const float* tbias;         // per-sample bias, row stride tb_stride, or null
  int tb_stride;
  const float* res_f32;       // residual [M][N] (fp32 master) or null
  const unsigned short* res_bf16;
  float* out_f32;             // [M][N] or null
  unsigned short* out_hi;     // [M][N] bf16 or null
  double* stats;              // [B][N][2] (sum, sumsq) in fp64 (no E[x^2]-mean^2 cancellation), atomics, or null
  int logits_C;               // > 0: out_f32 is (B, logits_C*H*W, N/logits_C): row = c*HW + p
  int ksplit;                 // > 1: grid.z workgroups each take a share of K and add into acc_buf
  float* acc_buf;             // [M][N] fp32, zeroed by the caller; finished by k_conv_finish
};

constexpr int BM = 128;

// ------------------------------------------------------------------ epilogue shared by the conv kernels
// One accumulator tile: lane column n, 16 registers = rows wrow0 + (r&3) + 8*(r>>2) + 4*g.
__device__ inline void conv_epilogue_tile(const ConvArgs& a, const f32x16& acc, int64_t wrow0, int n, int g,
                                          int64_t M, int HW) {
  const bool ncol = n < a.N;
  if (a.ksplit > 1) {                                     // partial sums only; k_conv_finish does the rest
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t p = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * g;
      if (p < M && ncol) atomicAdd(a.acc_buf + (size_t)p * a.N + n, acc[r]);
    }
    return;
  }
  const int b_first = (int)((wrow0 < M ? wrow0 : M - 1) / HW);
  const int64_t next_sample = (int64_t)(b_first + 1) * HW;   // a 32-row tile spans at most two samples (HW >= 32 or B small)
  const float bv = (a.bias && ncol) ? a.bias[n] : 0.0f;
  const float tb0 = (a.tbias && ncol) ? a.tbias[(size_t)b_first * a.tb_stride + n] : 0.0f;
  const float tb1 = (a.tbias && ncol && b_first + 1 < a.B) ? a.tbias[(size_t)(b_first + 1) * a.tb_stride + n] : 0.0f;
  const int S = a.logits_C > 0 ? a.N / a.logits_C : 1;
  const int lch = a.logits_C > 0 ? n / S : 0, ls = a.logits_C > 0 ? n % S : 0;
  double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;           // column sums for sample b_first / b_first+1
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t p = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * g;
    if (p < M && ncol) {
      const bool second = p >= next_sample;
      float v = acc[r] + bv + (second ? tb1 : tb0);
      const size_t o = (size_t)p * a.N + n;
      if (a.res_f32) v += a.res_f32[o];
      else if (a.res_bf16) v += from_bf16(a.res_bf16[o]);
      if (a.out_f32) {
        if (a.logits_C > 0) {
          const int b = b_first + (second ? 1 : 0);
          a.out_f32[(((size_t)b * a.logits_C + lch) * HW + (p - (int64_t)b * HW)) * S + ls] = v;
        } else {
          a.out_f32[o] = v;
        }
      }
      if (a.out_hi) a.out_hi[o] = to_bf16(v);
      if (second) { s1 += v; q1 += (double)v * v; } else { s0 += v; q0 += (double)v * v; }
    }
  }
  if (a.stats) {
    s0 += __shfl_xor(s0, 32, WAVE); q0 += __shfl_xor(q0, 32, WAVE);
    s1 += __shfl_xor(s1, 32, WAVE); q1 += __shfl_xor(q1, 32, WAVE);
    if (g == 0 && ncol) {
      double* st = a.stats + ((size_t)b_first * a.N + n) * 2;
      atomicAdd(st, s0);
      atomicAdd(st + 1, q0);
      if (b_first + 1 < a.B && (s1 != 0.0 || q1 != 0.0)) {
        atomicAdd(st + (size_t)a.N * 2, s1);
        atomicAdd(st + (size_t)a.N * 2 + 1, q1);
      }
    }
  }
}


// K-chunk walker: (segment, tap, channel offset), all wave-uniform
template <int BK>
struct ChunkIter {
  int seg, tap, c0, koff;
  __device__ void init() { seg = 0; tap = 0; c0 = 0; koff = 0; }
  __device__ void next(const ConvArgs& a) {
    koff += BK;
    c0 += BK;
    if (c0 >= a.seg[seg].C) {
      c0 = 0;
      ++tap;
      const int ntap = a.seg[seg].kind == SEG_1x1 ? 1 : 9;
      if (tap >= ntap) { tap = 0; ++seg; }
    }
  }
};

template <int BK, int BNT, bool F32>
__global__ __launch_bounds__(256) void k_conv_igemm(const ConvArgs a) {
  constexpr int BN = 32 * BNT;
  constexpr int ESZ = F32 ? 4 : 2;            // element bytes
  constexpr int EPV = 16 / ESZ;               // elements per 16-B vector
  constexpr int LDK = BK + EPV;               // LDS row length in elements (pad 16 B: conflict-free b128)
  constexpr int VPR = BK / EPV;               // 16-B vectors per row
  constexpr int AV = BM * VPR / 256;          // A vectors per thread
  constexpr int BV = (BN * VPR + 255) / 256;  // B vectors per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                                        // [BM][LDK]
  unsigned char* Bs = smem + (size_t)BM * LDK * ESZ;               // [BN][LDK]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HW = a.H * a.W;
  const int64_t M = (int64_t)a.B * HW;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // ---- per-thread A vector slots: output pixel -> (b, y, x)
  int pb[AV], py[AV], px[AV];
#pragma unroll
  for (int i = 0; i < AV; ++i) {
    const int v = tid + 256 * i;
    const int64_t p = m0 + v / VPR;
    if (p < M) {
      const int b = (int)(p / HW), r = (int)(p % HW);
      pb[i] = b; py[i] = r / a.W; px[i] = r % a.W;
    } else {
      pb[i] = -1; py[i] = 0; px[i] = 0;
    }
  }

  uint4 ra[AV], rb[BV];
  auto load_chunk = [&](const ChunkIter<BK>& it) {
    const ConvSeg sg = a.seg[it.seg];
    const int kind = sg.kind;
    const unsigned char* src = F32 ? (const unsigned char*)sg.f32 : (const unsigned char*)sg.hi;
    const unsigned char* wsrc = F32 ? (const unsigned char*)a.w_f32 : (const unsigned char*)a.w_hi;
    int dy = 0, dx = 0;
    if (kind != SEG_1x1) { dy = it.tap / 3; dx = it.tap % 3; }
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int v = tid + 256 * i;
      const int cv = v % VPR;
      int yy, xx;
      bool ok = pb[i] >= 0;
      if (kind == SEG_3x3) { yy = py[i] + dy - 1; xx = px[i] + dx - 1; }
      else if (kind == SEG_1x1) { yy = py[i]; xx = px[i]; }
      else if (kind == SEG_3x3_S2) { yy = 2 * py[i] + dy; xx = 2 * px[i] + dx; }       // pad right/bottom only
      else { yy = py[i] + dy - 1; xx = px[i] + dx - 1; }                               // on the upsampled grid
      if (kind == SEG_3x3_UP) {
        ok = ok && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        yy >>= 1; xx >>= 1;
      } else if (kind == SEG_3x3_S2 || kind == SEG_3x3) {
        ok = ok && yy >= 0 && yy < a.Hin && xx >= 0 && xx < a.Win;
      }
      const int hin = kind == SEG_1x1 ? a.H : a.Hin, win = kind == SEG_1x1 ? a.W : a.Win;
      const size_t off = (((size_t)pb[i] * hin + yy) * win + xx) * sg.C + it.c0 + cv * EPV;
      ra[i] = ok ? *(const uint4*)(src + off * ESZ) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const int v = tid + 256 * i;
      const int n = v / VPR, cv = v % VPR;
      const bool ok = v < BN * VPR && n0 + n < a.N;
      const size_t off = (size_t)(n0 + n) * a.Ktot + it.koff + cv * EPV;
      rb[i] = ok ? *(const uint4*)(wsrc + off * ESZ) : make_uint4(0, 0, 0, 0);
    }
  };
  // fp32 tiles are stored with even k in the first half of the row and odd k in the second, so
  // that the 32x32x2 operand of lane (i, h) -- k = 2s + h -- is four consecutive floats per ds_read_b128
  auto put = [&](unsigned char* base, int row, int cv, const uint4& v) {
    if (F32) {
      float* r = (float*)base + (size_t)row * LDK;
      *(uint2*)(r + 2 * cv) = make_uint2(v.x, v.z);
      *(uint2*)(r + BK / 2 + 2 * cv) = make_uint2(v.y, v.w);
    } else {
      *(uint4*)(base + ((size_t)row * LDK + cv * 8) * 2) = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int v = tid + 256 * i;
      put(As, v / VPR, v % VPR, ra[i]);
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const int v = tid + 256 * i;
      if (v < BN * VPR) put(Bs, v / VPR, v % VPR, rb[i]);
    }
  };

  f32x16 acc[BNT];
#pragma unroll
  for (int t = 0; t < BNT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

  const int nchunk = a.Ktot / BK;
  ChunkIter<BK> it;
  it.init();
  load_chunk(it);
  const int li = lane & 31, g = lane >> 5;
  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();                    // everyone finished reading the previous chunk from LDS
    store_chunk();
    __syncthreads();
    if (c + 1 < nchunk) {               // prefetch the next chunk into registers behind the MFMAs
      it.next(a);
      load_chunk(it);
    }
    if constexpr (F32) {
      const float* Aw = (const float*)As + (size_t)(wave * 32 + li) * LDK + g * (BK / 2);
      const float* Bw = (const float*)Bs + (size_t)li * LDK + g * (BK / 2);
#pragma unroll
      for (int s4 = 0; s4 < BK / 8; ++s4) {
        const float4 av = *(const float4*)(Aw + 4 * s4);
#pragma unroll
        for (int t = 0; t < BNT; ++t) {
          const float4 bv = *(const float4*)(Bw + (size_t)t * 32 * LDK + 4 * s4);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[t], 0, 0, 0);
        }
      }
    } else {
      const unsigned short* Aw = (const unsigned short*)As + (size_t)(wave * 32 + li) * LDK + g * 8;
      const unsigned short* Bw = (const unsigned short*)Bs + (size_t)li * LDK + g * 8;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        const bf16x8 ah = *(const bf16x8*)(Aw + ks * 16);
#pragma unroll
        for (int t = 0; t < BNT; ++t) {
          const bf16x8 bh = *(const bf16x8*)(Bw + (size_t)t * 32 * LDK + ks * 16);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue.  lane: column n = n0 + 32 t + li ; register r: row 32*wave + (r&3) + 8*(r>>2) + 4*g
#pragma unroll
  for (int t = 0; t < BNT; ++t) conv_epilogue_tile(a, acc[t], m0 + wave * 32, n0 + 32 * t + li, g, M, HW);
}

// ------------------------------------------------------------------ patch convolution (bf16, stride 1)
// The throughput kernel for 3x3 (pad 1) and 1x1 segments.  Output pixels are taken in flattened
// (b, y, x) order, so the inputs of a tile of BM consecutive pixels under ALL nine taps live in one
// contiguous slab of BM + 2(W+1) input pixels.  That slab (x BK channels) is staged in LDS ONCE
// per channel chunk and the nine taps read it at row offsets (dy-1)*W + (dx-1); image borders and
// sample boundaries are a per-lane predicate that zeroes the MFMA A fragment.  Compared with
// im2col-per-tap staging this cuts the LDS store traffic (the ~79 B/clk/CU ds_write path) and
// the L2 reads of the activations by ~6x; only the [BN][BK] weight tile is restaged per tap
// (double-buffered, one barrier per tap).  The next slab is prefetched into registers under the
// last taps.  4 waves, each WM rows x (32*BNT) columns; grid.z splits the channel chunks when M
// is too small to fill the chip (7x7 levels), partial sums meet in acc_buf.
template <int BK, int BNT, int WM>
__global__ __launch_bounds__(256, 2) void k_conv_patch(const ConvArgs a) {
  constexpr int BN = 32 * BNT, BMP = 4 * WM, MT = WM / 32;
  constexpr int LDK = BK + 8, VPR = BK / 8;
  constexpr int BV = (BN * VPR + 255) / 256;
  constexpr int PVMAX = ((BMP + 2 * 34) * VPR + 255) / 256;     // W <= 33
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, g = lane >> 5;
  const int HW = a.H * a.W, Wd = a.W;
  const int64_t M = (int64_t)a.B * HW;
  const int64_t p0 = (int64_t)blockIdx.x * BMP;
  const int n0 = blockIdx.y * BN;
  const int halo = Wd + 1;
  const int PR = BMP + 2 * halo;                                // slab rows
  unsigned short* Ap = (unsigned short*)smem;                   // [PR][LDK]
  unsigned short* Bs = Ap + (size_t)PR * LDK;                   // [2][BN][LDK]
  const int npv = PR * VPR;                                     // slab vectors

  // output pixel of this lane in every row tile: (y, x, in range)
  int oy[MT], ox[MT];
  bool oin[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int64_t p = p0 + wave * WM + mt * 32 + li;
    oin[mt] = p < M;
    const int r = (int)((oin[mt] ? p : 0) % HW);
    oy[mt] = r / Wd; ox[mt] = r % Wd;
  }

  // ---- channel-chunk units (segment, c0) owned by this z-slice
  int nunits = 0;
  for (int sgi = 0; sgi < a.nseg; ++sgi) nunits += a.seg[sgi].C / BK;
  const int zs = blockIdx.z, nz = a.ksplit > 1 ? a.ksplit : 1;

  uint4 rp[PVMAX], rb[BV];
  auto unit_info = [&](int u, int& sgi, int& c0, int& kbase) {   // kbase = K offset of (segment, tap 0, c0)
    int k = 0;
    sgi = 0;
    while (u >= a.seg[sgi].C / BK) {
      u -= a.seg[sgi].C / BK;
      k += a.seg[sgi].C * (a.seg[sgi].kind == SEG_1x1 ? 1 : 9);
      ++sgi;
    }
    c0 = u * BK;
    kbase = k + c0;
  };
  auto load_patch = [&](int sgi, int c0) {
    const ConvSeg sg = a.seg[sgi];
    const int hl = sg.kind == SEG_1x1 ? 0 : halo;
#pragma unroll
    for (int i = 0; i < PVMAX; ++i) {
      const int v = tid + 256 * i;
      const int64_t q = p0 - hl + v / VPR;
      const bool ok = v < npv && q >= 0 && q < M;
      rp[i] = ok ? *(const uint4*)(sg.hi + (size_t)q * sg.C + c0 + (v % VPR) * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < PVMAX; ++i) {
      const int v = tid + 256 * i;
      if (v < npv) *(uint4*)(Ap + (size_t)(v / VPR) * LDK + (v % VPR) * 8) = rp[i];
    }
  };
  auto load_b = [&](int koff) {
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const int v = tid + 256 * i;
      const int n = v / VPR;
      const bool ok = v < BN * VPR && n0 + n < a.N;
      rb[i] = ok ? *(const uint4*)(a.w_hi + (size_t)(n0 + n) * a.Ktot + koff + (v % VPR) * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const int v = tid + 256 * i;
      if (v < BN * VPR) *(uint4*)(Bs + ((size_t)buf * BN + v / VPR) * LDK + (v % VPR) * 8) = rb[i];
    }
  };

  f32x16 acc[MT][BNT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < BNT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][t][i] = 0.0f;

  int u = zs;
  int sgi, c0, kbase;
  if (u < nunits) {
    unit_info(u, sgi, c0, kbase);
    load_patch(sgi, c0);
    load_b(kbase);
  }
  int bbuf = 0;
  while (u < nunits) {
    const ConvSeg sg = a.seg[sgi];
    const bool one = sg.kind == SEG_1x1;
    const int ntap = one ? 1 : 9;
    const int hl = one ? 0 : halo;
    __syncthreads();                       // previous unit's slab and weight tiles are out of use
    store_patch();
    store_b(bbuf);
    __syncthreads();
    const int un = u + nz;                 // next unit of this slice
    int nsgi = 0, nc0 = 0, nkbase = 0;
    if (un < nunits) unit_info(un, nsgi, nc0, nkbase);
    for (int tap = 0; tap < ntap; ++tap) {
      // prefetch: next tap's weight tile, or (at the last tap) the next unit's first one + its slab
      if (tap + 1 < ntap) load_b(kbase + (tap + 1) * sg.C);
      else if (un < nunits) load_b(nkbase);
      if (tap == (ntap > 4 ? 4 : 0) && un < nunits) load_patch(nsgi, nc0);
      const int dy = one ? 1 : tap / 3, dx = one ? 1 : tap % 3;
      const int shift = (dy - 1) * Wd + (dx - 1);
      bool ok[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        ok[mt] = oin[mt] && (unsigned)(oy[mt] + dy - 1) < (unsigned)a.H && (unsigned)(ox[mt] + dx - 1) < (unsigned)Wd;
      const unsigned short* Aw = Ap + (size_t)(hl + wave * WM + li + shift) * LDK + g * 8;
      const unsigned short* Bw = Bs + ((size_t)bbuf * BN + li) * LDK + g * 8;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 af[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          uint4 raw = *(const uint4*)(Aw + (size_t)mt * 32 * LDK + ks * 16);
          if (!ok[mt]) raw = make_uint4(0, 0, 0, 0);
          af[mt] = __builtin_bit_cast(bf16x8, raw);
        }
#pragma unroll
        for (int t = 0; t < BNT; ++t) {
          const bf16x8 bf = *(const bf16x8*)(Bw + (size_t)t * 32 * LDK + ks * 16);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bf, acc[mt][t], 0, 0, 0);
        }
      }
      if (tap + 1 < ntap) {
        store_b(bbuf ^ 1);                 // the other buffer was last read before the previous barrier
        __syncthreads();
        bbuf ^= 1;
      }
    }
    u = un; sgi = nsgi; c0 = nc0; kbase = nkbase;
  }

#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < BNT; ++t)
      conv_epilogue_tile(a, acc[mt][t], p0 + wave * WM + mt * 32, n0 + 32 * t + li, g, M, HW);
}

// split-K finish: acc_buf (+ bias, time bias, residual) -> outputs and GroupNorm statistics
__global__ __launch_bounds__(256) void k_conv_finish(const ConvArgs a) {
  const int HW = a.H * a.W;
  const int64_t M = (int64_t)a.B * HW;
  const int b = blockIdx.y;
  // one workgroup per (sample, 32-column slab); threads stride over the sample's pixels
  const int n = blockIdx.x * 32 + (threadIdx.x & 31);
  if (n >= a.N) return;
  double s = 0.0, q = 0.0;
  const float bv = a.bias ? a.bias[n] : 0.0f;
  const float tb = a.tbias ? a.tbias[(size_t)b * a.tb_stride + n] : 0.0f;
  for (int px = threadIdx.x >> 5; px < HW; px += 8) {
    const int64_t p = (int64_t)b * HW + px;
    const size_t o = (size_t)p * a.N + n;
    float v = a.acc_buf[o] + bv + tb;
    if (a.res_f32) v += a.res_f32[o];
    else if (a.res_bf16) v += from_bf16(a.res_bf16[o]);
    if (a.out_f32) a.out_f32[o] = v;
    if (a.out_hi) a.out_hi[o] = to_bf16(v);
    s += v; q += (double)v * v;
  }
  (void)M;
  if (a.stats) {
    atomicAdd(a.stats + ((size_t)b * a.N + n) * 2, s);
    atomicAdd(a.stats + ((size_t)b * a.N + n) * 2 + 1, q);
  }
}

// nearest-neighbour 2x upsampling of an NHWC bf16 tensor (input of the Upsample conv, unet.py:79-85)
__global__ __launch_bounds__(256) void k_upsample2x(const unsigned short* __restrict__ x, int B, int H, int W, int C,
                                                    unsigned short* __restrict__ out) {
  const int vpp = C / 8;
  const int64_t total = (int64_t)B * 4 * H * W * vpp;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int cv = (int)(v % vpp);
    const int64_t p = v / vpp;
    const int xo = (int)(p % (2 * W)), yo = (int)((p / (2 * W)) % (2 * H)), b = (int)(p / ((int64_t)4 * H * W));
    *(uint4*)(out + (size_t)p * C + cv * 8) = *(const uint4*)(x + (((size_t)b * H + (yo >> 1)) * W + (xo >> 1)) * C + cv * 8);
  }
}

// ------------------------------------------------------------------ first conv (C_in = 1..4), fp32 direct
// x: (B, Cin, H, W) integer states as stored by the samplers (int64/int32) -> centre to [-1,1]
// (network_utils.center_data) -> conv3x3 pad 1 -> NHWC outputs.  Memory-bound; one thread per
// (pixel, 8 output channels).
struct FirstConvArgs {
  const int64_t* x64; const int32_t* x32;
  float lo, hi;               // x_min_max
  const float* w;             // [Cout][Cin][3][3] (torch layout)
  const float* bias;
  int B, Cin, H, W, Cout;
  float* out_f32; unsigned short* out_hi; double* stats;
  float* x0_f32;              // optional centred input (B,Cin,H,W) fp32 (logistic head needs it)
};
__global__ __launch_bounds__(256) void k_first_conv(const FirstConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sred[];     // [Cout][2] per-workgroup partial statistics
  const int cg = a.Cout / 8, HW = a.H * a.W, b = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * a.Cout; i += 256) sred[i] = 0.0;
  __syncthreads();
  // thread -> fixed channel group c8 and a strided set of this workgroup's pixels, so the
  // statistics accumulate in registers and reach LDS once per thread
  const int lanes = 256 / cg;                                         // pixel lanes per workgroup
  const int c8 = threadIdx.x % cg, pl = threadIdx.x / cg;
  const int per = (HW + gridDim.x - 1) / gridDim.x;
  const int p_lo = blockIdx.x * per, p_hi = min(p_lo + per, HW);
  float wreg[8][9 * 4];                                                // weights of this thread's 8 channels (Cin <= 4)
  if (pl < lanes)
    for (int j = 0; j < 8; ++j)
      for (int k = 0; k < 9 * a.Cin; ++k) wreg[j][k] = a.w[(size_t)(c8 * 8 + j) * a.Cin * 9 + k];
  double ssum[8], ssq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ssum[j] = 0.0; ssq[j] = 0.0; }
  if (pl < lanes)
    for (int r = p_lo + pl; r < p_hi; r += lanes) {
      const int y = r / a.W, x = r % a.W;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = a.bias[c8 * 8 + j];
      for (int ci = 0; ci < a.Cin; ++ci)
        for (int dy = 0; dy < 3; ++dy)
          for (int dx = 0; dx < 3; ++dx) {
            const int yy = y + dy - 1, xx = x + dx - 1;
            if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) continue;
            const size_t o = (((size_t)b * a.Cin + ci) * a.H + yy) * a.W + xx;
            const float raw = a.x64 ? (float)a.x64[o] : (float)a.x32[o];
            const float v = 2.0f * ((raw - a.lo) / (a.hi - a.lo)) - 1.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, wreg[j][(ci * 3 + dy) * 3 + dx], acc[j]);
          }
      if (a.x0_f32 && c8 == 0)
        for (int ci = 0; ci < a.Cin; ++ci) {
          const size_t o = (((size_t)b * a.Cin + ci) * a.H + y) * a.W + x;
          const float raw = a.x64 ? (float)a.x64[o] : (float)a.x32[o];
          a.x0_f32[o] = 2.0f * ((raw - a.lo) / (a.hi - a.lo)) - 1.0f;
        }
      const size_t o = ((size_t)b * HW + r) * a.Cout + c8 * 8;
      if (a.out_f32) {
        *(float4*)(a.out_f32 + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *(float4*)(a.out_f32 + o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
      }
      if (a.out_hi)
        *(uint4*)(a.out_hi + o) = make_uint4(pack2_bf16(acc[0], acc[1]), pack2_bf16(acc[2], acc[3]),
                                             pack2_bf16(acc[4], acc[5]), pack2_bf16(acc[6], acc[7]));
#pragma unroll
      for (int j = 0; j < 8; ++j) { ssum[j] += acc[j]; ssq[j] += (double)acc[j] * acc[j]; }
    }
  if (a.stats) {
    if (pl < lanes)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(&sred[(c8 * 8 + j) * 2], ssum[j]);
        atomicAdd(&sred[(c8 * 8 + j) * 2 + 1], ssq[j]);
      }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * a.Cout; i += 256) atomicAdd(a.stats + (size_t)b * a.Cout * 2 + i, sred[i]);
  }
}

// ------------------------------------------------------------------ GroupNorm (+Swish) application
// y = (x - mean_g) * rstd_g * gamma_c + beta_c [-> swish] on the channel-concatenation of up to two
// NHWC sources, group statistics folded from per-(b, channel) sums produced by the conv epilogues.
struct GnArgs {
  const float* s1_f32; const unsigned short* s1_bf16; const double* st1; int C1;
  const float* s2_f32; const unsigned short* s2_bf16; const double* st2; int C2;
  const float* gamma; const float* beta;
  int B, HW, G; float eps; int swish;
  unsigned short* out_hi; float* out_f32;
};
__global__ __launch_bounds__(256) void k_gn_apply(const GnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // scale[C], shift[C]
  const int C = a.C1 + a.C2, cg = C / a.G, b = blockIdx.y;
  float* scale = sm;
  float* shift = sm + C;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cg;
    double s = 0.0, q = 0.0;
    for (int j = g * cg; j < (g + 1) * cg; ++j) {
      const double* st = j < a.C1 ? a.st1 + ((size_t)b * a.C1 + j) * 2 : a.st2 + ((size_t)b * a.C2 + (j - a.C1)) * 2;
      s += st[0]; q += st[1];
    }
    const double n = (double)cg * (double)a.HW;
    const double mean = s / n;
    const double var = fmax(q / n - mean * mean, 0.0);
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    scale[c] = rstd * a.gamma[c];
    shift[c] = a.beta[c] - (float)mean * rstd * a.gamma[c];
  }
  __syncthreads();
  const int vpp = C / 8;                                   // 8-channel vectors per pixel
  const int64_t nv = (int64_t)a.HW * vpp;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += (int64_t)gridDim.x * 256) {
    const int px = (int)(v / vpp), c0 = (int)(v % vpp) * 8;
    float x[8];
    const bool first = c0 < a.C1;
    const int cc = first ? c0 : c0 - a.C1, Cs = first ? a.C1 : a.C2;
    const size_t off = ((size_t)b * a.HW + px) * Cs + cc;
    const float* f = first ? a.s1_f32 : a.s2_f32;
    const unsigned short* h = first ? a.s1_bf16 : a.s2_bf16;
    if (f) {
      const float4 u0 = *(const float4*)(f + off), u1 = *(const float4*)(f + off + 4);
      x[0] = u0.x; x[1] = u0.y; x[2] = u0.z; x[3] = u0.w; x[4] = u1.x; x[5] = u1.y; x[6] = u1.z; x[7] = u1.w;
    } else {
      const uint4 u = *(const uint4*)(h + off);
      x[0] = __uint_as_float(u.x << 16); x[1] = __uint_as_float(u.x & 0xFFFF0000u);
      x[2] = __uint_as_float(u.y << 16); x[3] = __uint_as_float(u.y & 0xFFFF0000u);
      x[4] = __uint_as_float(u.z << 16); x[5] = __uint_as_float(u.z & 0xFFFF0000u);
      x[6] = __uint_as_float(u.w << 16); x[7] = __uint_as_float(u.w & 0xFFFF0000u);
    }
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      y[j] = fmaf(x[j], scale[c0 + j], shift[c0 + j]);
      if (a.swish) y[j] = y[j] / (1.0f + expf(-y[j]));
    }
    const size_t oo = ((size_t)b * a.HW + px) * C + c0;
    if (a.out_hi)
      *(uint4*)(a.out_hi + oo) = make_uint4(pack2_bf16(y[0], y[1]), pack2_bf16(y[2], y[3]), pack2_bf16(y[4], y[5]),
                                            pack2_bf16(y[6], y[7]));
    if (a.out_f32) {
      *(float4*)(a.out_f32 + oo) = make_float4(y[0], y[1], y[2], y[3]);
      *(float4*)(a.out_f32 + oo + 4) = make_float4(y[4], y[5], y[6], y[7]);
    }
  }
}

// per-(b, channel) sum / sumsq of an NHWC tensor (for tensors no conv epilogue produced)
__global__ __launch_bounds__(256) void k_channel_stats(const float* __restrict__ x, int HW, int C, double* __restrict__ stats) {
  const int b = blockIdx.y;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
    double s = 0.0, q = 0.0;
    for (int p = 0; p < HW; ++p) {
      const double v = x[((size_t)b * HW + p) * C + c];
      s += v; q += v * v;
    }
    stats[((size_t)b * C + c) * 2] = s;
    stats[((size_t)b * C + c) * 2 + 1] = q;
  }
}

// ------------------------------------------------------------------ time embedding (unet.py:223-241, 332-337)
// temb = W2 swish(W1 [sin(t f) | cos(t f)] + b1) + b2 ; out = swish(temb)  (every ResBlock projects swish(temb))
struct TimeArgs {
  const float* t; int B, ch, tdim;
  const float* w1; const float* b1; const float* w2; const float* b2;   // torch Linear layouts [out][in]
  float* act;   // [B][tdim] = swish(temb)
};
__global__ __launch_bounds__(256) void k_time_mlp(const TimeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* e = sm;               // [ch]
  float* h = sm + a.ch;        // [tdim]
  const int b = blockIdx.x, half = a.ch / 2;
  const float t = a.t[b];
  for (int i = threadIdx.x; i < half; i += 256) {
    const float f = expf((float)i * (-logf(10000.0f) / (float)(half - 1)));
    e[i] = sinf(t * f);
    e[half + i] = cosf(t * f);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < a.tdim; o += 256) {
    float s = a.b1[o];
    for (int i = 0; i < a.ch; ++i) s = fmaf(a.w1[(size_t)o * a.ch + i], e[i], s);
    h[o] = s / (1.0f + expf(-s));
  }
  __syncthreads();
  for (int o = threadIdx.x; o < a.tdim; o += 256) {
    float s = a.b2[o];
    for (int i = 0; i < a.tdim; ++i) s = fmaf(a.w2[(size_t)o * a.tdim + i], h[i], s);
    a.act[(size_t)b * a.tdim + o] = s / (1.0f + expf(-s));
  }
}
// all ResBlocks' time projections in one launch: out[b][n] = sum_i Wt[i][n] act[b][i] + bias[n], n < Ntot
// (Wt = the concatenated Linear weights transposed to [tdim][Ntot]: coalesced over n)
__global__ __launch_bounds__(256) void k_time_proj(const float* __restrict__ act, const float* __restrict__ w,
                                                   const float* __restrict__ bias, int B, int tdim, int Ntot,
                                                   float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // act[b][:]
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < tdim; i += 256) sm[i] = act[(size_t)b * tdim + i];
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= Ntot) return;
  float s = bias[n];
  for (int i = 0; i < tdim; ++i) s = fmaf(w[(size_t)i * Ntot + n], sm[i], s);     // w is [tdim][Ntot]
  out[(size_t)b * Ntot + n] = s;
}

// ------------------------------------------------------------------ mid-block self-attention (unet.py:152-200)
// qkv: [B][T][3*C] fp32 with the reference's per-head channel order [q(ch) | k(ch) | v(ch)] per head;
// one workgroup per (b, head): w = softmax((q s)^T (k s)), s = ch^-1/4 ; out[b][t][head*ch + c].
struct AttnArgs { const float* qkv; int B, T, C, heads; unsigned short* out_hi; float* out_f32; };
__global__ __launch_bounds__(256) void k_attn_small(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x / a.heads, hd = blockIdx.x % a.heads, ch = a.C / a.heads, T = a.T;
  float* q = sm; float* k = q + T * ch; float* v = k + T * ch; float* w = v + T * ch;   // w: [T][T]
  const float sc = 1.0f / sqrtf(sqrtf((float)ch));
  for (int i = threadIdx.x; i < T * ch; i += 256) {
    const int t = i / ch, c = i % ch;
    const float* src = a.qkv + ((size_t)b * T + t) * 3 * a.C + hd * 3 * ch;
    q[i] = src[c] * sc; k[i] = src[ch + c] * sc; v[i] = src[2 * ch + c];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T * T; i += 256) {
    const int t = i / T, s = i % T;
    float d = 0.0f;
    for (int c = 0; c < ch; ++c) d = fmaf(q[t * ch + c], k[s * ch + c], d);
    w[i] = d;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 256) {
    float m = -INFINITY;
    for (int s = 0; s < T; ++s) m = fmaxf(m, w[t * T + s]);
    float z = 0.0f;
    for (int s = 0; s < T; ++s) { const float e = expf(w[t * T + s] - m); w[t * T + s] = e; z += e; }
    const float iz = 1.0f / z;
    for (int s = 0; s < T; ++s) w[t * T + s] *= iz;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T * ch; i += 256) {
    const int t = i / ch, c = i % ch;
    float o = 0.0f;
    for (int s = 0; s < T; ++s) o = fmaf(w[t * T + s], v[s * ch + c], o);
    const size_t oo = ((size_t)b * T + t) * a.C + hd * ch + c;
    if (a.out_hi) a.out_hi[oo] = to_bf16(o);
    if (a.out_f32) a.out_f32[oo] = o;
  }
}

// ------------------------------------------------------------------ logistic head (models.py:249-283)
// mu = tanh(loc + x0), logits[s] = log(sigmoid(r) - sigmoid(l)) via log_minus_exp, straight into (B,D,S)
struct LogisticArgs { const float* net; const float* x0; int B, C, HW, S, fix; float* out; };
__device__ inline float logsigmoidf(float x) { return fminf(x, 0.0f) - log1pf(expf(-fabsf(x))); }
__global__ __launch_bounds__(256) void k_logistic_head(const LogisticArgs a) {
  // net: NHWC [B][HW][2C] (loc channels 0..C-1, log_scale C..2C-1); x0: (B,C,HW) centred input
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)a.B * a.C * a.HW * a.S;
  if (idx >= total) return;
  const int s = (int)(idx % a.S);
  const int64_t d = idx / a.S;                      // b*C*HW + c*HW + p
  const int p = (int)(d % a.HW), c = (int)((d / a.HW) % a.C), b = (int)(d / ((int64_t)a.HW * a.C));
  const float* nr = a.net + ((size_t)b * a.HW + p) * 2 * a.C;
  const float mu = tanhf(nr[c] + a.x0[((size_t)b * a.C + c) * a.HW + p]);
  const float inv_scale = expf(-(nr[a.C + c] - 2.0f));
  const float bw = 2.0f / (float)a.S;
  const float centre = -1.0f + bw * 0.5f + (float)s * ((2.0f - bw) / (float)(a.S - 1));   // torch.linspace
  const float l = (centre - bw * 0.5f - mu) * inv_scale, r = (centre + bw * 0.5f - mu) * inv_scale;
  const float cl = logsigmoidf(l), cr = logsigmoidf(r);
  float v = cr + log1pf(-expf(cl - cr) + 1e-6f);
  if (a.fix) {
    const float a2 = -l + cl, b2 = -r + cr;
    v = fminf(v, a2 + log1pf(-expf(b2 - a2) + 1e-6f));
  }
  a.out[idx] = v;
}

}  // namespace ctdd
using namespace ctdd;

// ============================================================================ C ABI
template <int BK, int BNT, bool F32>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
  constexpr int ESZ = F32 ? 4 : 2;
  constexpr int LDK = BK + 16 / ESZ;
  const size_t lds = (size_t)(BM + 32 * BNT) * LDK * ESZ;
  const int64_t M = (int64_t)a.B * a.H * a.W;
  dim3 g((unsigned)((M + BM - 1) / BM), (unsigned)((a.N + 32 * BNT - 1) / (32 * BNT)));
  static bool attr_done = false;            // once per instantiation (not legal inside stream capture)
  if (lds > 48 * 1024 && !attr_done) {
    (void)hipFuncSetAttribute((const void*)k_conv_igemm<BK, BNT, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((k_conv_igemm<BK, BNT, F32>), g, dim3(256), lds, st, a);
  return finish_launch("k_conv_igemm");
}

extern "C" int ctdd_unet_conv(const void* args_, int bk, int bnt, int f32, void* stream) {
  const ConvArgs& a = *(const ConvArgs*)args_;
  CTDD_REQUIRE(a.nseg >= 1 && a.nseg <= 3, CTDD_EINVAL, "nseg=%d", a.nseg);
  CTDD_REQUIRE(a.Ktot % bk == 0, CTDD_EINVAL, "Ktot=%d not a multiple of BK=%d", a.Ktot, bk);
  for (int i = 0; i < a.nseg; ++i) {
    CTDD_REQUIRE(a.seg[i].C % bk == 0, CTDD_EINVAL, "segment %d: C=%d vs BK=%d", i, a.seg[i].C, bk);
    CTDD_REQUIRE(f32 ? a.seg[i].f32 != nullptr : a.seg[i].hi != nullptr, CTDD_EINVAL, "segment %d: null source", i);
  }
  CTDD_REQUIRE(f32 ? a.w_f32 != nullptr : a.w_hi != nullptr, CTDD_EINVAL, "null weights");
  hipStream_t st = (hipStream_t)stream;
#define CASE(BK_, BNT_) \
  if (bk == BK_ && bnt == BNT_) return f32 ? launch_conv<BK_, BNT_, true>(a, st) : launch_conv<BK_, BNT_, false>(a, st);
  CASE(96, 3) CASE(96, 4) CASE(96, 1) CASE(64, 4) CASE(64, 2) CASE(64, 1) CASE(32, 1) CASE(32, 3) CASE(32, 4)
  CASE(16, 1)
#undef CASE
  CTDD_REQUIRE(false, CTDD_ERANGE, "no conv instantiation for BK=%d BNT=%d", bk, bnt);
}

template <int BK, int BNT, int WM>
static int launch_patch(const ConvArgs& a, hipStream_t st) {
  constexpr int LDK = BK + 8;
  const int PR = 4 * WM + 2 * (a.W + 1);
  const size_t lds = ((size_t)PR + 2 * 32 * BNT) * LDK * 2;
  const int64_t M = (int64_t)a.B * a.H * a.W;
  dim3 g((unsigned)((M + 4 * WM - 1) / (4 * WM)), (unsigned)((a.N + 32 * BNT - 1) / (32 * BNT)), a.ksplit > 1 ? a.ksplit : 1);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)k_conv_patch<BK, BNT, WM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL((k_conv_patch<BK, BNT, WM>), g, dim3(256), lds, st, a);
  if (int rc = finish_launch("k_conv_patch")) return rc;
  if (a.ksplit > 1) {
    hipLaunchKernelGGL(k_conv_finish, dim3((a.N + 31) / 32, a.B), dim3(256), 0, st, a);
    return finish_launch("k_conv_finish");
  }
  return CTDD_OK;
}

extern "C" int ctdd_unet_conv_patch(const void* args_, int bk, int bnt, int wm, void* stream) {
  const ConvArgs& a = *(const ConvArgs*)args_;
  CTDD_REQUIRE(a.nseg >= 1 && a.nseg <= 3 && a.w_hi, CTDD_EINVAL, "bad conv arguments");
  CTDD_REQUIRE(a.W <= 33, CTDD_ERANGE, "W=%d > 33", a.W);
  CTDD_REQUIRE(a.Hin == a.H && a.Win == a.W, CTDD_EINVAL, "patch conv is stride 1");
  for (int i = 0; i < a.nseg; ++i) {
    CTDD_REQUIRE(a.seg[i].hi && a.seg[i].C % bk == 0, CTDD_EINVAL, "segment %d: C=%d vs BK=%d", i, a.seg[i].C, bk);
    CTDD_REQUIRE(a.seg[i].kind == SEG_3x3 || a.seg[i].kind == SEG_1x1, CTDD_EINVAL, "segment %d: kind %d", i, a.seg[i].kind);
  }
  CTDD_REQUIRE(a.ksplit <= 1 || (a.acc_buf && a.logits_C == 0), CTDD_EINVAL, "split-K needs acc_buf");
  hipStream_t st = (hipStream_t)stream;
#define CASEP(BK_, BNT_, WM_) if (bk == BK_ && bnt == BNT_ && wm == WM_) return launch_patch<BK_, BNT_, WM_>(a, st);
  CASEP(48, 3, 64) CASEP(48, 3, 32) CASEP(48, 4, 64) CASEP(48, 4, 32)
  CASEP(64, 4, 64) CASEP(64, 4, 32) CASEP(64, 2, 64) CASEP(64, 2, 32) CASEP(32, 1, 32) CASEP(32, 3, 32) CASEP(32, 4, 32)
  CASEP(16, 1, 32)
#undef CASEP
  CTDD_REQUIRE(false, CTDD_ERANGE, "no patch-conv instantiation for BK=%d BNT=%d WM=%d", bk, bnt, wm);
}

extern "C" int ctdd_unet_upsample2x(const void* x, int B, int H, int W, int C, void* out, void* stream) {
  CTDD_REQUIRE(x && out && C % 8 == 0, CTDD_EINVAL, "bad upsample arguments");
  const int64_t total = (int64_t)B * 4 * H * W * (C / 8);
  int gx = (int)((total + 255) / 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(k_upsample2x, dim3(gx), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, B, H, W, C,
                     (unsigned short*)out);
  return finish_launch("k_upsample2x");
}

extern "C" int ctdd_unet_first_conv(const void* args_, void* stream) {
  const FirstConvArgs& a = *(const FirstConvArgs*)args_;
  CTDD_REQUIRE((a.x64 || a.x32) && a.w && a.bias && a.Cout % 8 == 0, CTDD_EINVAL, "bad first-conv arguments");
  CTDD_REQUIRE(a.Cin <= 4 && a.Cout / 8 <= 256, CTDD_ERANGE, "first conv: Cin=%d Cout=%d", a.Cin, a.Cout);
  const int lanes = 256 / (a.Cout / 8);
  int gx = (a.H * a.W + 8 * lanes - 1) / (8 * lanes);     // ~8 pixels per thread
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(k_first_conv, dim3(gx, a.B), dim3(256), (size_t)2 * a.Cout * sizeof(double), (hipStream_t)stream, a);
  return finish_launch("k_first_conv");
}

extern "C" int ctdd_unet_gn_apply(const void* args_, void* stream) {
  const GnArgs& a = *(const GnArgs*)args_;
  const int C = a.C1 + a.C2;
  CTDD_REQUIRE(C % 8 == 0 && a.C1 % 8 == 0 && C % a.G == 0 && (a.out_hi || a.out_f32), CTDD_EINVAL, "bad GroupNorm arguments");
  const int64_t nv = (int64_t)a.HW * (C / 8);
  int gx = (int)((nv + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_gn_apply, dim3(gx, a.B), dim3(256), (size_t)2 * C * sizeof(float), (hipStream_t)stream, a);
  return finish_launch("k_gn_apply");
}

extern "C" int ctdd_unet_channel_stats(const float* x, int B, int HW, int C, double* stats, void* stream) {
  CTDD_REQUIRE(x && stats, CTDD_EINVAL, "null");
  hipLaunchKernelGGL(k_channel_stats, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, HW, C, stats);
  return finish_launch("k_channel_stats");
}

extern "C" int ctdd_unet_time(const void* args_, const float* proj_w, const float* proj_b, int Ntot, float* proj_out,
                              void* stream) {
  const TimeArgs& a = *(const TimeArgs*)args_;
  CTDD_REQUIRE(a.t && a.act && a.tdim % 4 == 0, CTDD_EINVAL, "bad time arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_time_mlp, dim3(a.B), dim3(256), (size_t)(a.ch + a.tdim) * sizeof(float), st, a);
  if (int rc = finish_launch("k_time_mlp")) return rc;
  hipLaunchKernelGGL(k_time_proj, dim3((Ntot + 255) / 256, a.B), dim3(256), (size_t)a.tdim * sizeof(float), st,
                     (const float*)a.act, proj_w, proj_b, a.B, a.tdim, Ntot, proj_out);
  return finish_launch("k_time_proj");
}

extern "C" int ctdd_unet_attention(const void* args_, void* stream) {
  const AttnArgs& a = *(const AttnArgs*)args_;
  const int ch = a.C / a.heads;
  const size_t lds = (size_t)(3 * a.T * ch + a.T * a.T) * sizeof(float);
  CTDD_REQUIRE(lds <= 160 * 1024, CTDD_ERANGE, "attention tile too large (T=%d)", a.T);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)k_attn_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_attn_small, dim3(a.B * a.heads), dim3(256), lds, (hipStream_t)stream, a);
  return finish_launch("k_attn_small");
}

extern "C" int ctdd_unet_logistic_head(const void* args_, void* stream) {
  const LogisticArgs& a = *(const LogisticArgs*)args_;
  const int64_t total = (int64_t)a.B * a.C * a.HW * a.S;
  hipLaunchKernelGGL(k_logistic_head, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return finish_launch("k_logistic_head");
}
