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
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;   // a 16-byte register quad (a native vector: plain loads, no struct memcpy)

__device__ inline unsigned pack2_bf16(float a, float b) {
  f32x2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
__device__ inline unsigned short to_bf16(float a) { return (unsigned short)(pack2_bf16(a, 0.0f) & 0xFFFFu); }
__device__ inline float from_bf16(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

enum SegKind { SEG_3x3 = 0, SEG_1x1 = 1, SEG_3x3_S2 = 2, SEG_3x3_UP = 3, SEG_3x3_S2T = 4 };

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
  int act;                    // 0 none, 1 ReLU, 2 GELU (erf), applied to acc + bias before the residual
  unsigned short* out_lo;     // row-major epilogue only: bf16(v - out_hi), the second term of a split operand, or null
};

constexpr int BM = 128;

// ------------------------------------------------------------------ epilogue shared by the conv kernels
// One accumulator tile: lane column n, 16 registers = rows wrow0 + (r&3) + 8*(r>>2) + 4*g.
// GroupNorm statistics of one workgroup tile: the lanes' (sample, column) partial sums meet in LDS
// (ds_add_f64) and leave as ONE global atomic per (sample, column, moment) and workgroup -- global
// fp64 atomics execute at the memory side at roughly a wave-instruction per 100+ cycles and CU, and
// one per lane and accumulator tile was costing more than the matrix work of a K = 864 convolution.
#ifndef CTDD_EPI_DBG
#define CTDD_EPI_DBG 0      // scratch ablations of the statistics path
#endif
struct TileStats {
  double* lds;       // [nb][BN][2] or null (no statistics / split-K: k_conv_finish computes them)
  int b0, nb, n0, BN;
};
__host__ __device__ inline int tile_stats_samples(int rows, int HW) { return (rows + HW - 1) / HW + 1; }
__device__ inline TileStats tile_stats_begin(const ConvArgs& a, unsigned char* smem, int64_t row0, int rows, int n0, int BN,
                                             int64_t M, int HW) {
  TileStats ts = {nullptr, 0, 0, n0, BN};
  if (!a.stats || a.ksplit > 1) return ts;
  const int64_t rlast = (row0 + rows - 1 < M ? row0 + rows - 1 : M - 1);
  ts.b0 = (int)((row0 < M ? row0 : M - 1) / HW);
  ts.nb = (int)(rlast / HW) - ts.b0 + 1;
  ts.lds = (double*)smem;
  __syncthreads();                           // the tiles in LDS are out of use
  for (int i = threadIdx.x; i < ts.nb * BN * 2; i += blockDim.x) ts.lds[i] = 0.0;
  __syncthreads();
  return ts;
}
__device__ inline void tile_stats_flush(const ConvArgs& a, const TileStats& ts) {
  if (!ts.lds) return;
  __syncthreads();
  for (int i = threadIdx.x; i < ts.nb * ts.BN * 2; i += blockDim.x) {
    const int b = ts.b0 + i / (ts.BN * 2), n = ts.n0 + (i >> 1) % ts.BN;
    const double v = ts.lds[i];
    if (n < a.N && b < a.B && v != 0.0 && !(CTDD_EPI_DBG & 2)) atomicAdd(a.stats + ((size_t)b * a.N + n) * 2 + (i & 1), v);
  }
}

__device__ inline void conv_epilogue_tile(const ConvArgs& a, const f32x16& acc, int64_t wrow0, int n, int g,
                                          int64_t M, int HW, const TileStats& ts) {
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
      if (a.act == 1) v = fmaxf(v, 0.0f);
      else if (a.act == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
      const size_t o = (size_t)p * a.N + n;
      if (a.res_f32) v += a.res_f32[o];
      else if (a.res_bf16) v += from_bf16(a.res_bf16[o]);
      size_t oo = o;                                       // logits layout (B, C*H*W, S): row = c*HW + p, fp32 and/or bf16
      if (a.logits_C > 0) {
        const int b = b_first + (second ? 1 : 0);
        oo = (((size_t)b * a.logits_C + lch) * HW + (p - (int64_t)b * HW)) * S + ls;
      }
      if (a.out_f32) a.out_f32[oo] = v;
      if (a.out_hi) a.out_hi[oo] = to_bf16(v);
      if (second) { s1 += v; q1 += (double)v * v; } else { s0 += v; q0 += (double)v * v; }
    }
  }
  if (ts.lds) {
    s0 += __shfl_xor(s0, 32, WAVE); q0 += __shfl_xor(q0, 32, WAVE);
    s1 += __shfl_xor(s1, 32, WAVE); q1 += __shfl_xor(q1, 32, WAVE);
    if (g == 0 && ncol && wrow0 < M) {
      double* st = ts.lds + ((size_t)(b_first - ts.b0) * ts.BN + (n - ts.n0)) * 2;
      atomicAdd(st, s0);
      atomicAdd(st + 1, q0);
      if (b_first + 1 < ts.b0 + ts.nb && (s1 != 0.0 || q1 != 0.0)) {
        atomicAdd(st + (size_t)ts.BN * 2, s1);
        atomicAdd(st + (size_t)ts.BN * 2 + 1, q1);
      }
    }
  }
}


// Row-major epilogue of the slab kernels (bf16 mode).  The accumulators of a 32-row slice of the
// wave's tile go through a wave-private fp32 LDS image and come back as 8-column pieces, one lane
// per (row, piece): bias / per-sample bias / residual are added with 16-byte loads and the outputs
// leave as 16-byte stores (the column-per-lane layout of the MFMA result costs one 2-byte store and
// one 2-byte residual load per element -- measured at half the run time of a K = 864 convolution).
// Each lane keeps its piece's column sums / sums of squares in fp32 over the <= 8 rows it visits per
// sample and adds them to the workgroup's fp64 LDS statistics.
// xt = this wave's [32][32*BNT + 4] fp32 region; rows wrow0 .. wrow0 + 32*MT of the output.
template <int BNT, int MT, bool EXT = false>   // EXT: activation + second bf16 output term (the plain-GEMM users); compiled out of the U-Net kernels
__device__ __attribute__((always_inline)) inline void conv_epilogue_rows(const ConvArgs& a, const f32x16 (&acc)[MT][BNT], float* xt, int64_t wrow0, int n0,
                                          int64_t M, int HW, const TileStats& ts, unsigned long long* epi_t = nullptr) {
#ifdef CTDD_RES_STAMPS
  unsigned long long e0, e1, e2, e3;
#define ESTAMP(t_) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
#else
#define ESTAMP(t_)
#endif
  constexpr int BN = 32 * BNT, LD = BN + 4, CH = BN / 8, RS = 64 / CH, STEPS = (32 + RS - 1) / RS;
  const int lane = threadIdx.x & 63, li = lane & 31, g = lane >> 5;
  // The argument struct lives in the kernel-argument segment; left alone the compiler re-loads a field next to every use
  // (260 scalar loads, each followed by a wait, in this unrolled epilogue).  Pin the integer fields in scalar registers.
  auto pin32 = [](int v) { asm volatile("" : "+s"(v)); return v; };
  const int aN = pin32(a.N), aB = pin32(a.B), a_tbs = pin32(a.tb_stride), a_lc = pin32(a.logits_C);
  const int a_act = EXT ? pin32(a.act) : 0;
  // (pointers stay as they are: laundering them through an integer loses the global address space -> flat_load/flat_store)
  float* const p_out_f32 = a.out_f32;
  unsigned short* const p_out_hi = a.out_hi;
  unsigned short* const p_out_lo = EXT ? a.out_lo : nullptr;
  const unsigned short* const p_res_bf16 = a.res_bf16;
  const float* const p_res_f32 = a.res_f32;
  const float* const p_tbias = a.tbias;
  if (a.ksplit > 1) {                                     // partial sums only; k_conv_finish does the rest
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < BNT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t p = wrow0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          const int n = n0 + 32 * t + li;
          if (p < M && n < a.N) atomicAdd(a.acc_buf + (size_t)p * a.N + n, acc[mt][t][r]);
        }
    return;
  }
  const int cc = lane % CH, rsub = lane / CH;
  const int nc = n0 + cc * 8;                              // first of this lane's 8 columns
  const bool lane_ok = rsub < RS && nc < aN;              // (N is a multiple of 8)
  float bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bv[j] = (a.bias && lane_ok) ? a.bias[nc + j] : 0.0f;
  const int S = a_lc > 0 ? aN / a_lc : 1;     // logits layout: 8 consecutive n stay inside one channel (S % 8 == 0)
  const int lch = a_lc > 0 ? nc / S : 0, ls = a_lc > 0 ? nc % S : 0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int64_t base = wrow0 + mt * 32;
    if (base >= M) break;
    ESTAMP(e0)
    // residual pieces of this lane's rows: issued before the LDS passes so that their latency hides behind them
    uint4 rres[STEPS];
    if (p_res_bf16) {
#pragma unroll
      for (int step = 0; step < STEPS; ++step) {
        const int row = step * RS + rsub;
        const int64_t p = base + row;
        rres[step] = (lane_ok && row < 32 && p < M) ? *(const uint4*)(p_res_bf16 + (size_t)p * aN + nc) : make_uint4(0, 0, 0, 0);
      }
    }
    // per-sample (time) bias of the one or two samples of this slice: requested here, used after the LDS round trip
    const int b_first = (int)(base / HW);
    const int64_t next_sample = (int64_t)(b_first + 1) * HW;   // a 32-row slice spans at most two samples (HW >= 32)
    float tb0[8], tb1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      tb0[j] = (p_tbias && lane_ok) ? p_tbias[(size_t)b_first * a_tbs + nc + j] : 0.0f;
      tb1[j] = (p_tbias && lane_ok && b_first + 1 < aB) ? p_tbias[(size_t)(b_first + 1) * a_tbs + nc + j] : 0.0f;
    }
    // ---- phase 1: column-per-lane accumulators -> row-major fp32 image
#pragma unroll
    for (int t = 0; t < BNT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) xt[((r & 3) + 8 * (r >> 2) + 4 * g) * LD + t * 32 + li] = acc[mt][t][r];
    ESTAMP(e1)
    // ---- phase 2: one lane per (row, 8-column piece).  All the LDS reads of the slice are issued first, unconditionally
    // (row clamped), so that their latencies overlap instead of one read -> wait -> use chain per step
    // Statistics of a slice that lies inside ONE sample (all of them at 28x28, none at 7x7; wave-uniform): the lane sums its
    // own rows' final values in registers, the RS lanes of a column piece meet through 2 RS rows of the (by then consumed)
    // image, and the column pass adds RS partial sums instead of walking 32 rows.
    const int nrows_s = (int)(M - base < 32 ? M - base : 32);
    const bool one_sample = ts.lds && next_sample - base >= nrows_s && 2 * RS <= 32;
    float cs[8], cq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { cs[j] = 0.0f; cq[j] = 0.0f; }
    float4 xr0[STEPS], xr1[STEPS];
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
      const int rowc = min(step * RS + rsub, 31);
      xr0[step] = *(const float4*)(xt + rowc * LD + cc * 8);
      xr1[step] = *(const float4*)(xt + rowc * LD + cc * 8 + 4);
    }
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
      const int row = step * RS + rsub;
      const int64_t p = base + row;
      if (lane_ok && row < 32 && p < M) {
        const bool second = p >= next_sample;
        const float4 x0 = xr0[step], x1 = xr1[step];
        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bv[j] + (second ? tb1[j] : tb0[j]);
        if (a_act == 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.0f);
        } else if (a_act == 2) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.5f * v[j] * (1.0f + erff(v[j] * 0.70710678118654752f));
        }
        const size_t o = (size_t)p * aN + nc;
        if (p_res_bf16) {
          const uint4 rr = rres[step];
          const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[2 * j] += __uint_as_float(rw[j] << 16);
            v[2 * j + 1] += __uint_as_float(rw[j] & 0xFFFF0000u);
          }
        } else if (p_res_f32) {
          const float4 r0 = *(const float4*)(p_res_f32 + o), r1 = *(const float4*)(p_res_f32 + o + 4);
          v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
        }
        size_t oo = o;                                     // logits layout (B, C*H*W, S): row = c*HW + p, fp32 and/or bf16
        if (a_lc > 0) {
          const int b = b_first + (second ? 1 : 0);
          oo = (((size_t)b * a_lc + lch) * HW + (p - (int64_t)b * HW)) * S + ls;
        }
        if (p_out_f32) {
          float* dst = p_out_f32 + oo;
          *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
          *(float4*)(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        if (p_out_hi) {
          const uint4 hv = make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
          *(uint4*)(p_out_hi + oo) = hv;
          if (p_out_lo) {
            const unsigned hw[4] = {hv.x, hv.y, hv.z, hv.w};
            unsigned lw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              lw[j] = pack2_bf16(v[2 * j] - __uint_as_float(hw[j] << 16), v[2 * j + 1] - __uint_as_float(hw[j] & 0xFFFF0000u));
            *(uint4*)(p_out_lo + o) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
          }
        }
        if (one_sample) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { cs[j] += v[j]; cq[j] = fmaf(v[j], v[j], cq[j]); }
        } else if (ts.lds) {                               // final values back into the image for the column pass
          *(float4*)(xt + row * LD + cc * 8) = make_float4(v[0], v[1], v[2], v[3]);
          *(float4*)(xt + row * LD + cc * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
      }
    }
    if (one_sample && lane_ok) {                           // (rsub < RS; every image read of this slice was issued above)
      *(float4*)(xt + rsub * LD + cc * 8) = make_float4(cs[0], cs[1], cs[2], cs[3]);
      *(float4*)(xt + rsub * LD + cc * 8 + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
      *(float4*)(xt + (RS + rsub) * LD + cc * 8) = make_float4(cq[0], cq[1], cq[2], cq[3]);
      *(float4*)(xt + (RS + rsub) * LD + cc * 8 + 4) = make_float4(cq[4], cq[5], cq[6], cq[7]);
    }
    ESTAMP(e2)
    // ---- phase 3: GroupNorm statistics, one lane per column down the slice's rows (fp64), so that the
    // LDS atomics are one per (sample, column, moment) and wave slice and hit distinct addresses
    // (a 64-bit LDS atomic costs ~75 cycles per wave-instruction, more when lanes share an address)
    if (ts.lds && !(CTDD_EPI_DBG & 1)) {
      const int nrows = (int)(M - base < 32 ? M - base : 32);
      const int split = (int)(next_sample - base < nrows ? next_sample - base : nrows);   // rows [0, split) belong to b_first
      if (one_sample) {
#pragma unroll
        for (int c0 = 0; c0 < BN; c0 += 64) {
          const int c = c0 + lane;
          if (c < BN && n0 + c < aN) {
            float sv[RS], qv[RS];
#pragma unroll
            for (int r = 0; r < RS; ++r) { sv[r] = xt[r * LD + c]; qv[r] = xt[(RS + r) * LD + c]; }
            double s0 = 0.0, q0 = 0.0;
#pragma unroll
            for (int r = 0; r < RS; ++r) { s0 += (double)sv[r]; q0 += (double)qv[r]; }
            double* st = ts.lds + ((size_t)(b_first - ts.b0) * ts.BN + c) * 2;
            atomicAdd(st, s0);
            atomicAdd(st + 1, q0);
          }
        }
      } else
#pragma unroll
      for (int c0 = 0; c0 < BN; c0 += 64) {
        const int c = c0 + lane;
        if (c < BN && n0 + c < aN) {
          double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
#pragma unroll
          for (int r8 = 0; r8 < 32; r8 += 8) {             // eight independent LDS reads in flight; fp32 sums of eight rows, fp64 across blocks
            float xv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) xv[k] = xt[(r8 + k) * LD + c];
            float a0 = 0.0f, b0 = 0.0f, a1 = 0.0f, b1 = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const int row = r8 + k;                      // (row < split, row < nrows: wave-uniform)
              const float x0 = row < split ? xv[k] : 0.0f;
              const float x1 = (row >= split && row < nrows) ? xv[k] : 0.0f;
              a0 += x0; b0 = fmaf(x0, x0, b0);
              a1 += x1; b1 = fmaf(x1, x1, b1);
            }
            s0 += (double)a0; q0 += (double)b0; s1 += (double)a1; q1 += (double)b1;
          }
          double* st = ts.lds + ((size_t)(b_first - ts.b0) * ts.BN + c) * 2;
          atomicAdd(st, s0);
          atomicAdd(st + 1, q0);
          if (split < nrows) {
            atomicAdd(st + (size_t)ts.BN * 2, s1);
            atomicAdd(st + (size_t)ts.BN * 2 + 1, q1);
          }
        }
      }
    }
#ifdef CTDD_RES_STAMPS
    ESTAMP(e3)
    if (epi_t) { epi_t[0] += e1 - e0; epi_t[1] += e2 - e1; epi_t[2] += e3 - e2; }
#endif
  }
}
// bytes of LDS the row-major epilogue needs for NW waves behind the statistics of `rows` output rows
__host__ __device__ inline size_t epilogue_rows_lds(int nw, int bnt, int rows, int HW) {
  return (size_t)tile_stats_samples(rows, HW) * 32 * bnt * 16 + (size_t)nw * 32 * (32 * bnt + 4) * 4;
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

  // Chunks in flight in registers: one.  (Two or three sets for the 32-column tiles of the tiny grids were measured, -DCTDD_IGEMM_DP:
  // 39.6 / 40.5 / 42.2 us for the 7x7 stride-2 convolution of 128 samples -- the compiler's wait in front of the loop's first store
  // falls back to vmcnt(0), so the extra sets only cost registers.  What did pay is below: branch-free staging loads.)
#ifndef CTDD_IGEMM_DP
#define CTDD_IGEMM_DP 1
#endif
  constexpr int DP = (BNT == 1 && !F32) ? CTDD_IGEMM_DP : 1;
  uint4 rra[DP][AV], rrb[DP][BV];
  unsigned rok[DP];
  auto load_chunk = [&](const ChunkIter<BK>& it, uint4 (&ra)[AV], uint4 (&rb)[BV]) {
    const ConvSeg sg = a.seg[it.seg];
    const int kind = sg.kind;
    const unsigned char* src = F32 ? (const unsigned char*)sg.f32 : (const unsigned char*)sg.hi;
    const unsigned char* wsrc = F32 ? (const unsigned char*)a.w_f32 : (const unsigned char*)a.w_hi;
    int dy = 0, dx = 0;
    if (kind != SEG_1x1) { dy = it.tap / 3; dx = it.tap % 3; }
    // Branch-free per vector: the segment kind only selects multipliers / offsets (wave-uniform scalars), validity is a bit
    // product, the address is clamped and the load unconditional -- written as `ok ? load : 0` under a switch on the kind the
    // compiler made every one of a chunk's vectors its own branch + load + wait (eight exposed latencies per chunk: the
    // stride-2 convolutions of the sampler's plans went 55.8 -> 39.6 and 37.8 -> 31.0 us per 128 samples with this form).
    const bool s2 = kind == SEG_3x3_S2, s2t = kind == SEG_3x3_S2T, up = kind == SEG_3x3_UP, one = kind == SEG_1x1;
    const int mul = s2 ? 2 : 1;                                        // input coordinate = mul * output + add (before the halving of S2T / UP)
    const int ady = one ? 0 : s2 ? dy : s2t ? -dy : dy - 1, adx = one ? 0 : s2 ? dx : s2t ? -dx : dx - 1;
    const int hin = one ? a.H : a.Hin, win = one ? a.W : a.Win;
    const int hlim = up ? a.H : hin, wlim = up ? a.W : win;            // bound of (yy, xx) before the halving
    const int sh = (s2t || up) ? 1 : 0;
    size_t offs[AV];
    bool oks[AV];
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int v = tid + 256 * i;
      const int cv = v % VPR;
      const int yy = mul * py[i] + ady, xx = mul * px[i] + adx;
      bool ok = (pb[i] >= 0) & (yy >= 0) & (xx >= 0) & (yy < (s2t ? 2 * hin : hlim)) & (xx < (s2t ? 2 * win : wlim));
      ok = ok & !(s2t & (((yy | xx) & 1) != 0));                      // transpose of the stride-2 conv: only even coordinates hit an input
      const int yh = yy >> sh, xh = xx >> sh;
      oks[i] = ok;
      offs[i] = ok ? (((size_t)pb[i] * hin + yh) * win + xh) * sg.C + it.c0 + cv * EPV : 0;
    }
#pragma unroll
    for (int i = 0; i < AV; ++i) ra[i] = *(const uint4*)(src + offs[i] * ESZ);
    unsigned okm = 0;                                                  // validity bits: applied when the vectors go to LDS (store_chunk), not
#pragma unroll                                                         // here -- a select on the loaded value would wait for it now
    for (int i = 0; i < AV; ++i) okm |= (oks[i] ? 1u : 0u) << i;
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const int v = tid + 256 * i;
      const int n = v / VPR, cv = v % VPR;
      const bool ok = (v < BN * VPR) & (n0 + n < a.N);
      const size_t off = ok ? (size_t)(n0 + n) * a.Ktot + it.koff + cv * EPV : 0;
      rb[i] = *(const uint4*)(wsrc + off * ESZ);
      okm |= (ok ? 1u : 0u) << (AV + i);
    }
    return okm;
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
  auto store_chunk = [&](const uint4 (&ra)[AV], const uint4 (&rb)[BV], unsigned okm) {
    auto masked = [](uint4 v, bool ok) { return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u); };
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int v = tid + 256 * i;
      put(As, v / VPR, v % VPR, masked(ra[i], (okm >> i) & 1u));
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const int v = tid + 256 * i;
      if (v < BN * VPR) put(Bs, v / VPR, v % VPR, masked(rb[i], (okm >> (AV + i)) & 1u));
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
#pragma unroll
  for (int j = 0; j < DP; ++j)
  {                                     // (loads are unconditional -- a set past the end re-reads the last chunk and is never stored --
    rok[j] = load_chunk(it, rra[j], rrb[j]);          //  so that every path has the same number of loads in flight: with a conditional
    if (j + 1 < nchunk) it.next(a);                   //  refill the compiler's wait before the next store fell back to vmcnt(0))
  }
  const int li = lane & 31, g = lane >> 5;
  for (int c0_ = 0; c0_ < nchunk; c0_ += DP) {
#pragma unroll
  for (int j = 0; j < DP; ++j) {
    const int c = c0_ + j;
    if (c >= nchunk) break;
    __syncthreads();                    // everyone finished reading the previous chunk from LDS
    store_chunk(rra[j], rrb[j], rok[j]);
    __syncthreads();
    rok[j] = load_chunk(it, rra[j], rrb[j]);          // refill this register set: chunk c + DP, in flight behind the next DP chunks' MFMAs
    if (c + DP + 1 < nchunk) it.next(a);
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
  }

  // ---- epilogue.  lane: column n = n0 + 32 t + li ; register r: row 32*wave + (r&3) + 8*(r>>2) + 4*g
  const TileStats ts = tile_stats_begin(a, smem, m0, BM, n0, BN, M, HW);
#pragma unroll
  for (int t = 0; t < BNT; ++t) conv_epilogue_tile(a, acc[t], m0 + wave * 32, n0 + 32 * t + li, g, M, HW, ts);
  tile_stats_flush(a, ts);
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
// ALLTAPS (the 32-column tiles of the small levels, 7x7 / 4x4): all NINE taps' weight tiles of a channel chunk are staged
// together with its slab -- one barrier pair and one round of global loads per chunk instead of one per tap.  With a
// [32][BK] weight tile per tap the per-tap version spent a global-load latency (~1.5k cycles) on every 4 matrix
// instructions: a 7x7 convolution of 128 samples ran at 6 % of the matrix pipe.
// DIRECT (bf16 output, no statistics, no logits layout, no split-K -- the small levels of the inference plans since their
// GroupNorm computes its own statistics): the matrix instruction's operands are swapped, C^T = W . X^T, so that a lane holds ONE
// PIXEL's 4-channel pieces (channels 8q + 4g + 0..3 of each 32-channel tile) instead of one channel's 16 pixels.  Bias and
// time bias are the accumulators' initial values, the residual is added as 8-byte pieces in that layout, two
// v_permlane32_swap per 16 channels give each lane 8 consecutive channels, and the row leaves as 16-byte stores straight from
// the registers: no LDS image, no barrier, ~2 instructions per output instead of ~8 (the row-major epilogue was 8 k of a 7x7
// workgroup's 30 k cycles).
template <int BK, int BNT, int WM, bool EXT = false, bool ALLTAPS = false, bool DIRECT = false>
__global__ __launch_bounds__(256, 2) void k_conv_patch(const ConvArgs a) {
  static_assert(!(DIRECT && EXT), "direct epilogue: plain bf16 outputs only");
  constexpr int BN = 32 * BNT, BMP = 4 * WM, MT = WM / 32;
  constexpr int LDK = BK + 8, VPR = BK / 8;
  constexpr int BV = (BN * VPR + 255) / 256;
  constexpr int PVMAX = ((BMP + 2 * 34) * VPR + 255) / 256;     // W <= 33
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef CTDD_PATCH_STAMPS
  unsigned long long pst[8];
  int psi = 0;
#define PSTAMP() { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (psi < 8) pst[psi++] = t_; }
#else
#define PSTAMP()
#endif
  PSTAMP()
  const int li = lane & 31, g = lane >> 5;
  const int HW = a.H * a.W, Wd = a.W;
  const int64_t M = (int64_t)a.B * HW;
  const int64_t p0 = (int64_t)blockIdx.x * BMP;
  const int n0 = blockIdx.y * BN;
  const int halo = Wd + 1;
  const int PR = BMP + 2 * halo;                                // slab rows
  unsigned short* Ap = (unsigned short*)smem;                   // [PR][LDK]
  unsigned short* Bs = Ap + (size_t)PR * LDK;                   // [2][BN][LDK]   (ALLTAPS: [9][BN][LDK])
  constexpr int NRB = ALLTAPS ? 9 * BV : BV;
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

  uint4 rp[PVMAX], rb[NRB];
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
  auto load_b_all = [&](int kb, int C, int ntap_) {             // ALLTAPS: the weight tiles of every tap of one chunk
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int i = 0; i < BV; ++i) {
        const int v = tid + 256 * i;
        const int n = v / VPR;
        const bool ok = tp < ntap_ && v < BN * VPR && n0 + n < a.N;
        if (ALLTAPS)
          rb[(ALLTAPS ? tp : 0) * BV + i] = ok ? *(const uint4*)(a.w_hi + (size_t)(n0 + n) * a.Ktot + kb + tp * C + (v % VPR) * 8) : make_uint4(0, 0, 0, 0);
      }
  };
  auto store_b_all = [&](int ntap_) {
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int i = 0; i < BV; ++i) {
        const int v = tid + 256 * i;
        if (ALLTAPS && tp < ntap_ && v < BN * VPR)
          *(uint4*)(Bs + ((size_t)tp * BN + v / VPR) * LDK + (v % VPR) * 8) = rb[(ALLTAPS ? tp : 0) * BV + i];
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
  if constexpr (DIRECT) {                                        // register r of tile (mt, t): channel n0 + 32 t + 8 (r >> 2) + 4 g + (r & 3)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int64_t p = p0 + wave * WM + mt * 32 + li;
      const int bsm = (int)((p < M ? p : M - 1) / HW);
#pragma unroll
      for (int t = 0; t < BNT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + 32 * t + 8 * q + 4 * g;
          if (n < a.N) {
            float4 v = a.bias ? *(const float4*)(a.bias + n) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (a.tbias) {
              const float4 tv = *(const float4*)(a.tbias + (size_t)bsm * a.tb_stride + n);
              v.x += tv.x; v.y += tv.y; v.z += tv.z; v.w += tv.w;
            }
            acc[mt][t][4 * q] = v.x; acc[mt][t][4 * q + 1] = v.y; acc[mt][t][4 * q + 2] = v.z; acc[mt][t][4 * q + 3] = v.w;
          }
        }
    }
  }

  int u = zs;
  int sgi, c0, kbase;
  if (u < nunits) {
    unit_info(u, sgi, c0, kbase);
    load_patch(sgi, c0);
    if (ALLTAPS) load_b_all(kbase, a.seg[sgi].C, a.seg[sgi].kind == SEG_1x1 ? 1 : 9);
    else load_b(kbase);
  }
  int bbuf = 0;
  while (u < nunits) {
    const ConvSeg sg = a.seg[sgi];
    const bool one = sg.kind == SEG_1x1;
    const int ntap = one ? 1 : 9;
    const int hl = one ? 0 : halo;
    __syncthreads();                       // previous unit's slab and weight tiles are out of use
    store_patch();
    if (ALLTAPS) store_b_all(ntap);
    else store_b(bbuf);
    __syncthreads();
    PSTAMP()
    const int un = u + nz;                 // next unit of this slice
    int nsgi = 0, nc0 = 0, nkbase = 0;
    if (un < nunits) unit_info(un, nsgi, nc0, nkbase);
    if (ALLTAPS && un < nunits) {          // next chunk's slab and all its weight tiles: in flight under this chunk's taps
      load_patch(nsgi, nc0);
      load_b_all(nkbase, a.seg[nsgi].C, a.seg[nsgi].kind == SEG_1x1 ? 1 : 9);
    }
    if constexpr (ALLTAPS && BNT == 1 && MT == 1) {
      // straight-line chunk: 9 taps x BK/16 K-steps (or BK/16 for a 1x1 segment), the fragments of step i+1 requested from
      // LDS before the matrix instruction of step i issues (with one MFMA per fragment pair a rolled tap loop exposed the
      // LDS latency on every instruction)
      constexpr int KS = BK / 16;
      const unsigned short* A0 = Ap + (size_t)(hl + wave * WM + li) * LDK + g * 8;
      const unsigned short* B0 = Bs + (size_t)li * LDK + g * 8;
      constexpr int DEPTH = 4;            // fragment pairs in flight: ~128 cycles of LDS latency / 32 cycles per matrix instruction
      uint4 fa[DEPTH], fb[DEPTH];
      auto fetch = [&](int tap_, int ks_, int buf_) {
        const int dy = one ? 1 : tap_ / 3, dx = one ? 1 : tap_ % 3;
        const int shift = (dy - 1) * Wd + (dx - 1);
        const bool okk = oin[0] && (unsigned)(oy[0] + dy - 1) < (unsigned)a.H && (unsigned)(ox[0] + dx - 1) < (unsigned)Wd;
        uint4 raw = *(const uint4*)(A0 + (ptrdiff_t)shift * LDK + ks_ * 16);
        if (!okk) raw = make_uint4(0, 0, 0, 0);
        fa[buf_] = raw;
        fb[buf_] = *(const uint4*)(B0 + (size_t)tap_ * BN * LDK + ks_ * 16);
      };
      if (one) {
        static_assert(KS <= DEPTH, "1x1 chunk: all K-steps in flight at once");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fetch(0, ks, ks);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          acc[0][0] = DIRECT ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[ks]), __builtin_bit_cast(bf16x8, fa[ks]), acc[0][0], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[ks]), __builtin_bit_cast(bf16x8, fb[ks]), acc[0][0], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) fetch(i / KS, i % KS, i);
#pragma unroll
        for (int i = 0; i < 9 * KS; ++i) {
          const bf16x8 af_ = __builtin_bit_cast(bf16x8, fa[i % DEPTH]), bf_ = __builtin_bit_cast(bf16x8, fb[i % DEPTH]);
          if (i + DEPTH < 9 * KS) fetch((i + DEPTH) / KS, (i + DEPTH) % KS, i % DEPTH);
          acc[0][0] = DIRECT ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf_, af_, acc[0][0], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af_, bf_, acc[0][0], 0, 0, 0);
        }
      }
    } else
    for (int tap = 0; tap < ntap; ++tap) {
      if (ALLTAPS) bbuf = tap;
      // prefetch: next tap's weight tile, or (at the last tap) the next unit's first one + its slab
      if (!ALLTAPS) {
        if (tap + 1 < ntap) load_b(kbase + (tap + 1) * sg.C);
        else if (un < nunits) load_b(nkbase);
        if (tap == (ntap > 4 ? 4 : 0) && un < nunits) load_patch(nsgi, nc0);
      }
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
            acc[mt][t] = DIRECT ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, af[mt], acc[mt][t], 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bf, acc[mt][t], 0, 0, 0);
        }
      }
      if (!ALLTAPS && tap + 1 < ntap) {
        store_b(bbuf ^ 1);                 // the other buffer was last read before the previous barrier
        __syncthreads();
        bbuf ^= 1;
      }
    }
    u = un; sgi = nsgi; c0 = nc0; kbase = nkbase;
  }

  PSTAMP()
  if constexpr (DIRECT) {
    unsigned short* const outp = a.out_hi;
    const unsigned short* const resp = a.res_bf16;
    const int aN = a.N;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int64_t p = p0 + wave * WM + mt * 32 + li;
      const bool prow = p < M;
      const size_t rowoff = (size_t)(prow ? p : 0) * aN;
      uint2 rr[BNT][4];
      if (resp) {
#pragma unroll
        for (int t = 0; t < BNT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = n0 + 32 * t + 8 * q + 4 * g;
            rr[t][q] = (prow && n < aN) ? *(const uint2*)(resp + rowoff + n) : make_uint2(0u, 0u);
          }
      }
#pragma unroll
      for (int t = 0; t < BNT; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          unsigned w[2][2];                                      // [piece q = 2h, 2h + 1][two packed pairs]
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int q = 2 * h + e;
            float v0 = acc[mt][t][4 * q], v1 = acc[mt][t][4 * q + 1], v2 = acc[mt][t][4 * q + 2], v3 = acc[mt][t][4 * q + 3];
            if (resp) {
              v0 += __uint_as_float(rr[t][q].x << 16); v1 += __uint_as_float(rr[t][q].x & 0xFFFF0000u);
              v2 += __uint_as_float(rr[t][q].y << 16); v3 += __uint_as_float(rr[t][q].y & 0xFFFF0000u);
            }
            w[e][0] = pack2_bf16(v0, v1); w[e][1] = pack2_bf16(v2, v3);
          }
          // lower lane of the pair: (its piece 2h, the upper lane's piece 2h) = channels 16h + 0..7; upper lane: the two pieces
          // 2h + 1 = channels 16h + 8..15
          const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
          const int n = n0 + 32 * t + 16 * h + 8 * g;
          if (prow && n < aN) *(uint4*)(outp + rowoff + n) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
    }
  } else {
  const TileStats ts = tile_stats_begin(a, smem, p0, BMP, n0, BN, M, HW);     // (barrier: the LDS tiles are out of use)
  if (!ts.lds) __syncthreads();
  float* xt = (float*)(smem + (size_t)tile_stats_samples(BMP, HW) * BN * 16) + (size_t)wave * 32 * (BN + 4);
  conv_epilogue_rows<BNT, MT, EXT>(a, acc, xt, p0 + wave * WM, n0, M, HW, ts);
  PSTAMP()
  tile_stats_flush(a, ts);
  }
  PSTAMP()
#ifdef CTDD_PATCH_STAMPS
  if (lane == 0 && a.acc_buf) {
    unsigned long long* o = (unsigned long long*)a.acc_buf + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
    for (int i = 0; i < 8; ++i) o[i] = i < psi ? pst[i] : 0;
  }
#endif
}

// ------------------------------------------------------------------ resident-weights patch convolution
// Same slab idea as k_conv_patch, but the weight tiles of ALL taps of a 32-channel chunk are
// staged together with the slab, so the nine taps run back to back with no barrier and no LDS
// store between them: per chunk two barriers frame 9 x 2 x MT x BNT matrix instructions per wave.
// 8 waves x 64 rows = 512 output pixels per workgroup; LDS = slab (<= 46 KB) + 9 x [32*BNT][32]
// weights (69 KB at BNT = 3), one workgroup per CU with two waves per SIMD.  The next chunk's slab
// and weights are prefetched into registers under the matrix instructions.
#ifdef CTDD_RES_STAMPS      // diagnostic build only: per-wave phase durations go to a.acc_buf
#define RSTAMP(t_) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
#else
#define RSTAMP(t_)
#endif
#ifndef CTDD_RES_DBG
#define CTDD_RES_DBG 0      // scratch ablations: -1 no matrix work, -2 no restaging, -3 no masks, -4 no LDS reads, -5 reads only
#endif
template <int BNT>
__global__ __launch_bounds__(512) void k_conv_res(const ConvArgs a) {
  constexpr int BK = 32, WM = 64, MT = 2, NW = 8;
  constexpr int BN = 32 * BNT, BMP = NW * WM;
  constexpr int LDK = BK + 8, VPR = BK / 8;
  constexpr int BVT = BN * VPR;                                  // weight vectors per tap: thread -> (n, cv), 9 taps each
  static_assert(BVT <= 512, "one weight vector per thread and tap");
  constexpr int PVMAX = ((BMP + 2 * 34) * VPR + 511) / 512;     // W <= 33
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, g = lane >> 5;
  const int HW = a.H * a.W, Wd = a.W;
  const int64_t M = (int64_t)a.B * HW;
  const int64_t p0 = (int64_t)blockIdx.x * BMP;
  const int n0 = blockIdx.y * BN;
  const int halo = Wd + 1;
  const int PR = BMP + 2 * halo;
  unsigned short* Ap = (unsigned short*)smem;                   // [PR][LDK]
  unsigned short* Bs = Ap + (size_t)PR * LDK;                   // [9][BN][LDK]
  const int npv = PR * VPR;

  // per-lane validity of (tap, row tile): bit tap*MT + mt
  unsigned vmask = 0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int64_t p = p0 + wave * WM + mt * 32 + li;
    const bool in = p < M;
    const int r = (int)((in ? p : 0) % HW);
    const int oy = r / Wd, ox = r % Wd;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      const bool ok = in && (unsigned)(oy + dy - 1) < (unsigned)a.H && (unsigned)(ox + dx - 1) < (unsigned)Wd;
      vmask |= (ok ? 1u : 0u) << (tap * MT + mt);
    }
  }

  int nunits = 0;
  for (int sgi = 0; sgi < a.nseg; ++sgi) nunits += a.seg[sgi].C / BK;
  const int zs = blockIdx.z, nz = a.ksplit > 1 ? a.ksplit : 1;

  // staging registers: slab vector i of this thread = slab row tid/4 + 128 i, 16-byte column tid%4;
  // weights: thread (n = tid/4, cv = tid%4) takes that 16-byte piece of all nine taps.  All the
  // addresses are one per-thread offset plus wave-uniform terms, recomputed per unit (the opaque
  // asm keeps the compiler from hoisting a dozen of loop-invariant pointers into registers).
  u32x4 rp[PVMAX], rb[9];
  auto unit_info = [&](int u, int& sgi, int& c0, int& kbase) {
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
  auto load_unit = [&](int sgi, int c0, int kbase) {
    const ConvSeg sg = a.seg[sgi];
    const bool one = sg.kind == SEG_1x1;
    const int hl = one ? 0 : halo;
    int tv = tid;
    asm volatile("" : "+v"(tv));
    const int row = tv >> 2, cv = tv & 3;
    // Every load is unconditional on a clamped address: slab rows outside the tensor (and weight
    // rows >= N) hold other rows' data, which is harmless -- such slab rows are only read by taps
    // whose validity bit is clear (the fragment is zeroed), such columns are never written back.
    // (Predicated loads made the compiler wait for memory in the middle of the issue sequence.)
    const int64_t q0 = p0 - hl;                                       // wave-uniform first slab pixel
    const unsigned short* sbase = sg.hi + c0 + cv * 8;
#pragma unroll
    for (int i = 0; i < PVMAX; ++i) {
      int64_t q = q0 + row + 128 * i;
      q = q < 0 ? 0 : (q >= M ? M - 1 : q);
      rp[i] = *(const u32x4*)(sbase + (size_t)q * sg.C);
    }
    int nrow = n0 + row;
    nrow = nrow < a.N ? nrow : a.N - 1;
    const unsigned short* wbase = a.w_hi + (size_t)nrow * a.Ktot + kbase + cv * 8;
    if (wave * 64 < BVT) {                                            // whole waves: BVT is a multiple of 64
      rb[0] = *(const u32x4*)wbase;
      if (!one) {
#pragma unroll
        for (int tap = 1; tap < 9; ++tap) rb[tap] = *(const u32x4*)(wbase + (size_t)tap * sg.C);
      }
    }
  };
  auto store_unit = [&](bool one) {
    int tv = tid;
    asm volatile("" : "+v"(tv));
    const int row = tv >> 2, cv = tv & 3;
    unsigned short* at = Ap + (size_t)row * LDK + cv * 8;
#pragma unroll
    for (int i = 0; i < PVMAX; ++i)
      if (row + 128 * i < PR) *(u32x4*)(at + (size_t)i * 128 * LDK) = rp[i];
    unsigned short* bt = Bs + (size_t)row * LDK + cv * 8;
    if (wave * 64 < BVT) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
        if (tap == 0 || !one) *(u32x4*)(bt + (size_t)tap * BN * LDK) = rb[tap];
    }
  };

  f32x16 acc[MT][BNT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < BNT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][t][i] = 0.0f;

  // (tap, k-step) pipeline: the fragments of step i+1 are read from LDS before the matrix
  // instructions of step i are issued (two fragment sets in registers); sched_barrier pins that order
  auto read_frags = [&](const unsigned short* Aw, const unsigned short* Bw, u32x4 (&af)[MT], u32x4 (&bf)[BNT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = *(const u32x4*)(Aw + (size_t)mt * 32 * LDK);
#pragma unroll
    for (int t = 0; t < BNT; ++t) bf[t] = *(const u32x4*)(Bw + (size_t)t * 32 * LDK);
  };
  auto mfma_step = [&](u32x4 (&af)[MT], const u32x4 (&bf)[BNT], unsigned okbits) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      if (!((okbits >> mt) & 1u)) af[mt] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int t = 0; t < BNT; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[mt]), __builtin_bit_cast(bf16x8, bf[t]),
                                                             acc[mt][t], 0, 0, 0);
  };

#ifdef CTDD_RES_STAMPS
  unsigned long long tk0, tk1, tk2, tk3, tk4, tk5, dur[6] = {0, 0, 0, 0, 0, 0}, tstart;
  RSTAMP(tstart)
#endif
  int u = zs;
  int sgi = 0, c0 = 0, kbase = 0;
  if (u < nunits) {
    unit_info(u, sgi, c0, kbase);
    load_unit(sgi, c0, kbase);
  }
  while (u < nunits) {
    const bool one = a.seg[sgi].kind == SEG_1x1;
    RSTAMP(tk0)
    __syncthreads();                       // previous unit's slab and weights are out of use
    RSTAMP(tk1)
    if (CTDD_RES_DBG != -2 || u == zs) store_unit(one);
    RSTAMP(tk2)
    __syncthreads();
    RSTAMP(tk3)
    const int un = u + nz;
    int nsgi = 0, nc0 = 0, nkbase = 0;
    if (un < nunits) {
      unit_info(un, nsgi, nc0, nkbase);
      if (CTDD_RES_DBG != -2) load_unit(nsgi, nc0, nkbase);
    }
    RSTAMP(tk4)
    // tap loop (runtime trip count: 9, or 1 for a 1x1 segment), two k-steps per tap, software
    // pipelined over two fragment sets: set 1 (k-step 1) is read before the k-step-0 matrix
    // instructions issue, the next tap's set 0 before the k-step-1 ones
    const unsigned short* Bw = Bs + (size_t)li * LDK + g * 8;
    const unsigned short* A0 = Ap + (size_t)((one ? 0 : halo) + wave * WM + li) * LDK + g * 8;
    const int ntap = CTDD_RES_DBG == -1 ? 0 : (one ? 1 : 9);
    u32x4 fa[2][MT], fb[2][BNT];
    read_frags(A0 + (one ? 0 : (-Wd - 1) * LDK), Bw, fa[0], fb[0]);
    constexpr int dbg = CTDD_RES_DBG;
#pragma unroll 1
    for (int tap = 0; tap < ntap; ++tap) {
      const int aoff = one ? 0 : ((tap / 3 - 1) * Wd + (tap % 3 - 1)) * LDK;
      const unsigned okbits = dbg == -3 ? 3u : (vmask >> ((one ? 4 : tap) * MT)) & 3u;
      if constexpr (dbg != -4) read_frags(A0 + aoff + 16, Bw + (size_t)tap * BN * LDK + 16, fa[1], fb[1]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (dbg != -5) mfma_step(fa[0], fb[0], okbits);
      __builtin_amdgcn_sched_barrier(0);
      const int tn = tap + 1 < ntap ? tap + 1 : tap;              // (the last tap re-reads itself: no branch)
      const int noff = one ? 0 : ((tn / 3 - 1) * Wd + (tn % 3 - 1)) * LDK;
      if constexpr (dbg != -4) read_frags(A0 + noff, Bw + (size_t)tn * BN * LDK, fa[0], fb[0]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (dbg != -5) mfma_step(fa[1], fb[1], okbits);
      else { acc[0][0][0] += __uint_as_float(fa[0][0][0] ^ fa[1][1][1] ^ fb[0][0][0] ^ fb[1][BNT - 1][3] ^ fb[0][1][0] ^ fb[1][0][2] ^ fa[0][1][0] ^ fa[1][0][0]); }
      __builtin_amdgcn_sched_barrier(0);
    }
    RSTAMP(tk5)
#ifdef CTDD_RES_STAMPS
    dur[0] += tk1 - tk0; dur[1] += tk2 - tk1; dur[2] += tk3 - tk2; dur[3] += tk4 - tk3; dur[4] += tk5 - tk4;
#endif
    u = un; sgi = nsgi; c0 = nc0; kbase = nkbase;
  }
#ifdef CTDD_RES_STAMPS
  {
    unsigned long long tend;
    RSTAMP(tend)
    if (lane == 0 && a.acc_buf) {
      unsigned long long* o = (unsigned long long*)a.acc_buf + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
      for (int i = 0; i < 5; ++i) o[i] = dur[i];
      o[5] = tstart; o[6] = tend; o[7] = tend - tstart;
    }
  }
#endif

  const TileStats ts = tile_stats_begin(a, smem, p0, BMP, n0, BN, M, HW);     // (barrier: the LDS tiles are out of use)
  if (!ts.lds) __syncthreads();
  float* xt = (float*)(smem + (size_t)tile_stats_samples(BMP, HW) * BN * 16) + (size_t)wave * 32 * (BN + 4);
  conv_epilogue_rows<BNT, MT>(a, acc, xt, p0 + wave * WM, n0, M, HW, ts);
  tile_stats_flush(a, ts);
}

// ------------------------------------------------------------------ ring convolution (LDS-DMA, no staging registers)
// The slab idea of k_conv_patch with the staging taken off the waves: per 16-channel unit the
// slab (19 pieces of 32 rows x 32 B) and the weights of all nine taps (9*BNT pieces) are moved
// global -> LDS by LDS-DMA (global_load_lds, 1 KiB per wave-instruction) into a ring of NBUF unit
// buffers, one or two units ahead of the matrix instructions.  No staging registers, no ds_write,
// ONE barrier per unit; the DMA of a later unit is issued piece by piece between the taps of the
// current one.  LDS rows are 32 B (unpadded, as the DMA writes them); the two 16-B halves of row r
// are swapped when bit 3 of r is set, which makes every ds_read_b128 lane group hit 16 distinct
// bank quads for any tap offset.  8 waves x 64 rows = 512 output pixels per workgroup, 32*BNT
// output channels.
//
// The matrix pipe shares each wave's issue slots with everything else the wave executes (one
// instruction per wave every four cycles, a 32x32x16 MFMA occupying the pipe for 32), so the tap
// body is kept to ~30 instructions per six MFMAs: the nine taps of a 3x3 unit are unrolled (tap
// offsets, fragment registers and validity bits become constants), per-lane addresses are
// computed once per kernel, and a DMA op is one vector instruction on a scalar base.  Units of
// 1x1 segments run in a second, rolled loop after all 3x3 units (two loops instead of one
// branchy body keep the accumulators in place).
template <int BNT, int NBUF, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k_conv_ring(const ConvArgs a) {
  constexpr int BK = 16, WM = 64, MT = 2;
  constexpr int BN = 32 * BNT, BMP = NW * WM;
  constexpr int SP = (BMP + 2 * 34 + 31) / 32;                   // slab pieces of 32 rows >= BMP + 2*(33+1) rows
  constexpr int SLABB = SP * 1024, BUFB = SLABB + 9 * BN * 32;
  constexpr int OPS3 = (SP + 9 * BNT + NW - 1) / NW;             // DMA ops per wave for a 3x3 unit
  constexpr int OPS1 = (BMP / 32 + BNT + NW - 1) / NW;           // ... for a 1x1 unit (BMP rows, one tap)
  constexpr int DEPTH = NBUF - 1;                                // units in flight ahead of the one computed
  static_assert(DEPTH == 1 || DEPTH == 2, "ring of two or three unit buffers");
  static_assert(OPS3 <= 18, "at most two DMA ops per tap");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef CTDD_RES_STAMPS
  unsigned long long tentry;
  RSTAMP(tentry)
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, g = lane >> 5;
  const int HW = a.H * a.W, Wd = a.W;
  const int64_t M = (int64_t)a.B * HW;
  const int64_t p0 = (int64_t)blockIdx.x * BMP;
  const int n0 = blockIdx.y * BN;
  const int halo = Wd + 1;
  const int Ntot = a.N, Ktot = a.Ktot;
  unsigned vmask = 0;                                            // per-lane validity bits, filled in after the first DMA ops

  // ---- unit order of this z-slice: all 16-channel units of 3x3 segments, then those of 1x1 segments.
  // The segment table is read from the kernel arguments ONCE, into scalars (a scalar load inside the
  // unit loop costs its latency on every unit, inside the tap loop it would also drain the LDS queue).
  const unsigned char* sg_hi[3];
  int sg_C[3], sg_nu[3], sg_k[3];                                // channels, units, K offset of the segment's tap 0
  bool sg_one[3];
  int n33 = 0, n11 = 0;
  {
    int kk = 0;
#pragma unroll
    for (int sgi = 0; sgi < 3; ++sgi) {
      const bool on = sgi < a.nseg;
      sg_hi[sgi] = (const unsigned char*)a.seg[on ? sgi : 0].hi;
      sg_C[sgi] = on ? a.seg[sgi].C : 0;
      sg_one[sgi] = on && a.seg[sgi].kind == SEG_1x1;
      sg_nu[sgi] = sg_C[sgi] / BK;
      sg_k[sgi] = kk;
      kk += sg_C[sgi] * (sg_one[sgi] ? 1 : 9);
      if (sg_one[sgi]) n11 += sg_nu[sgi]; else n33 += sg_nu[sgi];
    }
  }
  const int nunits = n33 + n11;
  const int zs = blockIdx.z, nz = a.ksplit > 1 ? a.ksplit : 1;
  const int myunits = zs < nunits ? (nunits - zs + nz - 1) / nz : 0;     // sorted units zs, zs+nz, ...
  const unsigned char* const w_base = (const unsigned char*)a.w_hi;

  struct Unit { const unsigned char* src; const unsigned char* wsrc; int C; bool one; };
  auto unit_info = [&](int su) {                                 // su = index in the sorted order
    const bool one = su >= n33;
    int u = one ? su - n33 : su;
    Unit r = {sg_hi[0], w_base, 16, one};
    bool found = false;
#pragma unroll
    for (int sgi = 0; sgi < 3; ++sgi) {
      const bool mine = !found && sg_one[sgi] == one && sg_nu[sgi] > 0 && u < sg_nu[sgi];
      if (mine) {
        r.C = sg_C[sgi];
        r.src = sg_hi[sgi] + (size_t)u * BK * 2;                         // channel chunk of the segment's tensor
        r.wsrc = w_base + (size_t)(sg_k[sgi] + u * BK) * 2;              // its K offset in a weight row (tap 0)
        found = true;
      } else if (!found && sg_one[sgi] == one) {
        u -= sg_nu[sgi];
      }
    }
    return r;
  };

  // DMA op j of this wave for unit `un` into ring buffer `buf`: piece = wave + 8 j (clamped: the
  // last pieces may be issued twice, same bytes to the same place).  Lane l fills LDS bytes
  // [16 l, 16 l + 16) of the piece = row l/2, stored half l&1.  Addresses are a wave-uniform 64-bit
  // base plus one 32-bit per-lane offset.  Slab rows outside the tensor read clamped addresses:
  // they are only consumed under a cleared validity bit.  Column tiles that would start beyond N
  // re-read the last 32 weight rows: never written back.
  const int rl = lane >> 1;
  const unsigned swz16 = (unsigned)(((lane & 1) ^ ((rl >> 3) & 1)) << 4);   // byte offset of the lane's half
  const unsigned wlane = (unsigned)rl * (unsigned)Ktot * 2u + swz16;
  auto issue_op = [&](const Unit& un, unsigned slane, int buf, int j) {      // slane = rl * C * 2 + swz16 for un.C
    const int nsp = un.one ? BMP / 32 : SP;
    const int total = nsp + (un.one ? BNT : 9 * BNT);
    int p = wave + NW * j;
    p = p < total ? p : total - 1;
    const unsigned char* src;
    unsigned char* dst = smem + (size_t)buf * BUFB;
    if (p < nsp) {
      const int64_t q0 = p0 - (un.one ? 0 : halo) + p * 32;               // wave-uniform first pixel of the piece
      if (q0 >= 0 && q0 + 32 <= M) {
        src = un.src + (size_t)q0 * un.C * 2 + slane;
      } else {
        int64_t q = q0 + rl;
        q = q < 0 ? 0 : (q >= M ? M - 1 : q);
        src = un.src + (size_t)q * un.C * 2 + swz16;
      }
      dst += (size_t)p * 1024;
    } else {
      const int wp = p - nsp;                                    // = tap * BNT + t
      const int tap = wp / BNT;
      int nb = n0 + (wp - tap * BNT) * 32;
      nb = nb + 32 <= Ntot ? nb : Ntot - 32;
      src = un.wsrc + ((size_t)nb * Ktot + (size_t)tap * un.C) * 2 + wlane;
      dst += SLABB + (size_t)wp * 1024;
    }
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
  };
  auto slane_of = [&](const Unit& un) { return (unsigned)rl * (unsigned)un.C * 2u + swz16; };

  f32x16 acc[MT][BNT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < BNT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][t][i] = 0.0f;

  // Fragment reads issue B first and A last, and the A fragments are the first thing a tap touches
  // (mask_a): the wait for them covers the whole set and sits BEFORE the next set's reads issue.
  // (An LDS-DMA op makes the compiler's next LDS wait an lgkmcnt(0); in this order that is the
  // count the tap needs anyway, so nothing in flight is drained early.)
  auto read_frags = [&](unsigned aoff, unsigned boff, u32x4 (&af)[MT], u32x4 (&bf)[BNT]) {
#pragma unroll
    for (int t = 0; t < BNT; ++t) bf[t] = *(const u32x4*)(smem + boff + t * 1024);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = *(const u32x4*)(smem + aoff + mt * 1024);
  };
  auto mask_a = [&](u32x4 (&af)[MT], int bit0) {                 // validity bits bit0 + mt of vmask
#pragma unroll
    for (int mt = MT - 1; mt >= 0; --mt) {
      const unsigned m = (unsigned)__builtin_amdgcn_sbfe(vmask, bit0 + mt, 1);   // 0 or ~0
      af[mt] &= m;
    }
  };
  auto mfma_step = [&](const u32x4 (&af)[MT], const u32x4 (&bf)[BNT]) {
#pragma unroll
    for (int t = 0; t < BNT; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[mt]), __builtin_bit_cast(bf16x8, bf[t]),
                                                             acc[mt][t], 0, 0, 0);
  };

  // slab byte offset (swizzled) of this lane's A fragment under every tap, row tile 0 (x + 32 has the
  // same bit 3: row tile 1 = +1024 B); 1x1 units have no halo rows
  unsigned aoff[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int x = halo + wave * WM + li + (tap / 3 - 1) * Wd + (tap % 3 - 1);
    aoff[tap] = (unsigned)(x * 32 + ((g ^ ((x >> 3) & 1)) << 4));
  }
  const unsigned aoff1 = (unsigned)((wave * WM + li) * 32 + ((g ^ ((li >> 3) & 1)) << 4));   // (wave*64 + li: bit 3 = bit 3 of li)
  const unsigned boff = (unsigned)(SLABB + li * 32 + ((g ^ ((li >> 3) & 1)) << 4));          // weight rows: bit 3 of the row = bit 3 of li

#ifdef CTDD_RES_STAMPS
  unsigned long long tk0, tk1, tk2, tk3, tk4, dur[6] = {0, 0, 0, 0, 0, 0}, tstart;
  unsigned long long tl0 = 0, tl1 = 0, tl2 = 0, tl3 = 0, tl4 = 0, tl5 = 0, tl6 = 0, tl7 = 0;
  RSTAMP(tstart)
#endif
  // ---- prologue: the first DEPTH units of this z-slice, all ops at once
  Unit cur = {nullptr, nullptr, 16, false}, nx1 = cur, nxt = cur;        // units k, k+1 (DEPTH 2), k+DEPTH
  if (myunits > 0) {
    cur = unit_info(zs);
    const unsigned sl = slane_of(cur);
    for (int j = 0; j < (cur.one ? OPS1 : OPS3); ++j) issue_op(cur, sl, 0, j);
  }
  if (DEPTH == 2 && myunits > 1) {
    nx1 = unit_info(zs + nz);
    const unsigned sl = slane_of(nx1);
    for (int j = 0; j < (nx1.one ? OPS1 : OPS3); ++j) issue_op(nx1, sl, 1, j);
  }

  // per-lane validity of (tap, row tile): bit tap*MT + mt.  (After the first DMA ops are in flight; 32-bit arithmetic:
  // the host checks B*H*W < 2^31.)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int64_t p = p0 + wave * WM + mt * 32 + li;
    const bool in = p < M;
    const unsigned r = (unsigned)(in ? p : 0) % (unsigned)HW;
    const int oy = (int)(r / (unsigned)Wd), ox = (int)(r % (unsigned)Wd);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      const bool ok = in && (unsigned)(oy + dy - 1) < (unsigned)a.H && (unsigned)(ox + dx - 1) < (unsigned)Wd;
      vmask |= (ok ? 1u : 0u) << (tap * MT + mt);
    }
  }

  int buf = 0, k = 0;
  // top of a unit: wait for its pieces, barrier, pick the unit to fetch meanwhile
  int nbuf = 0, nops = 0;
  unsigned nsl = 0;
  auto unit_top = [&]() {
    RSTAMP(tk0)
    // unit k has landed once at most the ops of the younger unit in flight remain
    if (DEPTH == 2 && k + 1 < myunits) {
      if (nx1.one) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS3) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    RSTAMP(tk1)
    __builtin_amdgcn_s_barrier();          // everyone's pieces of unit k are in LDS; unit k-1's buffer is out of use
    RSTAMP(tk2)
    const int kn = k + DEPTH;              // the unit to start fetching now, into the buffer unit k-1 used
    nbuf = buf == 0 ? NBUF - 1 : buf - 1;
    nops = 0;
    if (kn < myunits) {
      nxt = unit_info(zs + kn * nz);
      nops = nxt.one ? OPS1 : OPS3;
      nsl = slane_of(nxt);
    }
    RSTAMP(tk4)
  };
  auto unit_bottom = [&]() {
    RSTAMP(tk3)
#ifdef CTDD_RES_STAMPS
    dur[0] += tk1 - tk0; dur[1] += tk2 - tk1; dur[3] += tk4 - tk2; dur[4] += tk3 - tk4;
    {                                       // time line of the first eight units (scalar registers only: nothing in the memory queues)
      const unsigned long long rel = tk2 - tentry;
      tl0 = k == 0 ? rel : tl0; tl1 = k == 1 ? rel : tl1; tl2 = k == 2 ? rel : tl2; tl3 = k == 3 ? rel : tl3;
      tl4 = k == 4 ? rel : tl4; tl5 = k == 5 ? rel : tl5; tl6 = k == 6 ? rel : tl6; tl7 = k == 7 ? tk0 - tentry : tl7;
    }
#endif
    if (DEPTH == 2) { cur = nx1; nx1 = nxt; } else { cur = nxt; }
    buf = buf + 1 == NBUF ? 0 : buf + 1;
    ++k;
  };

  // ---- 3x3 units: nine unrolled taps, two fragment sets alternating
  const int my33 = zs < n33 ? (n33 - zs + nz - 1) / nz : 0;              // how many of my units are 3x3
  for (; k < my33;) {
    unit_top();
    const unsigned sb = (unsigned)buf * BUFB;
    u32x4 fa[2][MT], fb[2][BNT];
    read_frags(sb + aoff[0], sb + boff, fa[0], fb[0]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // wait + mask this tap's A | read the next tap's set | one DMA op | matrix instructions
      mask_a(fa[tap & 1], tap * MT);
      __builtin_amdgcn_sched_barrier(0);
      if (tap + 1 < 9) read_frags(sb + aoff[tap + 1], sb + boff + (tap + 1) * BN * 32, fa[(tap + 1) & 1], fb[(tap + 1) & 1]);
      if (tap < OPS3 && tap < nops) issue_op(nxt, nsl, nbuf, tap);
      if (tap + 9 < OPS3 && tap + 9 < nops) issue_op(nxt, nsl, nbuf, tap + 9);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(fa[tap & 1], fb[tap & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    unit_bottom();
  }
  // ---- 1x1 units: one tap each (centre-tap validity)
  for (; k < myunits;) {
    unit_top();
    const unsigned sb = (unsigned)buf * BUFB;
    u32x4 fa[MT], fb[BNT];
    read_frags(sb + aoff1, sb + boff, fa, fb);
    for (int j = 0; j < nops; ++j) issue_op(nxt, nsl, nbuf, j);
    mask_a(fa, 4 * MT);
    mfma_step(fa, fb);
    unit_bottom();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef CTDD_RES_STAMPS
  {
    unsigned long long tend;
    RSTAMP(tend)
    if (lane == 0 && a.acc_buf) {
      unsigned long long* o = (unsigned long long*)a.acc_buf + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
      for (int i = 0; i < 5; ++i) o[i] = dur[i];
      o[2] = tstart - tentry;                  // prologue (validity bits, segment table, offsets)
      o[5] = tstart; o[6] = tend; o[7] = tend - tentry;
    }
  }
#endif

  const TileStats ts = tile_stats_begin(a, smem, p0, BMP, n0, BN, M, HW);     // (barrier: the LDS tiles are out of use)
  if (!ts.lds) __syncthreads();
  float* xt = (float*)(smem + (size_t)tile_stats_samples(BMP, HW) * BN * 16) + (size_t)wave * 32 * (BN + 4);
#ifdef CTDD_RES_STAMPS
  unsigned long long ept[3] = {0, 0, 0}, tep0, tep1;
  RSTAMP(tep0)
  conv_epilogue_rows<BNT, MT>(a, acc, xt, p0 + wave * WM, n0, M, HW, ts, ept);
  RSTAMP(tep1)
  tile_stats_flush(a, ts);
  {
    unsigned long long tdone;
    RSTAMP(tdone)
    if (lane == 0 && a.acc_buf) {
      unsigned long long* o = (unsigned long long*)a.acc_buf + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
      unsigned long long* o2 = o + (size_t)gridDim.x * gridDim.y * NW * 8;     // second table: the unit time line + the raw main-loop sums
      o2[0] = tl0; o2[1] = tl1; o2[2] = tl2; o2[3] = tl3; o2[4] = tl4; o2[5] = tl5; o2[6] = o[6] - tentry; o2[7] = tep1 - tentry;
      o2 += (size_t)gridDim.x * gridDim.y * NW * 8;                               // third: dma wait, barrier wait, prologue, unit set-up, taps
      o2[0] = o[0]; o2[1] = o[1]; o2[2] = o[2]; o2[3] = o[3]; o2[4] = o[4]; o2[5] = tentry; o2[6] = tl6; o2[7] = tl7;
      o[0] = ept[0]; o[1] = ept[1]; o[2] = ept[2];          // epilogue phases 1, 2, 3 (overwrite the dma-wait / barrier / prologue slots)
      o[3] = tep0 - o[6];                                     // tile_stats_begin (barriers + zeroing)
      o[4] = tdone - tep1;                                    // flush
      o[7] = tdone - tentry;
    }
  }
#else
  conv_epilogue_rows<BNT, MT>(a, acc, xt, p0 + wave * WM, n0, M, HW, ts);
  tile_stats_flush(a, ts);
#endif
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
// The centred input rows a workgroup needs are converted once into an LDS slab with a zero border (no bounds tests and no
// int -> float / division work in the tap loop); statistics leave a thread as fp32 sums over its <= ~8 pixels, meet in LDS
// as plain stores, and one thread per (channel, moment) adds them up in fp64: one global atomic each, no LDS atomics.
__host__ __device__ inline int first_conv_rows(int HW, int W, int gx) { return ((HW + gx - 1) / gx + W - 1) / W + 3; }
template <int CIN>
__global__ __launch_bounds__(256) void k_first_conv(const FirstConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  const int cg = a.Cout / 8, HW = a.H * a.W, b = blockIdx.y, Wp = a.W + 2;
  const int lanes = 256 / cg;                                         // pixel lanes per workgroup
  const int c8 = threadIdx.x % cg, pl = threadIdx.x / cg;
  const int per = (HW + gridDim.x - 1) / gridDim.x;
  const int p_lo = blockIdx.x * per, p_hi = min(p_lo + per, HW);
  const int nrows_max = first_conv_rows(HW, a.W, gridDim.x);
  float* slab = fsm;                                                   // [CIN][nrows_max][W + 2]
  float* part = fsm + CIN * nrows_max * Wp;                            // [lanes][Cout][2]
  if (p_lo >= p_hi) return;                                            // (uniform)
  const int y_lo = p_lo / a.W, y_hi = (p_hi - 1) / a.W, nrows = y_hi - y_lo + 3;
  for (int idx = threadIdx.x; idx < CIN * nrows * Wp; idx += 256) {
    const int ci = idx / (nrows * Wp), rem = idx % (nrows * Wp), ry = rem / Wp, xc = rem % Wp;
    const int yy = y_lo - 1 + ry, xx = xc - 1;
    float v = 0.0f;
    if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
      const size_t o = (((size_t)b * CIN + ci) * a.H + yy) * a.W + xx;
      const float raw = a.x64 ? (float)a.x64[o] : (float)a.x32[o];
      v = 2.0f * ((raw - a.lo) / (a.hi - a.lo)) - 1.0f;
      if (a.x0_f32 && yy * a.W + xx >= p_lo && yy * a.W + xx < p_hi) a.x0_f32[o] = v;
    }
    slab[(ci * nrows_max + ry) * Wp + xc] = v;
  }
  float wreg[8][9 * CIN];                                              // weights of this thread's 8 channels
  float bj[8];
  if (pl < lanes) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bj[j] = a.bias[c8 * 8 + j];
#pragma unroll
      for (int k = 0; k < 9 * CIN; ++k) wreg[j][k] = a.w[(size_t)(c8 * 8 + j) * CIN * 9 + k];
    }
  }
  __syncthreads();
  float ssum[8], ssq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ssum[j] = 0.0f; ssq[j] = 0.0f; }
  if (pl < lanes)
    for (int r = p_lo + pl; r < p_hi; r += lanes) {
      const int y = r / a.W, x = r % a.W;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = bj[j];
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const float* row = slab + (ci * nrows_max + (y - y_lo + dy)) * Wp + x;     // columns x-1 .. x+1 of the image = x .. x+2 here
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float v = row[dx];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, wreg[j][(ci * 3 + dy) * 3 + dx], acc[j]);
          }
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
      for (int j = 0; j < 8; ++j) { ssum[j] += acc[j]; ssq[j] = fmaf(acc[j], acc[j], ssq[j]); }
    }
  if (a.stats) {
    if (pl < lanes) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        part[(pl * a.Cout + c8 * 8 + j) * 2] = ssum[j];
        part[(pl * a.Cout + c8 * 8 + j) * 2 + 1] = ssq[j];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * a.Cout; i += 256) {
      double t = 0.0;
      for (int q = 0; q < lanes; ++q) t += (double)part[q * a.Cout * 2 + i];
      atomicAdd(a.stats + (size_t)b * a.Cout * 2 + i, t);
    }
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
  if (!a.s1_f32 && !a.out_f32 && nv < (int64_t)1 << 30) {
    // bf16 in / bf16 out (the throughput mode): four 16-byte vectors in flight per thread, 32-bit index arithmetic
    const unsigned n32 = (unsigned)nv, stride = gridDim.x * 256u, uvpp = (unsigned)vpp;
    if (stride % uvpp == 0u) {
      // the launcher made the thread stride a multiple of the vectors per pixel: a thread's vectors are all the SAME eight channels --
      // scale / shift sit in registers (they were four 16-byte LDS reads per vector, with 2-way bank conflicts), the source and the
      // channel offset are fixed, a step is a whole number of pixels
      using f2 = __attribute__((ext_vector_type(2))) float;
      const unsigned vfirst = blockIdx.x * 256u + threadIdx.x, px0 = vfirst / uvpp, c0 = (vfirst - px0 * uvpp) * 8u, pstep = stride / uvpp;
      const bool first = (int)c0 < a.C1;
      const unsigned Cs = first ? (unsigned)a.C1 : (unsigned)a.C2, cc = first ? c0 : c0 - (unsigned)a.C1;
      const unsigned short* src = (first ? a.s1_bf16 : a.s2_bf16) + (size_t)b * a.HW * Cs + cc;
      unsigned short* dst = a.out_hi + (size_t)b * a.HW * C + c0;
      const float4 sc0 = *(const float4*)(scale + c0), sc1 = *(const float4*)(scale + c0 + 4);
      const float4 sh0 = *(const float4*)(shift + c0), sh1 = *(const float4*)(shift + c0 + 4);
      const f2 scv[4] = {{sc0.x, sc0.y}, {sc0.z, sc0.w}, {sc1.x, sc1.y}, {sc1.z, sc1.w}};
      const f2 shv[4] = {{sh0.x, sh0.y}, {sh0.z, sh0.w}, {sh1.x, sh1.y}, {sh1.z, sh1.w}};
      const unsigned HWu = (unsigned)a.HW;
      for (unsigned px = px0; px < HWu; px += 4u * pstep) {
        uint4 u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned pk = px + (unsigned)k * pstep;
          u[k] = pk < HWu ? *(const uint4*)(src + (size_t)pk * Cs) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned pk = px + (unsigned)k * pstep;
          if (pk >= HWu) break;
          const unsigned w[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
          unsigned ow[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f2 x = {__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xFFFF0000u)};
            f2 y = __builtin_elementwise_fma(x, scv[j], shv[j]);
            if (a.swish) {
              const f2 z = y * (f2){-1.4426950408889634f, -1.4426950408889634f};
              const f2 d = (f2){__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)} + (f2){1.0f, 1.0f};
              y = y * (f2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
            }
            ow[j] = pack2_bf16(y.x, y.y);
          }
          *(uint4*)(dst + (size_t)pk * C) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
      }
      return;
    }
    for (unsigned v0 = blockIdx.x * 256u + threadIdx.x; v0 < n32; v0 += 4u * stride) {
      uint4 u[4];
      unsigned c0[4];
      size_t oo[4];
      bool ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned v = v0 + (unsigned)k * stride;
        ok[k] = v < n32;
        const unsigned px = ok[k] ? v / uvpp : 0u;
        c0[k] = (ok[k] ? v - px * uvpp : 0u) * 8u;
        const bool first = (int)c0[k] < a.C1;
        const unsigned cc = first ? c0[k] : c0[k] - (unsigned)a.C1;
        const size_t off = ((size_t)b * a.HW + px) * (first ? a.C1 : a.C2) + cc;
        oo[k] = ((size_t)b * a.HW + px) * C + c0[k];
        u[k] = ok[k] ? *(const uint4*)((first ? a.s1_bf16 : a.s2_bf16) + off) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!ok[k]) continue;
        const unsigned w[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
        // two channels per instruction where the hardware has packed fp32 forms (fma, mul, add); exp2 / rcp stay scalar
        using f2 = __attribute__((ext_vector_type(2))) float;
        const float4 sc0 = *(const float4*)(scale + c0[k]), sc1 = *(const float4*)(scale + c0[k] + 4);
        const float4 sh0 = *(const float4*)(shift + c0[k]), sh1 = *(const float4*)(shift + c0[k] + 4);
        const f2 scv[4] = {{sc0.x, sc0.y}, {sc0.z, sc0.w}, {sc1.x, sc1.y}, {sc1.z, sc1.w}};
        const f2 shv[4] = {{sh0.x, sh0.y}, {sh0.z, sh0.w}, {sh1.x, sh1.y}, {sh1.z, sh1.w}};
        unsigned ow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f2 x = {__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xFFFF0000u)};
          f2 y = __builtin_elementwise_fma(x, scv[j], shv[j]);
          if (a.swish) {                                                  // y * 1 / (1 + 2^(-y log2 e)): hardware exp2 / rcp (1 ulp)
            const f2 z = y * (f2){-1.4426950408889634f, -1.4426950408889634f};
            const f2 d = (f2){__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)} + (f2){1.0f, 1.0f};
            y = y * (f2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
          }
          ow[j] = pack2_bf16(y.x, y.y);
        }
        *(uint4*)(a.out_hi + oo[k]) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
      }
    }
    return;
  }
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
    const float4 sc0 = *(const float4*)(scale + c0), sc1 = *(const float4*)(scale + c0 + 4);
    const float4 sh0 = *(const float4*)(shift + c0), sh1 = *(const float4*)(shift + c0 + 4);
    const float scv[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
    const float shv[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      y[j] = fmaf(x[j], scv[j], shv[j]);
      if (a.swish) {
        if (a.out_f32) y[j] = y[j] / (1.0f + expf(-y[j]));                       // fp32 mode: exact
        else y[j] = y[j] * __builtin_amdgcn_rcpf(1.0f + __expf(-y[j]));           // bf16 outputs: hardware exp2 / rcp (1 ulp)
      }
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

// GroupNorm (+ Swish) of bf16 tensors in ONE pass over memory: a workgroup owns (sample, slab of whole groups), reads the slab's
// HW x slabC values once into registers, reduces their sums / sums of squares across the workgroup (fp32 per thread over <= MAXV
// pixels, fp64 across threads and group members, like the atomics of the convolution epilogues it replaces), and writes the
// normalised values from the registers.  No statistics in the producing convolutions' epilogues, no statistics buffers to
// zero, half the passes over the tensor of k_gn_apply behind a statistics epilogue.
// Thread t = (pixel lane pl = t / noct, octet oct = t % noct): 8 channels of pixels pl, pl + npl, ...: consecutive lanes read
// consecutive 16-byte pieces of a pixel's slab, then the next pixel.  The slab may straddle the two concatenated sources
// (8-channel vectors never do: C1 % 8 == 0).
template <int MAXV>
__global__ __launch_bounds__(1024) void k_gn_onepass(const GnArgs a, int slabC, int noct, int npl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  const int C = a.C1 + a.C2, cg = C / a.G, b = blockIdx.y, cs0 = blockIdx.x * slabC, HW = a.HW;
  const int t = threadIdx.x, T = noct * npl;
  constexpr int PST = 20;                                      // floats per thread record: 16 used; 80-byte records put the 16-byte stores
                                                               // of 16 neighbouring threads on all 64 banks (64-byte records: 16-way conflicts)
  float* part = (float*)gsm;                                   // [T][PST]: sums, sums of squares of the thread's 8 channels
  double* red = (double*)(gsm + (size_t)T * PST * 4);          // [2][slabC]
  float* scale = (float*)(red + 2 * slabC);                    // [slabC]
  float* shift = scale + slabC;
  const bool act = t < T;
  const int oct = act ? t % noct : 0, pl = act ? t / noct : 0;
  const int c0 = cs0 + oct * 8;
  const bool first = c0 < a.C1;
  const int Cs = first ? a.C1 : a.C2;
  const unsigned short* src = (first ? a.s1_bf16 : a.s2_bf16) + (size_t)b * HW * Cs + (first ? c0 : c0 - a.C1);
  uint4 u[MAXV];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int p = pl + k * npl;
    u[k] = (act && p < HW) ? *(const uint4*)(src + (size_t)p * Cs) : make_uint4(0, 0, 0, 0);
  }
  float sx[8], sq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sx[j] = 0.0f; sq[j] = 0.0f; }
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {                             // (absent pixels are zeros: they add nothing)
    const unsigned w[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = __uint_as_float(w[j] << 16), x1 = __uint_as_float(w[j] & 0xFFFF0000u);
      sx[2 * j] += x0; sq[2 * j] = fmaf(x0, x0, sq[2 * j]);
      sx[2 * j + 1] += x1; sq[2 * j + 1] = fmaf(x1, x1, sq[2 * j + 1]);
    }
  }
  if (act) {
    float4* pt = (float4*)(part + (size_t)t * PST);
    pt[0] = make_float4(sx[0], sx[1], sx[2], sx[3]); pt[1] = make_float4(sx[4], sx[5], sx[6], sx[7]);
    pt[2] = make_float4(sq[0], sq[1], sq[2], sq[3]); pt[3] = make_float4(sq[4], sq[5], sq[6], sq[7]);
  }
  __syncthreads();
  for (int r = t; r < 2 * slabC; r += blockDim.x) {            // (moment m, local channel cl): over the pixel lanes, in fp64
    const int m = r / slabC, cl = r - m * slabC;
    const float* pp = part + (size_t)(cl >> 3) * PST + m * 8 + (cl & 7);
    double acc = 0.0;
    for (int q = 0; q < npl; ++q) acc += (double)pp[(size_t)q * noct * PST];
    red[r] = acc;
    // training plans: the per-(sample, channel) sums also go to the sources' statistics buffers (what the producers'
    // epilogues would have accumulated), for the GroupNorm backward
    const int c = cs0 + cl;
    if (c < a.C1) { if (a.st1) ((double*)a.st1)[((size_t)b * a.C1 + c) * 2 + m] = acc; }
    else if (a.st2) ((double*)a.st2)[((size_t)b * a.C2 + (c - a.C1)) * 2 + m] = acc;
  }
  __syncthreads();
  for (int cl = t; cl < slabC; cl += blockDim.x) {
    const int c = cs0 + cl, gl0 = (c / cg) * cg - cs0;          // the group's first channel, local (slabs hold whole groups)
    double s = 0.0, q = 0.0;
    for (int j = gl0; j < gl0 + cg; ++j) { s += red[j]; q += red[slabC + j]; }
    const double n = (double)cg * (double)HW;
    const double mean = s / n;
    const double var = fmax(q / n - mean * mean, 0.0);
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    scale[cl] = rstd * a.gamma[c];
    shift[cl] = a.beta[c] - (float)mean * rstd * a.gamma[c];
  }
  __syncthreads();
  if (!act) return;
  using f2 = __attribute__((ext_vector_type(2))) float;
  const float4 sc0 = *(const float4*)(scale + oct * 8), sc1 = *(const float4*)(scale + oct * 8 + 4);
  const float4 sh0 = *(const float4*)(shift + oct * 8), sh1 = *(const float4*)(shift + oct * 8 + 4);
  const f2 scv[4] = {{sc0.x, sc0.y}, {sc0.z, sc0.w}, {sc1.x, sc1.y}, {sc1.z, sc1.w}};
  const f2 shv[4] = {{sh0.x, sh0.y}, {sh0.z, sh0.w}, {sh1.x, sh1.y}, {sh1.z, sh1.w}};
  unsigned short* dst = a.out_hi + (size_t)b * HW * C + c0;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int p = pl + k * npl;
    if (p >= HW) break;
    const unsigned w[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
    unsigned ow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f2 x = {__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xFFFF0000u)};
      f2 y = __builtin_elementwise_fma(x, scv[j], shv[j]);
      if (a.swish) {                                                      // as k_gn_apply: hardware exp2 / rcp
        const f2 z = y * (f2){-1.4426950408889634f, -1.4426950408889634f};
        const f2 d = (f2){__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)} + (f2){1.0f, 1.0f};
        y = y * (f2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
      }
      ow[j] = pack2_bf16(y.x, y.y);
    }
    *(uint4*)(dst + (size_t)p * C) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
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
  const float* w1; const float* b1; const float* w2; const float* b2;   // Linear weights transposed to [in][out]
  float* hid;   // [B][tdim] scratch = swish(W1 e + b1)
  float* act;   // [B][tdim] = swish(temb)
};
// One row-blocked linear layer serves the three steps (sinusoid -> W1 -> swish, W2 -> swish, all the
// ResBlock projections): out[b][n] = f(bias[n] + sum_i wt[i][n] x[b][i]), wt = [K][N] (coalesced over
// n).  A workgroup = 64 columns x 16 batch rows; its four waves take a quarter of K each (every weight
// is streamed once per 16 rows; one row per workgroup re-read the whole matrix from L2 per row, 1.5 GB
// per forward at batch 256) and the four partial sums are added in wave order.
constexpr int TRB = 16;
template <bool SINUSOID, bool SWISH>
__global__ __launch_bounds__(256) void k_rows_linear(const float* __restrict__ x, const float* __restrict__ wt,
                                                     const float* __restrict__ bias, int B, int K, int N,
                                                     float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [TRB][K] inputs, then [4][TRB][64] partial sums
  const int b0 = blockIdx.y * TRB;
  for (int i = threadIdx.x; i < TRB * K; i += 256) {
    const int r = i / K, c = i % K, b = b0 + r < B ? b0 + r : B - 1;
    if (SINUSOID) {                                            // x = t[B]; embedding [sin(t f) | cos(t f)], f_j = 10000^(-j/(half-1))
      const int half = K / 2, j = c < half ? c : c - half;
      const float f = expf((float)j * (-logf(10000.0f) / (float)(half - 1)));
      sm[i] = c < half ? sinf(x[b] * f) : cosf(x[b] * f);
    } else {
      sm[i] = x[(size_t)b * K + c];
    }
  }
  __syncthreads();
  const int col = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + col, nc = n < N ? n : N - 1;
  const int kq = K / 4, k0 = ks * kq;                          // K % 16 == 0
  float acc[TRB];
#pragma unroll
  for (int r = 0; r < TRB; ++r) acc[r] = 0.0f;
#pragma unroll 2
  for (int i = k0; i < k0 + kq; i += 4) {
    const float w0 = wt[(size_t)i * N + nc], w1 = wt[(size_t)(i + 1) * N + nc], w2 = wt[(size_t)(i + 2) * N + nc],
                w3 = wt[(size_t)(i + 3) * N + nc];
#pragma unroll
    for (int r = 0; r < TRB; ++r) {
      const float4 xv = *(const float4*)(sm + r * K + i);
      acc[r] = fmaf(w3, xv.w, fmaf(w2, xv.z, fmaf(w1, xv.y, fmaf(w0, xv.x, acc[r]))));
    }
  }
  __syncthreads();                                             // inputs are out of use: reuse the LDS for the partial sums
  float* part = sm;
#pragma unroll
  for (int r = 0; r < TRB; ++r) part[(ks * TRB + r) * 64 + col] = acc[r];
  __syncthreads();
  for (int i = threadIdx.x; i < TRB * 64; i += 256) {
    const int r = i >> 6, c = i & 63, nn = blockIdx.x * 64 + c;
    if (nn < N && b0 + r < B) {
      const float v = (((bias[nn] + part[(0 * TRB + r) * 64 + c]) + part[(1 * TRB + r) * 64 + c]) + part[(2 * TRB + r) * 64 + c]) +
                      part[(3 * TRB + r) * 64 + c];
      out[(size_t)(b0 + r) * N + nn] = SWISH ? v / (1.0f + expf(-v)) : v;
    }
  }
}

// Every sample at the SAME time (all the samplers: t * ones((N,)), sampling.py:119-121): the time path for ONE row, two
// launches.  k_time_uniform_act: workgroup g computes columns 64 g .. 64 g + 63 of act = swish(W2 swish(W1 e + b1) + b2); each
// recomputes the first hidden layer (ch x tdim weights, 147 KB at MNIST size) rather than wait on a third launch.
// k_time_uniform_proj: workgroup g computes 64 columns of all the ResBlock projections.  In both, the four waves split the
// contraction index and meet in LDS, the loads of a wave's quarter are independent (unrolled 8 deep: the row-blocked
// three-launch path above is a chain of L2 latencies, ~80 us at the head of every forward; this is ~10).
__device__ inline float time_quarter_dot(const float* __restrict__ w, const float* __restrict__ src, int K, int N, int n, int ks) {
  const int kq = K / 4, k0 = ks * kq;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll 2
  for (int i = k0; i < k0 + kq; i += 4) {
    s0 = fmaf(w[(size_t)i * N + n], src[i], s0); s1 = fmaf(w[(size_t)(i + 1) * N + n], src[i + 1], s1);
    s2 = fmaf(w[(size_t)(i + 2) * N + n], src[i + 2], s2); s3 = fmaf(w[(size_t)(i + 3) * N + n], src[i + 3], s3);
  }
  return (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void k_time_uniform_act(const TimeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // e[ch] | hid[tdim] | part[4][64]
  float* e = sm;
  float* hid = e + a.ch;
  float* part = hid + a.tdim;
  const float t = a.t[0];
  const int half = a.ch / 2;
  for (int c = threadIdx.x; c < a.ch; c += 256) {
    const int j = c < half ? c : c - half;
    const float f = expf((float)j * (-logf(10000.0f) / (float)(half - 1)));
    e[c] = c < half ? sinf(t * f) : cosf(t * f);
  }
  __syncthreads();
  const int col = threadIdx.x & 63, ks = threadIdx.x >> 6;
  for (int g0 = 0; g0 < a.tdim; g0 += 64) {                    // first hidden layer, all columns (ch / 4 loads per thread and group)
    part[ks * 64 + col] = time_quarter_dot(a.w1, e, a.ch, a.tdim, g0 + col, ks);
    __syncthreads();
    if (threadIdx.x < 64) {
      const float v = ((a.b1[g0 + col] + part[col]) + part[64 + col]) + (part[128 + col] + part[192 + col]);
      hid[g0 + col] = v / (1.0f + expf(-v));
    }
    __syncthreads();
  }
  const int n = blockIdx.x * 64 + col;
  part[ks * 64 + col] = time_quarter_dot(a.w2, hid, a.tdim, a.tdim, n, ks);
  __syncthreads();
  if (threadIdx.x < 64) {
    const float v = ((a.b2[n] + part[col]) + part[64 + col]) + (part[128 + col] + part[192 + col]);
    a.act[n] = v / (1.0f + expf(-v));
  }
}
__global__ __launch_bounds__(256) void k_time_uniform_proj(const float* __restrict__ act, int tdim, const float* __restrict__ proj_w,
                                                           const float* __restrict__ proj_b, int Ntot, float* __restrict__ proj_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // act[tdim] | part[4][64]
  float* av = sm;
  float* part = av + tdim;
  for (int i = threadIdx.x; i < tdim; i += 256) av[i] = act[i];
  __syncthreads();
  const int col = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + col, nc = n < Ntot ? n : Ntot - 1;
  part[ks * 64 + col] = time_quarter_dot(proj_w, av, tdim, Ntot, nc, ks);
  __syncthreads();
  if (threadIdx.x < 64 && n < Ntot)
    proj_out[n] = ((proj_b[n] + part[col]) + part[64 + col]) + (part[128 + col] + part[192 + col]);
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
  // softmax rows: four neighbouring lanes per row (one thread per row left 200 of the 256 threads idle through three serial
  // passes of T exps), the normalisation folded into the output pass below
  float* rz = w + T * T;                                        // [T] 1 / row sum
  for (int t4 = threadIdx.x; t4 < ((T + 63) / 64) * 256; t4 += 256) {
    const int t = t4 >> 2, l4 = t4 & 3;
    float m = -INFINITY;
    if (t < T)
      for (int s = l4; s < T; s += 4) m = fmaxf(m, w[t * T + s]);
    m = fmaxf(m, __shfl_xor(m, 1, WAVE));
    m = fmaxf(m, __shfl_xor(m, 2, WAVE));
    float z = 0.0f;
    if (t < T)
      for (int s = l4; s < T; s += 4) { const float e = expf(w[t * T + s] - m); w[t * T + s] = e; z += e; }
    z += __shfl_xor(z, 1, WAVE);
    z += __shfl_xor(z, 2, WAVE);
    if (t < T && l4 == 0) rz[t] = 1.0f / z;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T * ch; i += 256) {
    const int t = i / ch, c = i % ch;
    float o = 0.0f;
    for (int s = 0; s < T; ++s) o = fmaf(w[t * T + s], v[s * ch + c], o);
    o *= rz[t];
    const size_t oo = ((size_t)b * T + t) * a.C + hd * ch + c;
    if (a.out_hi) a.out_hi[oo] = to_bf16(o);
    if (a.out_f32) a.out_f32[oo] = o;
  }
}

// The same with 4 x 4 register tiles: both products were LDS-bound with two 4-byte reads per multiply-add -- and the k rows of
// the score product, read at a stride of ch floats by neighbouring lanes, all fell into one bank (ch = 192 = 3 x 64: a 64-way
// conflict): 46 us per 128 samples at 7x7, C = 192.  Here q and k sit TRANSPOSED in LDS ([channel][token], rows padded to a
// multiple of four tokens), a thread owns a 4-token x 4-token block of the scores (one 16-byte read of each operand per 16
// multiply-adds, neighbouring lanes read neighbouring pieces), the scores are kept token-major-transposed (wT[s][t]) so that the
// output product reads them the same way against 16-byte pieces of v.  Same summation orders as k_attn_small (c ascending, s ascending,
// the 4-lane softmax split): identical results.
__global__ __launch_bounds__(256) void k_attn_small_t4(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x / a.heads, hd = blockIdx.x % a.heads, ch = a.C / a.heads, T = a.T, Tp = (T + 3) & ~3;
  float* qT = sm; float* kT = qT + ch * Tp; float* v = kT + ch * Tp; float* wT = v + T * ch; float* rz = wT + T * Tp;
  const float sc = 1.0f / sqrtf(sqrtf((float)ch));
  for (int i = threadIdx.x; i < ch * (Tp - T); i += 256) {            // pad tokens: zeros (their score columns are never used)
    const int c = i / (Tp - T), t = T + i % (Tp - T);
    qT[c * Tp + t] = 0.0f; kT[c * Tp + t] = 0.0f;
  }
  for (int i = threadIdx.x; i < T * ch; i += 256) {
    const int t = i / ch, c = i % ch;
    const float* src = a.qkv + ((size_t)b * T + t) * 3 * a.C + hd * 3 * ch;
    qT[c * Tp + t] = src[c] * sc; kT[c * Tp + t] = src[ch + c] * sc; v[i] = src[2 * ch + c];
  }
  __syncthreads();
  const int nbt = Tp / 4;
  for (int blk = threadIdx.x; blk < nbt * nbt; blk += 256) {
    const int tb = blk / nbt, sb = blk % nbt;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
    for (int c = 0; c < ch; ++c) {
      const float4 qa = *(const float4*)(qT + c * Tp + 4 * tb), kb = *(const float4*)(kT + c * Tp + 4 * sb);
      const float qv[4] = {qa.x, qa.y, qa.z, qa.w}, kv[4] = {kb.x, kb.y, kb.z, kb.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(qv[i], kv[j], acc[i][j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * sb + j < T) *(float4*)(wT + (4 * sb + j) * Tp + 4 * tb) = make_float4(acc[0][j], acc[1][j], acc[2][j], acc[3][j]);
  }
  __syncthreads();
  for (int t4 = threadIdx.x; t4 < ((T + 63) / 64) * 256; t4 += 256) {          // softmax over s for token t: four lanes per token
    const int t = t4 >> 2, l4 = t4 & 3;
    float m = -INFINITY;
    if (t < T)
      for (int s_ = l4; s_ < T; s_ += 4) m = fmaxf(m, wT[s_ * Tp + t]);
    m = fmaxf(m, __shfl_xor(m, 1, WAVE));
    m = fmaxf(m, __shfl_xor(m, 2, WAVE));
    float z = 0.0f;
    if (t < T)
      for (int s_ = l4; s_ < T; s_ += 4) { const float e = expf(wT[s_ * Tp + t] - m); wT[s_ * Tp + t] = e; z += e; }
    z += __shfl_xor(z, 1, WAVE);
    z += __shfl_xor(z, 2, WAVE);
    if (t < T && l4 == 0) rz[t] = 1.0f / z;
  }
  __syncthreads();
  const int nbc = ch / 4;
  for (int blk = threadIdx.x; blk < nbt * nbc; blk += 256) {
    const int tb = blk / nbc, cb = blk % nbc;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
    for (int s_ = 0; s_ < T; ++s_) {
      const float4 wa = *(const float4*)(wT + s_ * Tp + 4 * tb), vb = *(const float4*)(v + s_ * ch + 4 * cb);
      const float wv[4] = {wa.x, wa.y, wa.z, wa.w}, vv[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(wv[i], vv[j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = 4 * tb + i;
      if (t >= T) break;
      const float r = rz[t];
      const float o0 = acc[i][0] * r, o1 = acc[i][1] * r, o2 = acc[i][2] * r, o3 = acc[i][3] * r;
      const size_t oo = ((size_t)b * T + t) * a.C + hd * ch + 4 * cb;
      if (a.out_hi) *(uint2*)(a.out_hi + oo) = make_uint2((unsigned)to_bf16(o0) | ((unsigned)to_bf16(o1) << 16), (unsigned)to_bf16(o2) | ((unsigned)to_bf16(o3) << 16));
      if (a.out_f32) *(float4*)(a.out_f32 + oo) = make_float4(o0, o1, o2, o3);
    }
  }
}

// ------------------------------------------------------------------ logistic head (models.py:249-283)
// mu = tanh(loc + x0), logits[s] = log(sigmoid(r) - sigmoid(l)) via log_minus_exp, straight into (B,D,S)
struct LogisticArgs { const float* net; const float* x0; int B, C, HW, S, fix; float* out; int fast;
                      unsigned short* out_bf16; };      // (fast mode, S % 4 == 0) the logits as bf16 instead: what the bf16 step kernel reads
__device__ inline float logsigmoidf(float x) { return fminf(x, 0.0f) - log1pf(expf(-fabsf(x))); }
// hardware exp2 / log2 forms for the bf16 engine mode (~1e-6 relative; the mode's logits carry ~1e-2 already)
__device__ inline float logsigmoid_fast(float x) { return fminf(x, 0.0f) - __logf(1.0f + __expf(-fabsf(x))); }
// One wave per (b, c, pixel) row: mu and the inverse scale are computed once per row (the first version recomputed
// tanh and exp for each of the S bins), lanes stride over the bins.
__global__ __launch_bounds__(256) void k_logistic_head(const LogisticArgs a) {
  // net: NHWC [B][HW][2C] (loc channels 0..C-1, log_scale C..2C-1); x0: (B,C,HW) centred input
  const int lane = threadIdx.x & 63;
  // (a wave walks rows with the grid's stride: one row per wave and launch -- 24 576 workgroups of a few hundred instructions at
  //  CIFAR batch 32 -- spent its time in workgroup dispatch: 60 us for 50 MB)
  const int64_t nrows = (int64_t)a.B * a.C * a.HW;
  for (int64_t d = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); d < nrows; d += (int64_t)gridDim.x * 4) {
  const int p = (int)(d % a.HW), c = (int)((d / a.HW) % a.C), b = (int)(d / ((int64_t)a.HW * a.C));
  const float* nr = a.net + ((size_t)b * a.HW + p) * 2 * a.C;
  const float mu = tanhf(nr[c] + a.x0[((size_t)b * a.C + c) * a.HW + p]);
  const float inv_scale = expf(-(nr[a.C + c] - 2.0f));
  const float bw = 2.0f / (float)a.S, stepc = (2.0f - bw) / (float)(a.S - 1);
  if (a.out_bf16) {
    // bf16 logits (the sampler loops of the bf16 engine): a lane computes four consecutive bins and stores them as 8 bytes
    unsigned short* ob = a.out_bf16 + (size_t)d * a.S;
    for (int s0 = 4 * lane; s0 < a.S; s0 += 256) {
      float v4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float centre = -1.0f + bw * 0.5f + (float)(s0 + e) * stepc;
        const float l = (centre - bw * 0.5f - mu) * inv_scale, r = (centre + bw * 0.5f - mu) * inv_scale;
        const float cl = logsigmoid_fast(l), cr = logsigmoid_fast(r);
        float v = cr + __logf(1.0f - __expf(cl - cr) + 1e-6f);
        if (a.fix) {
          const float a2 = -l + cl, b2 = -r + cr;
          v = fminf(v, a2 + __logf(1.0f - __expf(b2 - a2) + 1e-6f));
        }
        v4[e] = v;
      }
      *(uint2*)(ob + s0) = make_uint2(pack2_bf16(v4[0], v4[1]), pack2_bf16(v4[2], v4[3]));
    }
    continue;
  }
  float* out = a.out + (size_t)d * a.S;
  for (int s = lane; s < a.S; s += 64) {
    const float centre = -1.0f + bw * 0.5f + (float)s * stepc;            // torch.linspace
    const float l = (centre - bw * 0.5f - mu) * inv_scale, r = (centre + bw * 0.5f - mu) * inv_scale;
    float v;
    if (a.fast) {
      const float cl = logsigmoid_fast(l), cr = logsigmoid_fast(r);
      v = cr + __logf(1.0f - __expf(cl - cr) + 1e-6f);
      if (a.fix) {
        const float a2 = -l + cl, b2 = -r + cr;
        v = fminf(v, a2 + __logf(1.0f - __expf(b2 - a2) + 1e-6f));
      }
    } else {
      const float cl = logsigmoidf(l), cr = logsigmoidf(r);
      v = cr + log1pf(-expf(cl - cr) + 1e-6f);
      if (a.fix) {
        const float a2 = -l + cl, b2 = -r + cr;
        v = fminf(v, a2 + log1pf(-expf(b2 - a2) + 1e-6f));
      }
    }
    out[s] = v;
  }
  }
}

}  // namespace ctdd
using namespace ctdd;

// ============================================================================ C ABI
template <int BK, int BNT, bool F32>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
  constexpr int ESZ = F32 ? 4 : 2;
  constexpr int LDK = BK + 16 / ESZ;
  size_t lds = (size_t)(BM + 32 * BNT) * LDK * ESZ;
  const size_t stats_lds = (size_t)tile_stats_samples(BM, a.H * a.W) * 32 * BNT * 16;
  if (lds < stats_lds) lds = stats_lds;
  const int64_t M = (int64_t)a.B * a.H * a.W;
  dim3 g((unsigned)((M + BM - 1) / BM), (unsigned)((a.N + 32 * BNT - 1) / (32 * BNT)));
  static bool attr_done[16] = {};
  ensure_lds_ceiling((const void*)k_conv_igemm<BK, BNT, F32>, attr_done);
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

template <int BK, int BNT, int WM, bool EXT = false, bool ALLTAPS = false, bool DIRECT = false>
static int launch_patch(const ConvArgs& a, hipStream_t st) {
  constexpr int LDK = BK + 8;
  const int PR = 4 * WM + 2 * (a.W + 1);
  size_t lds = ((size_t)PR + (ALLTAPS ? 9 : 2) * 32 * BNT) * LDK * 2;
  const size_t epi_lds = epilogue_rows_lds(4, BNT, 4 * WM, a.H * a.W);
  if (lds < epi_lds) lds = epi_lds;
  const int64_t M = (int64_t)a.B * a.H * a.W;
  dim3 g((unsigned)((M + 4 * WM - 1) / (4 * WM)), (unsigned)((a.N + 32 * BNT - 1) / (32 * BNT)), a.ksplit > 1 ? a.ksplit : 1);
  static bool attr_done[16] = {};
  ensure_lds_ceiling((const void*)k_conv_patch<BK, BNT, WM, EXT, ALLTAPS, DIRECT>, attr_done);
  hipLaunchKernelGGL((k_conv_patch<BK, BNT, WM, EXT, ALLTAPS, DIRECT>), g, dim3(256), lds, st, a);
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
  CTDD_REQUIRE(a.N % 8 == 0 && ((int64_t)a.H * a.W >= 32 || a.H * a.W == 16 || a.B == 1), CTDD_ERANGE,
               "patch conv: N=%d H*W=%d (a 32-row slice may span two samples at most)", a.N, a.H * a.W);
  CTDD_REQUIRE(a.logits_C <= 0 || (a.N % a.logits_C == 0 && (a.N / a.logits_C) % 8 == 0), CTDD_EINVAL, "logits layout needs S %% 8 == 0");
  for (int i = 0; i < a.nseg; ++i) {
    CTDD_REQUIRE(a.seg[i].hi && a.seg[i].C % bk == 0, CTDD_EINVAL, "segment %d: C=%d vs BK=%d", i, a.seg[i].C, bk);
    CTDD_REQUIRE(a.seg[i].kind == SEG_3x3 || a.seg[i].kind == SEG_1x1, CTDD_EINVAL, "segment %d: kind %d", i, a.seg[i].kind);
  }
  CTDD_REQUIRE(a.ksplit <= 1 || (a.acc_buf && a.logits_C == 0), CTDD_EINVAL, "split-K needs acc_buf");
  hipStream_t st = (hipStream_t)stream;
  if (a.act != 0 || a.out_lo) {                 // plain-GEMM use (hollow transformer linears): activation / hi + lo outputs
#define CASEX(BK_, BNT_) if (bk == BK_ && bnt == BNT_ && wm == 32) return launch_patch<BK_, BNT_, 32, true>(a, st);
    CASEX(64, 4) CASEX(64, 2) CASEX(64, 1) CASEX(48, 4) CASEX(48, 3) CASEX(48, 2) CASEX(48, 1) CASEX(32, 4) CASEX(32, 3) CASEX(32, 1) CASEX(16, 1)
#undef CASEX
    CTDD_REQUIRE(false, CTDD_ERANGE, "no patch-conv instantiation with activation / split output for BK=%d BNT=%d WM=%d (wm must be 32)", bk, bnt, wm);
  }
  // bf16 output without statistics / logits layout / split-K / fp32 copies: the direct (transposed-accumulator) epilogue
  static const bool no_direct = [] { const char* e = getenv("CTDD_CONV_NO_DIRECT"); return e && e[0] == '1'; }();
  const bool direct = !no_direct && !a.stats && a.logits_C <= 0 && a.ksplit <= 1 && a.out_hi && !a.out_f32 && !a.res_f32 &&
                      (!a.bias || ((uintptr_t)a.bias & 15) == 0) && (!a.tbias || (((uintptr_t)a.tbias & 15) == 0 && a.tb_stride % 4 == 0));
  if (direct) {
    if (bnt == 1 && wm == 32 && bk == 64) return launch_patch<64, 1, 32, false, true, true>(a, st);
    if (bnt == 1 && wm == 32 && bk == 48) return launch_patch<48, 1, 32, false, true, true>(a, st);
    if (bnt == 3 && wm == 64 && bk == 48) return launch_patch<48, 3, 64, false, false, true>(a, st);
    if (bnt == 3 && wm == 32 && bk == 48) return launch_patch<48, 3, 32, false, false, true>(a, st);
  }
  if (bnt == 1 && wm == 32 && a.ksplit <= 1) {  // 32-column tiles (the small levels): every tap's weight tile staged per chunk
    if (bk == 64) return launch_patch<64, 1, 32, false, true>(a, st);
    if (bk == 48) return launch_patch<48, 1, 32, false, true>(a, st);
  }
#define CASEP(BK_, BNT_, WM_) if (bk == BK_ && bnt == BNT_ && wm == WM_) return launch_patch<BK_, BNT_, WM_>(a, st);
  CASEP(48, 3, 64) CASEP(48, 3, 32) CASEP(48, 4, 64) CASEP(48, 4, 32)
  CASEP(64, 4, 64) CASEP(64, 4, 32) CASEP(64, 2, 64) CASEP(64, 2, 32) CASEP(32, 1, 32) CASEP(32, 3, 32) CASEP(32, 4, 32)
  CASEP(16, 1, 32) CASEP(48, 1, 32) CASEP(48, 2, 32) CASEP(64, 1, 32) CASEP(32, 2, 32) CASEP(48, 2, 64)
#undef CASEP
  CTDD_REQUIRE(false, CTDD_ERANGE, "no patch-conv instantiation for BK=%d BNT=%d WM=%d", bk, bnt, wm);
}

template <int BNT>
static int launch_res(const ConvArgs& a, hipStream_t st) {
  const int64_t M = (int64_t)a.B * a.H * a.W;
  const int nz = a.ksplit > 1 ? a.ksplit : 1;
  dim3 g((unsigned)((M + 511) / 512), (unsigned)((a.N + 32 * BNT - 1) / (32 * BNT)), (unsigned)nz);
  size_t lds = ((size_t)(512 + 2 * (a.W + 1)) + (size_t)9 * 32 * BNT) * 40 * 2;
  const size_t epi_lds = epilogue_rows_lds(8, BNT, 512, a.H * a.W);
  if (lds < epi_lds) lds = epi_lds;
  CTDD_REQUIRE(lds <= 160 * 1024, CTDD_ERANGE, "resident conv: %zu bytes of LDS", lds);
  static bool attr_done[16] = {};
  ensure_lds_ceiling((const void*)k_conv_res<BNT>, attr_done);
  hipLaunchKernelGGL((k_conv_res<BNT>), g, dim3(512), lds, st, a);
  if (int rc = finish_launch("k_conv_res")) return rc;
  if (nz > 1) {
    hipLaunchKernelGGL(k_conv_finish, dim3((a.N + 31) / 32, a.B), dim3(256), 0, st, a);
    return finish_launch("k_conv_finish");
  }
  return CTDD_OK;
}

extern "C" int ctdd_unet_conv_res(const void* args_, int bnt, void* stream) {
  CTDD_REQUIRE(args_ != nullptr, CTDD_EINVAL, "null args");
  const ConvArgs& a = *(const ConvArgs*)args_;
  CTDD_REQUIRE(a.nseg >= 1 && a.nseg <= 3 && a.w_hi && a.W <= 33, CTDD_EINVAL, "resident conv: nseg=%d W=%d", a.nseg, a.W);
  CTDD_REQUIRE(a.H == a.Hin && a.W == a.Win, CTDD_EINVAL, "resident conv needs stride 1");
  CTDD_REQUIRE(a.N % 8 == 0, CTDD_ERANGE, "resident conv: N=%d", a.N);
  CTDD_REQUIRE(a.logits_C <= 0 || (a.N % a.logits_C == 0 && (a.N / a.logits_C) % 8 == 0), CTDD_EINVAL, "logits layout needs S %% 8 == 0");
  CTDD_REQUIRE((int64_t)a.H * a.W >= 32 || a.H * a.W == 16 || a.B == 1, CTDD_ERANGE, "resident conv needs H*W >= 32 (or 16)");
  int k = 0;
  for (int i = 0; i < a.nseg; ++i) {
    CTDD_REQUIRE(a.seg[i].hi && (a.seg[i].kind == SEG_3x3 || a.seg[i].kind == SEG_1x1) && a.seg[i].C % 32 == 0, CTDD_EINVAL,
                 "resident conv segment %d: kind=%d C=%d", i, a.seg[i].kind, a.seg[i].C);
    k += a.seg[i].C * (a.seg[i].kind == SEG_1x1 ? 1 : 9);
  }
  CTDD_REQUIRE(k == a.Ktot, CTDD_EINVAL, "Ktot=%d but segments sum to %d", a.Ktot, k);
  CTDD_REQUIRE(a.ksplit <= 1 || (a.acc_buf && a.logits_C == 0), CTDD_EINVAL, "split-K needs acc_buf");
  hipStream_t st = (hipStream_t)stream;
  switch (bnt) {
    case 2: return launch_res<2>(a, st);
    case 3: return launch_res<3>(a, st);
    case 4: return launch_res<4>(a, st);
  }
  CTDD_REQUIRE(false, CTDD_ERANGE, "unsupported resident conv tile bnt=%d", bnt);
}

template <int BNT, int NBUF, int NW>
static int launch_ring(const ConvArgs& a, hipStream_t st) {
  constexpr int BMP = NW * 64, SP = (BMP + 2 * 34 + 31) / 32;
  const int64_t M = (int64_t)a.B * a.H * a.W;
  const int nz = a.ksplit > 1 ? a.ksplit : 1;
  dim3 g((unsigned)((M + BMP - 1) / BMP), (unsigned)((a.N + 32 * BNT - 1) / (32 * BNT)), (unsigned)nz);
  size_t lds = (size_t)NBUF * (SP * 1024 + 9 * 32 * BNT * 32);
  const size_t epi_lds = epilogue_rows_lds(NW, BNT, BMP, a.H * a.W);
  if (lds < epi_lds) lds = epi_lds;
  CTDD_REQUIRE(lds <= 160 * 1024, CTDD_ERANGE, "ring conv: %zu bytes of LDS", lds);
  static bool attr_done[16] = {};
  ensure_lds_ceiling((const void*)k_conv_ring<BNT, NBUF, NW>, attr_done);
  hipLaunchKernelGGL((k_conv_ring<BNT, NBUF, NW>), g, dim3(NW * 64), lds, st, a);
  if (int rc = finish_launch("k_conv_ring")) return rc;
  if (nz > 1) {
    hipLaunchKernelGGL(k_conv_finish, dim3((a.N + 31) / 32, a.B), dim3(256), 0, st, a);
    return finish_launch("k_conv_finish");
  }
  return CTDD_OK;
}

extern "C" int ctdd_unet_conv_ring(const void* args_, int bnt, void* stream) {
  CTDD_REQUIRE(args_ != nullptr, CTDD_EINVAL, "null args");
  const ConvArgs& a = *(const ConvArgs*)args_;
  CTDD_REQUIRE(a.nseg >= 1 && a.nseg <= 3 && a.w_hi && a.W <= 33, CTDD_EINVAL, "ring conv: nseg=%d W=%d", a.nseg, a.W);
  CTDD_REQUIRE(a.H == a.Hin && a.W == a.Win, CTDD_EINVAL, "ring conv needs stride 1");
  CTDD_REQUIRE(a.logits_C <= 0 || (a.N % a.logits_C == 0 && (a.N / a.logits_C) % 8 == 0), CTDD_EINVAL, "logits layout needs S %% 8 == 0");
  CTDD_REQUIRE((int64_t)a.H * a.W >= 32 || a.H * a.W == 16 || a.B == 1, CTDD_ERANGE, "ring conv needs H*W >= 32 (or 16)");
  int k = 0;
  for (int i = 0; i < a.nseg; ++i) {
    CTDD_REQUIRE(a.seg[i].hi && (a.seg[i].kind == SEG_3x3 || a.seg[i].kind == SEG_1x1) && a.seg[i].C % 16 == 0, CTDD_EINVAL,
                 "ring conv segment %d: kind=%d C=%d", i, a.seg[i].kind, a.seg[i].C);
    k += a.seg[i].C * (a.seg[i].kind == SEG_1x1 ? 1 : 9);
  }
  CTDD_REQUIRE(k == a.Ktot, CTDD_EINVAL, "Ktot=%d but segments sum to %d", a.Ktot, k);
  CTDD_REQUIRE(a.Ktot % 8 == 0 && a.N % 32 == 0 && a.N >= 32, CTDD_EINVAL, "ring conv: Ktot=%d N=%d (need Ktot %% 8 == 0, N %% 32 == 0)", a.Ktot, a.N);
  CTDD_REQUIRE((int64_t)a.Ktot * 64 < (int64_t)1 << 31, CTDD_ERANGE, "ring conv: Ktot=%d too large for 32-bit lane offsets", a.Ktot);
  CTDD_REQUIRE((int64_t)a.B * a.H * a.W < (int64_t)1 << 31, CTDD_ERANGE, "ring conv: B*H*W does not fit 31 bits");
  CTDD_REQUIRE(a.ksplit <= 1 || (a.acc_buf && a.logits_C == 0), CTDD_EINVAL, "split-K needs acc_buf");
  hipStream_t st = (hipStream_t)stream;
  switch (bnt) {                            // bnt + 10: 256-pixel tiles, four waves, two workgroups per CU, ring of two
    case 2: return launch_ring<2, 3, 8>(a, st);
    case 3: return launch_ring<3, 3, 8>(a, st);
    case 4: return launch_ring<4, 2, 8>(a, st);
    case 12: return launch_ring<2, 2, 4>(a, st);
    case 13: return launch_ring<3, 2, 4>(a, st);
  }
  CTDD_REQUIRE(false, CTDD_ERANGE, "unsupported ring conv tile bnt=%d", bnt);
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
  CTDD_REQUIRE(a.Cin >= 1 && a.Cin <= 4 && a.Cout / 8 <= 256, CTDD_ERANGE, "first conv: Cin=%d Cout=%d", a.Cin, a.Cout);
  const int lanes = 256 / (a.Cout / 8);
  int gx = (a.H * a.W + 8 * lanes - 1) / (8 * lanes);     // ~8 pixels per thread
  if (gx < 1) gx = 1;
  const size_t lds = ((size_t)a.Cin * first_conv_rows(a.H * a.W, a.W, gx) * (a.W + 2) + (size_t)lanes * a.Cout * 2) * sizeof(float);
  CTDD_REQUIRE(lds <= 64 * 1024, CTDD_ERANGE, "first conv: %zu bytes of LDS", lds);
  hipStream_t st = (hipStream_t)stream;
  switch (a.Cin) {
    case 1: hipLaunchKernelGGL(k_first_conv<1>, dim3(gx, a.B), dim3(256), lds, st, a); break;
    case 2: hipLaunchKernelGGL(k_first_conv<2>, dim3(gx, a.B), dim3(256), lds, st, a); break;
    case 3: hipLaunchKernelGGL(k_first_conv<3>, dim3(gx, a.B), dim3(256), lds, st, a); break;
    default: hipLaunchKernelGGL(k_first_conv<4>, dim3(gx, a.B), dim3(256), lds, st, a); break;
  }
  return finish_launch("k_first_conv");
}

extern "C" int ctdd_unet_gn_apply(const void* args_, void* stream) {
  const GnArgs& a = *(const GnArgs*)args_;
  const int C = a.C1 + a.C2;
  CTDD_REQUIRE(C % 8 == 0 && a.C1 % 8 == 0 && C % a.G == 0 && (a.out_hi || a.out_f32), CTDD_EINVAL, "bad GroupNorm arguments");
  const int64_t nv = (int64_t)a.HW * (C / 8);
  int gx = (int)((nv + 2047) / 2048);      // >= 8 vectors per thread: the per-workgroup scale/shift prologue (fp64 divide + sqrt) stays small
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  {
    // a thread stride (gx * 256 vectors) that is a multiple of the C / 8 vectors per pixel lets a thread keep its channels' scale /
    // shift in registers (k_gn_apply's bf16 path): round gx up to the next such count when that costs at most half again as many
    // workgroups
    const int vpp = C / 8;
    int g = vpp, r = 256;
    while (r) { const int t = g % r; g = r; r = t; }                 // gcd(vpp, 256)
    const int unit = vpp / g, gu = ((gx + unit - 1) / unit) * unit;
    if (gu <= gx + (gx + 1) / 2 + 1 && gu <= 96) gx = gu;
  }
  hipLaunchKernelGGL(k_gn_apply, dim3(gx, a.B), dim3(256), (size_t)2 * C * sizeof(float), (hipStream_t)stream, a);
  return finish_launch("k_gn_apply");
}

// slabC: channels per workgroup (a multiple of lcm(C / G, 8) that divides C); 0: chosen here.  max_threads: workgroup size limit
// (0: 1024).  Returns CTDD_ERANGE when no slab fits the registers (HW * slabC / 8 vectors over <= max_threads threads, <= 12
// each): the caller then keeps k_gn_apply behind statistics.
extern "C" int ctdd_unet_gn_onepass(const void* args_, int slabC, int max_threads, void* stream) {
  if (max_threads <= 0 || max_threads > 1024) max_threads = 1024;
  const GnArgs& a = *(const GnArgs*)args_;
  const int C = a.C1 + a.C2;
  CTDD_REQUIRE(C % 8 == 0 && a.C1 % 8 == 0 && a.G > 0 && C % a.G == 0 && a.out_hi && a.s1_bf16 && (a.C2 == 0 || a.s2_bf16) && a.HW > 0 && a.B > 0,
               CTDD_EINVAL, "bad one-pass GroupNorm arguments (bf16 sources and output, C %% 8 == 0)");
  const int cg = C / a.G;
  int L = cg;
  while (L % 8) L += cg;                                         // lcm(cg, 8)
  auto shape = [&](int sc, int& noct, int& npl, int& nvec) {
    noct = sc / 8;
    if (noct > max_threads) return false;
    npl = max_threads / noct < a.HW ? max_threads / noct : a.HW;
    nvec = (a.HW + npl - 1) / npl;
    return nvec <= 12;
  };
  int noct = 0, npl = 0, nvec = 0;
  if (slabC <= 0) {
    // the largest slab that still gives >= 256 workgroups; failing that, the smallest that fits
    int best = 0;
    for (int sc = L; sc <= C; sc += L) {
      if (C % sc) continue;
      int o, p_, v;
      if (!shape(sc, o, p_, v)) continue;
      const long wgs = (long)a.B * (C / sc);
      if (best == 0 || wgs >= 256) best = sc;
      if (wgs < 256) break;
    }
    slabC = best;
  }
  CTDD_REQUIRE(slabC > 0 && slabC % L == 0 && C % slabC == 0 && shape(slabC, noct, npl, nvec), CTDD_ERANGE,
               "one-pass GroupNorm: no slab of whole groups fits (HW=%d C=%d G=%d slab=%d)", a.HW, C, a.G, slabC);
  int threads = ((noct * npl + 63) / 64) * 64;
  const int need = 2 * slabC;                                    // the reduction wants a thread per (moment, channel) at best
  if (threads < 256) threads = 256;
  (void)need;
  const size_t lds = (size_t)noct * npl * 80 + (size_t)2 * slabC * 8 + (size_t)2 * slabC * 4;     // (80-byte thread records: k_gn_onepass PST)
  const dim3 g((unsigned)(C / slabC), (unsigned)a.B);
  hipStream_t st = (hipStream_t)stream;
  static bool attr_done[4][16] = {};
  auto go = [&](auto kernel, int slot) {
    ensure_lds_ceiling((const void*)kernel, attr_done[slot]);
    hipLaunchKernelGGL(kernel, g, dim3(threads), lds, st, a, slabC, noct, npl);
  };
  if (nvec <= 2) go(k_gn_onepass<2>, 0);
  else if (nvec <= 4) go(k_gn_onepass<4>, 1);
  else if (nvec <= 8) go(k_gn_onepass<8>, 2);
  else go(k_gn_onepass<12>, 3);
  return finish_launch("k_gn_onepass");
}

extern "C" int ctdd_unet_channel_stats(const float* x, int B, int HW, int C, double* stats, void* stream) {
  CTDD_REQUIRE(x && stats, CTDD_EINVAL, "null");
  hipLaunchKernelGGL(k_channel_stats, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, HW, C, stats);
  return finish_launch("k_channel_stats");
}

extern "C" int ctdd_unet_time(const void* args_, const float* proj_w, const float* proj_b, int Ntot, float* proj_out,
                              void* stream) {
  const TimeArgs& a = *(const TimeArgs*)args_;
  CTDD_REQUIRE(a.t && a.act && a.hid && a.tdim % 16 == 0 && a.ch % 16 == 0, CTDD_EINVAL, "bad time arguments");
  hipStream_t st = (hipStream_t)stream;
  const unsigned rb = (unsigned)((a.B + TRB - 1) / TRB);
  auto lds = [](int K) { return (size_t)TRB * (K > 256 ? K : 256) * sizeof(float); };     // inputs, or the 4 x 16 x 64 partial sums
  hipLaunchKernelGGL((k_rows_linear<true, true>), dim3((a.tdim + 63) / 64, rb), dim3(256), lds(a.ch), st,
                     a.t, a.w1, a.b1, a.B, a.ch, a.tdim, a.hid);
  if (int rc = finish_launch("k_rows_linear")) return rc;
  hipLaunchKernelGGL((k_rows_linear<false, true>), dim3((a.tdim + 63) / 64, rb), dim3(256), lds(a.tdim), st,
                     (const float*)a.hid, a.w2, a.b2, a.B, a.tdim, a.tdim, a.act);
  if (int rc = finish_launch("k_rows_linear")) return rc;
  hipLaunchKernelGGL((k_rows_linear<false, false>), dim3((Ntot + 63) / 64, rb), dim3(256), lds(a.tdim), st,
                     (const float*)a.act, proj_w, proj_b, a.B, a.tdim, Ntot, proj_out);
  return finish_launch("k_rows_linear");
}

extern "C" int ctdd_unet_time_uniform(const void* args_, const float* proj_w, const float* proj_b, int Ntot, float* proj_out,
                                      void* stream) {
  const TimeArgs& a = *(const TimeArgs*)args_;
  CTDD_REQUIRE(a.t && a.act && a.tdim % 64 == 0 && a.ch % 16 == 0 && Ntot > 0, CTDD_EINVAL, "bad time arguments (tdim %% 64, ch %% 16)");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_time_uniform_act, dim3(a.tdim / 64), dim3(256), (size_t)(a.ch + a.tdim + 256) * sizeof(float), st, a);
  if (int rc = finish_launch("k_time_uniform_act")) return rc;
  hipLaunchKernelGGL(k_time_uniform_proj, dim3((Ntot + 63) / 64), dim3(256), (size_t)(a.tdim + 256) * sizeof(float), st,
                     (const float*)a.act, a.tdim, proj_w, proj_b, Ntot, proj_out);
  return finish_launch("k_time_uniform_proj");
}

extern "C" int ctdd_unet_attention(const void* args_, void* stream) {
  const AttnArgs& a = *(const AttnArgs*)args_;
  const int ch = a.C / a.heads;
  {
    const int Tp = (a.T + 3) & ~3;
    const size_t lds4 = (size_t)(2 * ch * Tp + a.T * ch + a.T * Tp + a.T) * sizeof(float);
    static const bool old_attn = [] { const char* e = getenv("CTDD_ATTN_SMALL_OLD"); return e && e[0] == '1'; }();    // (A/B)
    if (!old_attn && ch % 4 == 0 && a.C % 4 == 0 && lds4 <= 160 * 1024) {
      static bool done[16] = {};
      ensure_lds_ceiling((const void*)k_attn_small_t4, done);
      hipLaunchKernelGGL(k_attn_small_t4, dim3(a.B * a.heads), dim3(256), lds4, (hipStream_t)stream, a);
      return finish_launch("k_attn_small_t4");
    }
  }
  const size_t lds = (size_t)(3 * a.T * ch + a.T * a.T + a.T) * sizeof(float);
  CTDD_REQUIRE(lds <= 160 * 1024, CTDD_ERANGE, "attention tile too large (T=%d)", a.T);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)k_attn_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_attn_small, dim3(a.B * a.heads), dim3(256), lds, (hipStream_t)stream, a);
  return finish_launch("k_attn_small");
}

extern "C" int ctdd_unet_logistic_head(const void* args_, void* stream) {
  const LogisticArgs& a = *(const LogisticArgs*)args_;
  const int64_t rows = (int64_t)a.B * a.C * a.HW;
  CTDD_REQUIRE(a.out || a.out_bf16, CTDD_EINVAL, "logistic head: no output");
  CTDD_REQUIRE(!a.out_bf16 || (a.fast && a.S % 4 == 0), CTDD_EINVAL, "logistic head: bf16 logits are the fast mode's, S %% 4 == 0");
  const int64_t wgs = (rows + 3) / 4;
  hipLaunchKernelGGL(k_logistic_head, dim3((unsigned)(wgs < 4096 ? wgs : 4096)), dim3(256), 0, (hipStream_t)stream, a);
  return finish_launch("k_logistic_head");
}
