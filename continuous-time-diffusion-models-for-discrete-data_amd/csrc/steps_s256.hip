// steps_s256.hip -- S = 256 (MNIST / CIFAR-10) fused tau-leaping step on the matrix cores.
//
// Per (n,d) row the reference computes (lib/sampling/sampling.py:32-59, 119-160)
//     ratio[s] = sum_s0 softmax(logits)[s0] / (qt0[s0][x] + eps) * qt0[s0][s]       (2*S^2 FLOP)
//     R^[s]    = beta * R[s][x] * ratio[s],  own state masked,  jumps ~ Poisson(R^ h)
// i.e. a (rows x 256) x (256 x 256) contraction with a row-dependent left operand: 131 kFLOP per
// 1 KiB row, so at S = 256 the step is MFMA-bound before it is HBM-bound.
//
// Mapping (one 256-thread workgroup = 4 waves = 128 rows, each wave owns 32 rows end to end):
//   phase 1  wave loads its rows coalesced (one 1-KiB row per dwordx4 wave-instruction), row max
//            and sum by wave reduction, w = exp(l-max) * invq[x][s0]  (invq = 1/(qt0[s0][x]+eps),
//            a per-step table), splits w into bf16 hi+lo and writes both planes to LDS in the MFMA
//            B-operand image (XOR-swizzled 16-B chunks: conflict-free ds_read_b128);
//   phase 2  out^T = qt0^T (A, 256 x 256) . w^T (B, 256 x 32 per wave) on v_mfma_f32_32x32x16_bf16
//            with the split-precision products hi*hi + hi*lo + lo*hi (fp32 accumulate; the
//            dropped lo*lo term is 2^-16 relative; all terms are >= 0 so there is no
//            cancellation: measured max rel. error vs fp64 < 2e-5, tests/test_gpu_s256.py).
//            The wave first pulls its w image back into registers as MFMA B fragments; the whole
//            160 KiB of LDS then serves as a 10-deep ring of 16-KiB K-chunks of A, filled by
//            global_load_lds from a per-step image pre-arranged by k_step_tables (linear copy =
//            fragment order) and retired with counted s_waitcnt vmcnt(N) + raw s_barrier;
//            the transposed orientation leaves each row's 256 outputs in ONE lane pair, so
//   phase 3  the epilogue is in-register: acc *= RT0[x][s] (forward rate, own state pre-zeroed,
//            read in accumulator order from the L2-resident view), T = sum, Lambda = beta*h*T/Z,
//            then the jump draw of csrc/draw.hpp over the lane pair.
// LDS: 160 KiB (w images 128 KiB + two early A slots, then ten A slots) -> one workgroup per CU.
#include <type_traits>

#include "draw.hpp"
#include "steps_s256.hpp"

namespace ctdd {

// mirror of the struct in steps_generic.hip (kept in sync by hand; both files are small)
enum Mode { MODE_RATES = 0, MODE_LOGPROB = 1, MODE_TAULEAP = 2, MODE_LBJF = 3, MODE_MIDPOINT = 4,
            MODE_DRAW_ONLY = 5, MODE_EXACT = 6 };

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int TILE_ROWS = 128;
constexpr int CHUNK_BYTES = S256_CHUNK_BYTES;          // one K-step (16 s0) of A: [plane 2][g 2][s 256][8 bf16]
constexpr size_t STEP_TABLE_BYTES = S256_STEP_TABLE_BYTES;

// ------------------------------------------------------------------ per-step derived tables
// invq[x][s0] = 1 / (qt0[s0][x] + eps)                          (fp32, IEEE division)
// A_img[kk][plane][g][s][j] = bf16 hi / lo of qt0[s0 = 16kk + 8g + j][s]
__device__ inline unsigned short bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);           // inputs are finite and >= 0 here
  return (unsigned short)(u >> 16);
}
__device__ inline float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

__global__ __launch_bounds__(256) void k_step_tables(const float* __restrict__ qt0, float eps, int nT, int crm,
                                                     unsigned char* __restrict__ out) {
  const int t = blockIdx.y;
  const float* q = qt0 + (size_t)t * S256 * S256;
  unsigned char* base = out + (size_t)t * STEP_TABLE_BYTES;
  float* invq = (float*)base;
  unsigned short* img = (unsigned short*)(base + (size_t)S256 * S256 * 4);
  unsigned short* invq16 = (unsigned short*)(base + S256_INVQ16_OFFSET);      // the same table in bf16 (single-product kernel)
  // each workgroup handles 16 s0 rows (one K-chunk): threads = s
  const int kk = blockIdx.x, s = threadIdx.x;
  for (int r = 0; r < 16; ++r) {
    const int s0 = 16 * kk + r;
    const float v = q[(size_t)s0 * S256 + s];
    const float iq = crm ? 1.0f : 1.0f / (v + eps);                 // CRM branch: plain p0t @ qt0
    invq[(size_t)s * S256 + s0] = iq;                               // transposed write (x = s here)
    invq16[(size_t)s * S256 + s0] = bf16_rne(iq);
    const unsigned short hi = bf16_rne(v);
    const unsigned short lo = bf16_rne(v - bf16_to_f32(hi));
    const int g = r >> 3, j = r & 7;
    const size_t e = (size_t)kk * (CHUNK_BYTES / 2) + ((size_t)(0 * 2 + g) * S256 + s) * 8 + j;
    img[e] = hi;
    img[e + (size_t)2 * S256 * 8] = lo;                        // plane 1
  }
}

// RT0[x][s] = R[s][x] (s != x) else 0 ;  R0[x][s] = R[x][s] (s != x) else 0     (per model)
__global__ void k_rate_views(const float* __restrict__ R, int S, float* __restrict__ RT0, float* __restrict__ R0) {
  const int x = blockIdx.x;
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    RT0[(size_t)x * S + s] = (s == x) ? 0.0f : R[(size_t)s * S + x];
    R0[(size_t)x * S + s] = (s == x) ? 0.0f : R[(size_t)x * S + s];
  }
}


// ---- wave-wide reductions of FOUR independent values at once, entirely on DPP: the four chains
// are interleaved so every DPP read sits >= 3 instructions behind the write it depends on (the
// VALU-write -> DPP-read hazard needs 2 wait states; nothing is padded inside inline asm).
// After row_bcast:15 / row_bcast:31 lane 63 holds the wave-wide result.
#define DPP4(op, ctrl)                                         \
  op " %0, %0, %0 " ctrl "\n\t" op " %1, %1, %1 " ctrl "\n\t" \
  op " %2, %2, %2 " ctrl "\n\t" op " %3, %3, %3 " ctrl "\n\t"
#define REDUCE4(op)                                                                      \
  asm("s_nop 1\n\t" DPP4(op, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")          \
      DPP4(op, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")                         \
      DPP4(op, "row_ror:4 row_mask:0xf bank_mask:0xf")                                   \
      DPP4(op, "row_ror:8 row_mask:0xf bank_mask:0xf")                                   \
      DPP4(op, "row_bcast:15 row_mask:0xa bank_mask:0xf")                                \
      DPP4(op, "row_bcast:31 row_mask:0xc bank_mask:0xf")                                \
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3))
__device__ inline float rl63(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ inline void wave_max4(float& v0, float& v1, float& v2, float& v3) {
  REDUCE4("v_max_f32_dpp");
  v0 = rl63(v0); v1 = rl63(v1); v2 = rl63(v2); v3 = rl63(v3);
}
__device__ inline void wave_sum4(float& v0, float& v1, float& v2, float& v3) {
  REDUCE4("v_add_f32_dpp");
  v0 = rl63(v0); v1 = rl63(v1); v2 = rl63(v2); v3 = rl63(v3);
}

using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ inline unsigned pack_bf16(float a, float b) {       // -> v_cvt_pk_bf16_f32 (RNE)
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

#ifdef CTDD_S256_STAMPS      // diagnostic build only: per-wave phase time stamps go to a.out_changed
#define STAMP(i) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps[i] = t_; }
#else
#define STAMP(i)
#endif

constexpr int NSLOT = 10;      // 160 KiB of LDS = ten 16-KiB K-chunks of the A image
// LDS-DMA bookkeeping (4 ops per chunk per wave).  Ten chunks are issued up front; chunk c+10 is
// issued in the middle of step c+1 (c = 0..5), right after the barrier that retires chunk c's slot.
// mid_wait(kk) = ops that may stay in flight when the middle of step kk waits for chunk kk+1:
// the younger chunks plus the epilogue's RT0 prefetches (16 loads in step 8, 16 in step 12).
__host__ __device__ constexpr int mid_wait(int kk) {
  const int refills = kk - 1 < 0 ? 0 : (kk - 1 > 6 ? 6 : kk - 1);
  return 4 * (8 + refills - kk) + (kk >= 13 ? 32 : kk >= 9 ? 16 : 0);
}
template <int N> __device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(256, 1) void k_tauleap_s256(const S256Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef CTDD_S256_STAMPS
  unsigned long long stamps[6];
#endif
  STAMP(0)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t wrow0 = (int64_t)blockIdx.x * TILE_ROWS + wave * 32;
  const float* invq = (const float*)a.tables;
  const unsigned char* aimg = a.tables + (size_t)S256 * S256 * 4;
  const int j = lane & 31, g = lane >> 5;

  // LDS = ring of NSLOT 16-KiB slots.  Slots 8,9 are free from the start; slots 2w,2w+1 first
  // hold wave w's w image (bf16 hi | lo planes, 32 rows x 512 B each) until it sits in registers.
  auto stage_chunk = [&](int kk, int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave * 4 + i;
      const unsigned char* src = aimg + (size_t)kk * CHUNK_BYTES + piece * 1024 + lane * 16;
      unsigned char* dst = smem + slot * CHUNK_BYTES + piece * 1024;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                       (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
    }
  };
  stage_chunk(0, 8);
  stage_chunk(1, 9);

  // ---- phase 1: softmax pieces of the wave's 32 rows -> w = exp(l - max) * invq[x] as bf16 hi/lo
  unsigned char* wl_hi = smem + wave * 32768;
  unsigned char* wl_lo = wl_hi + 16384;
  float zv = 1.0f;                     // lane l keeps Z of row (l & 31)
  int xj, xcur;                        // rate-state / current state of row (l & 31)
  {
    const int64_t rj = wrow0 + j;
    const int64_t rjc = rj < a.R ? rj : a.R - 1;
    xcur = min(max(a.x[rjc], 0), S256 - 1);
    xj = a.x_base ? min(max(a.x_base[rjc], 0), S256 - 1) : xcur;
  }
  constexpr float LOG2E = 1.4426950408889634f;
  float4 l[2][8], iq[2][8];                 // two batches of 8 rows in flight (prefetch one ahead)
  auto load_batch = [&](int b, float4 (&L)[8], float4 (&Q)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int64_t row = wrow0 + b * 8 + r;
      const int64_t rowc = row < a.R ? row : a.R - 1;
      const int xr = __builtin_amdgcn_readlane(xj, b * 8 + r);
      L[r] = *(const float4*)(a.logits + (size_t)rowc * S256 + lane * 4);
      Q[r] = *(const float4*)(invq + (size_t)xr * S256 + lane * 4);
    }
  };
  auto do_batch = [&](int b, const float4 (&L)[8], const float4 (&Q)[8]) {
#pragma unroll
    for (int r4 = 0; r4 < 8; r4 += 4) {
      float m0 = fmaxf(fmaxf(L[r4].x, L[r4].y), fmaxf(L[r4].z, L[r4].w));
      float m1 = fmaxf(fmaxf(L[r4 + 1].x, L[r4 + 1].y), fmaxf(L[r4 + 1].z, L[r4 + 1].w));
      float m2 = fmaxf(fmaxf(L[r4 + 2].x, L[r4 + 2].y), fmaxf(L[r4 + 2].z, L[r4 + 2].w));
      float m3 = fmaxf(fmaxf(L[r4 + 3].x, L[r4 + 3].y), fmaxf(L[r4 + 3].z, L[r4 + 3].w));
      wave_max4(m0, m1, m2, m3);
      const float ms[4] = {m0 * LOG2E, m1 * LOG2E, m2 * LOG2E, m3 * LOG2E};
      float e[4][4], z[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = L[r4 + i];
        e[i][0] = __builtin_amdgcn_exp2f(fmaf(v.x, LOG2E, -ms[i]));
        e[i][1] = __builtin_amdgcn_exp2f(fmaf(v.y, LOG2E, -ms[i]));
        e[i][2] = __builtin_amdgcn_exp2f(fmaf(v.z, LOG2E, -ms[i]));
        e[i][3] = __builtin_amdgcn_exp2f(fmaf(v.w, LOG2E, -ms[i]));
        z[i] = (e[i][0] + e[i][1]) + (e[i][2] + e[i][3]);
      }
      wave_sum4(z[0], z[1], z[2], z[3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr_ = b * 8 + r4 + i;
        zv = (j == rr_) ? z[i] : zv;
        const float4 q = Q[r4 + i];
        const float w0 = e[i][0] * q.x, w1 = e[i][1] * q.y, w2 = e[i][2] * q.z, w3 = e[i][3] * q.w;
        const unsigned h01 = pack_bf16(w0, w1), h23 = pack_bf16(w2, w3);
        const unsigned o01 = pack_bf16(w0 - __uint_as_float(h01 << 16), w1 - __uint_as_float(h01 & 0xFFFF0000u));
        const unsigned o23 = pack_bf16(w2 - __uint_as_float(h23 << 16), w3 - __uint_as_float(h23 & 0xFFFF0000u));
        // element s0 = 4*lane.. lives in 16-B chunk c = lane/2 (8 bf16), half lane&1; swizzle c ^= (row & 15)
        const int off = rr_ * 512 + (((lane >> 1) ^ (rr_ & 15)) << 4) + ((lane & 1) << 3);
        *(uint2*)(wl_hi + off) = make_uint2(h01, h23);
        *(uint2*)(wl_lo + off) = make_uint2(o01, o23);
      }
    }
  };
  load_batch(0, l[0], iq[0]);
  load_batch(1, l[1], iq[1]);
  do_batch(0, l[0], iq[0]);
  load_batch(2, l[0], iq[0]);
  do_batch(1, l[1], iq[1]);
  load_batch(3, l[1], iq[1]);
  do_batch(2, l[0], iq[0]);
  do_batch(3, l[1], iq[1]);
  // B operand of the wave's 32 rows into registers: lane (j,g) holds w[row j][16kk + 8g .. +7]
  bf16x8 bh[16], bl[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int boff = j * 512 + (((2 * kk + g) ^ (j & 15)) << 4);
    bh[kk] = *(const bf16x8*)(wl_hi + boff);
    bl[kk] = *(const bf16x8*)(wl_lo + boff);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();          // every wave's w image is in registers: slots 0..7 are free
#pragma unroll
  for (int c = 2; c < 10; ++c) stage_chunk(c, c - 2);
  STAMP(1)

  // ---- phase 2: acc[m] (32 s x 32 rows) += A_chunk(kk) . B_chunk(kk), 16 K-steps, A ring 10 deep
  f32x16 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.0f;
  // forward-rate pieces RT0[x_j][s] for the epilogue, prefetched in accumulator order during the
  // last K-steps (B registers of finished steps are free by then): m-tiles 0-3 in step 8, 4-7 in 12
  const float* frow = a.RT0 + (size_t)xj * S256 + 4 * g;
  float4 fpre[32];
  // Software pipeline over quarter steps (groups of 2 s-tiles = 6 MFMAs): the A fragments of group
  // G+2 are requested from LDS before the MFMAs of group G issue, so two groups (384 cycles) cover
  // the LDS latency and at most 12 ds_reads are ever in flight (lgkmcnt is a 4-bit counter).
  bf16x8 fh[3][2], fl[3][2];
  auto read_group = [&](auto GG) {
    constexpr int G = decltype(GG)::value, kk = G / 4, i = G % 4, bsel = G % 3;
    const unsigned char* ab = smem + ((8 + kk) % NSLOT) * CHUNK_BYTES + (g * S256 + j) * 16 + i * 1024;
    fh[bsel][0] = *(const bf16x8*)(ab);
    fh[bsel][1] = *(const bf16x8*)(ab + 512);
    fl[bsel][0] = *(const bf16x8*)(ab + 8192);
    fl[bsel][1] = *(const bf16x8*)(ab + 8192 + 512);
  };
  wait_vmcnt<36>();                              // chunk 0 (nine younger chunks may stay in flight)
  __builtin_amdgcn_s_barrier();
  read_group(std::integral_constant<int, 0>{});
  read_group(std::integral_constant<int, 1>{});
  auto kgroup = [&](auto GG) {
    constexpr int G = decltype(GG)::value, kk = G / 4, i = G % 4, bsel = G % 3;
    if constexpr (i == 2 && kk + 1 < 16) {
      wait_vmcnt<mid_wait(kk)>();                 // my four pieces of chunk kk+1 have landed
      __builtin_amdgcn_s_barrier();               // ... and everyone's; chunk kk-1 is out of use
      if constexpr (kk >= 1 && kk - 1 + 10 < 16) stage_chunk(kk - 1 + 10, (8 + kk - 1) % NSLOT);
      if constexpr (kk == 8 || kk == 12) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          constexpr int base = kk == 8 ? 0 : 16;
          fpre[base + q] = *(const float4*)(frow + 32 * ((base + q) >> 2) + 8 * ((base + q) & 3));
        }
      }
    }
    if constexpr (G + 2 < 64) read_group(std::integral_constant<int, G + 2>{});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
      const int m = 2 * i + mm;
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[bsel][mm], bh[kk], acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[bsel][mm], bl[kk], acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[bsel][mm], bh[kk], acc[m], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto kstep = [&](auto KK) {
    constexpr int kk = decltype(KK)::value;
    kgroup(std::integral_constant<int, 4 * kk + 0>{});
    kgroup(std::integral_constant<int, 4 * kk + 1>{});
    kgroup(std::integral_constant<int, 4 * kk + 2>{});
    kgroup(std::integral_constant<int, 4 * kk + 3>{});
  };
#define KS(n) kstep(std::integral_constant<int, n>{});
  KS(0) KS(1) KS(2) KS(3) KS(4) KS(5) KS(6) KS(7) KS(8) KS(9) KS(10) KS(11) KS(12) KS(13) KS(14) KS(15)
#undef KS
  STAMP(2)

  // ---- phase 3: epilogue.  lane (j,g) holds out[s][row j] for s = 32m + 8q + 4g + p, reg = 4q + p.
  // acc *= RT0[x_j][s] (forward rate R[s][x], own state pre-zeroed), read in accumulator order
  // straight from the L2-resident 256-KiB view (two adjacent 16-B pieces per row per instruction).
  const bool corrector = a.flags & CTDD_STEP_CORRECTOR;
  // CRM branch with logit_type reverse_prob (sampling.py:61-73, model_utils.py:30-60): the step tables carry invq = 1, so
  // acc[s] = Z (p0t @ qt0)[s]; ratio[s] = exp(ll_all[s] - ll_xt) = (acc[s] + 1e-35 Z) / (acc[x] + 1e-35 Z) and the rate
  // view is R[x][s] (the launcher passes R0 as the view): same epilogue with acc[x] + 1e-35 Z in the place of Z.
  const bool crm = a.flags & CTDD_STEP_CRM;
  float norm = zv, addn = 0.0f;
  if (crm) {
    float ox = 0.0f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) ox = (32 * m + 8 * (r >> 2) + 4 * g + (r & 3) == xj) ? acc[m][r] : ox;
    ox += __shfl_xor(ox, 32, WAVE);                            // (the pair's other lane holds it, or 0)
    addn = 1e-35f * zv;
    norm = ox + addn;
  }
  const float invz = 1.0f / norm;
  const float scale = a.beta * invz;                            // true rate = scale * r
  const int64_t myrow = wrow0 + j;
  const bool live = myrow < a.R;
  const float* crow = a.R0 + (size_t)xj * S256 + 4 * g;
  // the masked rates r (units of beta/Z) go to the wave's own LDS region as [block 32][lane 64]
  // float4 (block = 8 consecutive destinations shared by the lane pair), so that the draw code
  // below can be rolled loops instead of 128 unrolled copies
  __builtin_amdgcn_s_barrier();                                 // every wave has left the A ring
  float4* rl4 = (float4*)(smem + wave * 32768) + lane;
  float T = 0.0f;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 f = fpre[4 * m + q];
      float v0 = (acc[m][4 * q + 0] + addn) * f.x, v1 = (acc[m][4 * q + 1] + addn) * f.y, v2 = (acc[m][4 * q + 2] + addn) * f.z,
            v3 = (acc[m][4 * q + 3] + addn) * f.w;
      if (corrector) {                                          // r_s += Z * R[x][s]  (Z: the normaliser in use)
        const float4 c = *(const float4*)(crow + 32 * m + 8 * q);
        v0 = fmaf(norm, c.x, v0); v1 = fmaf(norm, c.y, v1); v2 = fmaf(norm, c.z, v2); v3 = fmaf(norm, c.w, v3);
      }
      T += (v0 + v1) + (v2 + v3);
      rl4[(4 * m + q) * 64] = make_float4(v0, v1, v2, v3);
      if (a.out_rates && live)
        *(float4*)(a.out_rates + (size_t)myrow * S256 + 32 * m + 8 * q + 4 * g) =
            make_float4(scale * v0, scale * v1, scale * v2, scale * v3);
    }
    __builtin_amdgcn_sched_barrier(0);                          // keep the loads' live ranges per m-tile
  }
  T += __shfl_xor(T, 32, WAVE);
  if (a.out_rates && !a.out_x) return;
  STAMP(3)

  const float Lam = scale * T * a.h;
  const bool ordinal = a.flags & CTDD_STEP_ORDINAL;
  const uint64_t rngrow = (uint64_t)(live ? myrow : a.R - 1);
  int jump = 0, njumps = 0;               // njumps: jump events drawn for this dimension (sum_s k_s)
  if (Lam > 0.0f && Lam <= SUPERPOSE_MAX_LAMBDA) {
    PhiloxStream rng(a.seed, a.offset, rngrow, 0u);             // both lanes of the pair: same stream
    const int K = poisson_row(Lam, rng);
    njumps = K;
    if (K > 0 && (ordinal || K == 1)) {
      // cumulative rates in destination order (block of 8 = g0's four then g1's four), written
      // back over the lane's own LDS column; the ends of the lane's 32 blocks stay in registers
      float cend[32];
      {
        float mine[32], other[32];
#pragma unroll
        for (int blk = 0; blk < 32; ++blk) {
          const float4 v = rl4[blk * 64];
          mine[blk] = (v.x + v.y) + (v.z + v.w);
        }
#pragma unroll
        for (int blk = 0; blk < 32; ++blk) other[blk] = __shfl_xor(mine[blk], 32, WAVE);
        float run = 0.0f;
#pragma unroll
        for (int blk = 0; blk < 32; ++blk) {
          const float4 v = rl4[blk * 64];
          float c = run + (g == 0 ? 0.0f : other[blk]);
          float4 cs;
          c += v.x; cs.x = c; c += v.y; cs.y = c; c += v.z; cs.z = c; c += v.w; cs.w = c;
          rl4[blk * 64] = cs;
          cend[blk] = c;
          run += (g == 0 ? mine[blk] + other[blk] : other[blk] + mine[blk]);
        }
      }
      for (int d = 0; d < K; ++d) {
        const float target = rng.next() * T;
        // destinations with cumulative <= target precede the pick: count mine, add the partner's
        int nb = 0;
#pragma unroll
        for (int blk = 0; blk < 32; ++blk) nb += (cend[blk] <= target) ? 1 : 0;
        int cnt = 4 * nb;
        if (nb < 32) {
          const float4 cs = rl4[nb * 64];
          cnt += (cs.x <= target) + (cs.y <= target) + (cs.z <= target);
        }
        cnt += __shfl_xor(cnt, 32, WAVE);
        jump += min(cnt, S256 - 1) - xj;
      }
    }
  } else if (Lam > SUPERPOSE_MAX_LAMBDA) {
    // dense regime: per sub-block of 4 consecutive destinations (draw.hpp: subblock_draw); the lane
    // owns sub-blocks b = 8m + 2q + g.  Each lane builds Philox block 2m+g and trades it with its
    // partner, so both see blocks 2m and 2m+1 (uniform (b & 3) of block b >> 2).
    const float sh = scale * a.h;
    int cnt = 0;
    long long jl = 0;
    for (int m = 0; m < 8; ++m) {
      const u4 mine = philox_row(a.seed, a.offset, rngrow, DENSE_DRAW0 + (uint32_t)(2 * m + g));
      u4 oth;
      oth.x = __shfl_xor(mine.x, 32, WAVE); oth.y = __shfl_xor(mine.y, 32, WAVE);
      oth.z = __shfl_xor(mine.z, 32, WAVE); oth.w = __shfl_xor(mine.w, 32, WAVE);
      const u4 lo = g == 0 ? mine : oth, hi = g == 0 ? oth : mine;      // blocks 2m, 2m+1
      // q = 0,1 -> block 2m components g, 2+g ; q = 2,3 -> block 2m+1 components g, 2+g
      const uint32_t w0 = g == 0 ? lo.x : lo.y, w1 = g == 0 ? lo.z : lo.w;
      const uint32_t w2 = g == 0 ? hi.x : hi.y, w3 = g == 0 ? hi.z : hi.w;
      const float4 v0 = rl4[(4 * m + 0) * 64], v1 = rl4[(4 * m + 1) * 64], v2 = rl4[(4 * m + 2) * 64],
                   v3 = rl4[(4 * m + 3) * 64];
      const int b0 = 8 * m + g;
      cnt += min(subblock_draw(v0.x, v0.y, v0.z, v0.w, sh, u01(w0), a.seed, a.offset, rngrow, b0, xj, 4, &jl), 1 << 20);
      cnt += min(subblock_draw(v1.x, v1.y, v1.z, v1.w, sh, u01(w1), a.seed, a.offset, rngrow, b0 + 2, xj, 4, &jl), 1 << 20);
      cnt += min(subblock_draw(v2.x, v2.y, v2.z, v2.w, sh, u01(w2), a.seed, a.offset, rngrow, b0 + 4, xj, 4, &jl), 1 << 20);
      cnt += min(subblock_draw(v3.x, v3.y, v3.z, v3.w, sh, u01(w3), a.seed, a.offset, rngrow, b0 + 6, xj, 4, &jl), 1 << 20);
    }
    cnt += __shfl_xor(cnt, 32, WAVE);
    jl += __shfl_xor(jl, 32, WAVE);
    jl = jl > S256 ? S256 : (jl < -S256 ? -S256 : jl);        // |jump| >= S - 1 saturates the state clamp either way
    jump = (ordinal || cnt <= 1) ? (int)jl : 0;
    njumps = cnt;
  }
  STAMP(4)
#ifdef CTDD_S256_STAMPS
  if (lane == 0) {
    unsigned long long* o = (unsigned long long*)a.out_changed + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int i = 0; i < 5; ++i) o[i] = stamps[i];
    o[5] = __builtin_amdgcn_s_memrealtime();
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    o[6] = ((unsigned long long)xcc << 32) | hwid;
  }
  return;
#endif
  bool moved = false;
  if (live && g == 0) {
    const int xn = min(max(xcur + jump, 0), S256 - 1);
    a.out_x[myrow] = xn;
    moved = (a.flags & CTDD_STEP_COUNT_RAW) ? (jump != 0) : (xn != xcur);
  }
  if (a.out_changed) {                       // one atomic per wave instead of one per row on a single address
    const int nmoved = __builtin_popcountll(__ballot(moved));
    if (lane == 0 && nmoved) atomicAdd(a.out_changed, nmoved);
    if (a.flags & CTDD_STEP_COUNT_JUMPS) {   // sampling.py:489-495: dimensions with >= 1 and with > 1 jump events
      const int n1 = __builtin_popcountll(__ballot(live && g == 0 && njumps > 0));
      const int n2 = __builtin_popcountll(__ballot(live && g == 0 && njumps > 1));
      if (lane == 0 && n1) atomicAdd(a.out_changed + 1, n1);
      if (lane == 0 && n2) atomicAdd(a.out_changed + 2, n2);
    }
  }
}

}  // namespace ctdd
using namespace ctdd;

extern "C" int64_t ctdd_s256_step_table_bytes(void) { return (int64_t)STEP_TABLE_BYTES; }

static int s256_prepare(const float* qt0, const float* base_rate, float eps, int nT, int crm, void* out_step_tables, float* out_RT0,
                        float* out_R0, void* stream);
extern "C" int ctdd_s256_prepare(const float* qt0, const float* base_rate, float eps, int nT,
                                 void* out_step_tables, float* out_RT0, float* out_R0, void* stream) {
  return s256_prepare(qt0, base_rate, eps, nT, 0, out_step_tables, out_RT0, out_R0, stream);
}
extern "C" int ctdd_s256_prepare_crm(const float* qt0, const float* base_rate, int nT, void* out_step_tables, float* out_RT0,
                                     float* out_R0, void* stream) {
  return s256_prepare(qt0, base_rate, 0.0f, nT, 1, out_step_tables, out_RT0, out_R0, stream);
}
static int s256_prepare(const float* qt0, const float* base_rate, float eps, int nT, int crm, void* out_step_tables, float* out_RT0,
                        float* out_R0, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (out_step_tables) {
    CTDD_REQUIRE(qt0 && nT > 0 && nT <= 65535, CTDD_EINVAL, "bad qt0 / nT=%d", nT);
    hipLaunchKernelGGL(k_step_tables, dim3(16, nT), dim3(256), 0, st, qt0, eps, nT, crm, (unsigned char*)out_step_tables);
    if (int rc = finish_launch("k_step_tables")) return rc;
  }
  if (out_RT0 || out_R0) {
    CTDD_REQUIRE(base_rate && out_RT0 && out_R0, CTDD_EINVAL, "base_rate, RT0 and R0 are needed together");
    hipLaunchKernelGGL(k_rate_views, dim3(S256), dim3(256), 0, st, base_rate, S256, out_RT0, out_R0);
    if (int rc = finish_launch("k_rate_views")) return rc;
  }
  return CTDD_OK;
}

extern "C" int ctdd_tauleap_step_s256(const void* logits, const int32_t* x, const int32_t* x_base,
                                      const void* step_tables, const float* RT0, const float* R0, float beta,
                                      float h, uint32_t flags, uint64_t seed, uint64_t offset, int N, int D,
                                      float* out_rates, int32_t* out_x, int32_t* out_changed, void* stream) {
  CTDD_REQUIRE(logits && x && step_tables && RT0 && R0, CTDD_EINVAL, "null input");
  CTDD_REQUIRE(out_x || out_rates, CTDD_EINVAL, "no output requested");
  CTDD_REQUIRE(N > 0 && D > 0, CTDD_EINVAL, "N=%d D=%d must be positive", N, D);
  CTDD_REQUIRE(!(flags & CTDD_STEP_LOGITS_BF16) || (flags & CTDD_STEP_BF16), CTDD_EINVAL, "bf16 logits need CTDD_STEP_BF16");
  S256Args a{};
  a.logits = (const float*)logits; a.x = x; a.x_base = x_base; a.tables = (const unsigned char*)step_tables;
  a.RT0 = (flags & CTDD_STEP_CRM) ? R0 : RT0;      // CRM branch: forward rate out of x, R[x][s]
  a.R0 = R0; a.beta = beta; a.h = h; a.flags = flags; a.seed = seed; a.offset = offset;
  a.R = (int64_t)N * D; a.out_rates = out_rates; a.out_x = out_x; a.out_changed = out_changed;
  if (flags & CTDD_STEP_BF16) return launch_tauleap_s256_b16(a, (hipStream_t)stream);
  const int64_t grid = (a.R + TILE_ROWS - 1) / TILE_ROWS;
  CTDD_REQUIRE(grid < (1ll << 31), CTDD_ERANGE, "too many rows");
  static bool attr_done[16] = {};
  ensure_lds_ceiling((const void*)k_tauleap_s256, attr_done);
  hipLaunchKernelGGL(k_tauleap_s256, dim3((unsigned)grid), dim3(256), 163840, (hipStream_t)stream, a);
  return finish_launch("k_tauleap_s256");
}

namespace ctdd {
struct StepArgs;
int try_s256(const StepArgs&, void*, int*) { return 0; }   // plain entry points keep the generic kernel
}  // namespace ctdd
