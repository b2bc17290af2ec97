// steps_s256_b16.hip -- S = 256 fused tau-leaping step, single-product bf16 variant (CTDD_STEP_BF16).
//
// Same mathematics, tables, flags and draw rule as steps_s256.hip (lib/sampling/sampling.py:32-59, 119-160):
//     ratio[s] = sum_s0 softmax(logits)[s0] / (qt0[s0][x] + eps) * qt0[s0][s],  R^[s] = beta R[s][x] ratio[s],
//     own state masked, jumps ~ Poisson(R^ h)  (superposition draw of draw.hpp)
// but the contraction is ONE v_mfma_f32_32x32x16_bf16 product per tile (w and qt0 rounded to bf16 once; every term is
// >= 0, so the relative error of a rate is <= 3 * 2^-8 = 1.2e-2: w, qt0 and the bf16 copy of 1/(qt0+eps) each round once) -- the mode the bf16 score network runs with.  A third of the
// matrix work, half the B-fragment registers and half the LDS of the parity kernel, which is what lets TWO workgroups
// share a CU (<= 256 registers per lane, 80 KiB of LDS each): one workgroup's softmax and draw phases (vector ALU,
// memory latency) run under the other's matrix phase.
//
// Workgroup = 256 threads = 4 waves = 128 rows; a wave owns 32 rows end to end.
//   phase 1  the wave's rows come in by LDS-DMA, 16 rows (16 KiB) at a time, into a wave-private staging buffer (16-byte
//            pieces XOR-swizzled on the SOURCE address so the reads below are conflict-free).  Four lanes share a row: each
//            reads a quarter row in the order of the MFMA B fragments, so max / sum are in-lane reductions plus two lane
//            exchanges (no 64-lane reductions), w = e^{l-max} / (qt0[s0][x] + eps) is formed in fragment order from a
//            gathered invq row, and packed to bf16.  Two v_permlane16_swap per fragment register pair move the quarter
//            rows to the lane pair (row j, half g) the matrix instruction wants: no LDS round trip for w.
//   phase 2  out^T = qt0^T . w^T: the hi plane of the per-step A image streamed by LDS-DMA through a 5-slot ring of
//            16-KiB chunks (two K-steps each) with counted vmcnt waits and one raw barrier per chunk; A fragments are
//            read three MFMA pairs ahead.
//   phase 3  rates = acc * RT0[x] in registers; rows in the superposition regime prefix-sum them IN PLACE, keep every
//            8th cumulative value in registers (16 per lane) and park the rest in LDS as bf16 offsets from their group
//            start (20 KiB per wave); a pick is 16 register compares + one 16-byte LDS read.  Dense rows
//            (Lambda > 64) run the sub-block draw of draw.hpp on the raw rates first.
// Destination order, Philox counters and every decision rule are those of steps_s256.hip / oracle/philox.py.
#include <type_traits>

#include "draw.hpp"
#include "steps_s256.hpp"

namespace ctdd {

namespace b16 {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x4 = __attribute__((ext_vector_type(4))) float;    // (staging arrays of the struct type float4 went to scratch memory)

constexpr int TILE_ROWS = 128;
constexpr int LDS_BYTES = 81920;            // two workgroups per CU
constexpr int SLOT_BYTES = 16384;           // one ring slot = two K-steps of the hi plane (2 x 8 KiB)
constexpr int NSLOT = 5;
constexpr int NCHUNK = 8;

__device__ inline unsigned pack_bf16(float a, float b) {       // -> v_cvt_pk_bf16_f32 (RNE)
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ inline float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ inline float bf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

// The two lanes of a pair (l, l ^ 32) both get (value of the lower lane, value of the upper lane): one v_permlane32_swap on the
// vector ALU instead of a ds_bpermute round trip through the LDS queue.
__device__ inline void pair_values(unsigned mine, unsigned& lower, unsigned& upper) {
  const auto r = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
  lower = r[0]; upper = r[1];
}
__device__ inline float pair_sum(float v) {                     // lower + upper, bitwise the same in both lanes
  unsigned lo, hi;
  pair_values(__float_as_uint(v), lo, hi);
  return __uint_as_float(lo) + __uint_as_float(hi);
}
__device__ inline int pair_sum(int v) {
  unsigned lo, hi;
  pair_values((unsigned)v, lo, hi);
  return (int)(lo + hi);
}
__device__ inline float quad_max(float v) {                     // over the four lanes l ^ {0, 16, 32, 48}
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  unsigned lo, hi;
  pair_values(__float_as_uint(v), lo, hi);
  return fmaxf(__uint_as_float(lo), __uint_as_float(hi));
}
__device__ inline float quad_sum(float v) {                     // (v[l] + v[l ^ 16]) + (the same of the other half), as the shuffles summed
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  return pair_sum(v);
}

template <int N> __device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__host__ __device__ constexpr int slot_of(int c) { return (c + 4) % NSLOT; }
// vector-memory operations that may stay in flight when chunk c's group 5 waits for chunk c+1: the DMA pieces of the younger
// chunks (4 per chunk per wave) and, in chunk 6, the twelve forward-rate loads of the epilogue's first group, requested at
// group 48 (they are younger than chunk 7's pieces; waiting for them there stalled the last two chunks on an L2 round trip)
__host__ __device__ constexpr int mid_wait(int c) { return c == 0 ? 12 : c <= 4 ? 8 : c == 5 ? 4 : 12; }

#ifdef CTDD_S256_STAMPS      // diagnostic build only: per-wave phase time stamps go to a.out_changed
#define B16_STAMP(i) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps[i] = t_; }
#else
#define B16_STAMP(i)
#endif

// GENERAL = false: the sampler's plain step (no corrector term, no CRM normaliser, no rate output, no x_base): the same
// code with those uniform branches compiled out, so the epilogue is straight-line.
// L16 = true: the logits are bf16 (the bf16 network's output convolution writes them that way: half the bytes of the
// step's one large read); both passes then fit the staging buffer and come in by LDS-DMA.
template <bool GENERAL, bool L16>
__global__ __launch_bounds__(256, 2) void k_tauleap_s256_b16(const S256Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef CTDD_S256_STAMPS
  unsigned long long stamps[6];
  const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime();
#endif
  B16_STAMP(0)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t wrow0 = (int64_t)blockIdx.x * TILE_ROWS + wave * 32;
  const unsigned char* aimg = a.tables + (size_t)S256 * S256 * 4;
  const int j = lane & 31, g = lane >> 5;          // matrix layout: row j of the wave, K half g
  const int j16 = lane & 15, q4 = lane >> 4;       // softmax layout: row j16 of the pass, quarter q4

  // big chunk c = K-steps 2c, 2c+1 of the hi plane: 16 pieces of 1 KiB, four per wave
  auto stage_chunk = [&](int c, int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave * 4 + i;
      const unsigned char* src = aimg + (size_t)(2 * c + (piece >> 3)) * S256_CHUNK_BYTES + (piece & 7) * 1024 + lane * 16;
      unsigned char* dst = smem + slot * SLOT_BYTES + piece * 1024;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                       (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
    }
  };
  int xj, xcur;                        // rate-state / current state of row (lane & 31): first, the invq gathers hang on them
  {
    const int64_t rj = wrow0 + j;
    const int64_t rjc = rj < a.R ? rj : a.R - 1;
    xcur = a.x[rjc];
    xj = (GENERAL && a.x_base) ? a.x_base[rjc] : xcur;
  }
  stage_chunk(0, slot_of(0));

  // ---- phase 1: two passes of 16 rows; lane (j16, q4) handles fragment pieces f = 2 (8 (q4 & 1) + i) + (q4 >> 1), i = 0..7
  // (piece f = the 8 consecutive s0 of K-step f >> 1, half f & 1) of row 16 p + j16.  Everything the phase reads is put in
  // flight at once: pass 0's rows by LDS-DMA, pass 1's rows through registers (they take the staging buffer's place once
  // pass 0 has been read out), both passes' invq rows (bf16 table) as fragment-shaped gathers.
  unsigned char* stg = smem + wave * 16384;
  constexpr float LOG2E = 1.4426950408889634f;
  u32x4 bw[16];                        // B fragments: bw[kk] = w[row j][16 kk + 8 g .. + 7] as 8 bf16 (after the lane swap)
  float zv = 1.0f;                     // Z of row (lane & 31)
  const unsigned short* invq16 = (const unsigned short*)(a.tables + S256_INVQ16_OFFSET);
  if constexpr (L16) {
    // 32 rows x 512 B: sixteen DMA instructions of two rows each; pass p = rows 16p .. 16p+15 at stg + 8192 p, a row's
    // 32 pieces XOR-swizzled by its number (source side)
    const unsigned short* lg = (const unsigned short*)a.logits;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = 2 * i + (lane >> 5);
      const int64_t row = wrow0 + r;
      const int64_t rowc = row < a.R ? row : a.R - 1;
      const unsigned char* src = (const unsigned char*)(lg + (size_t)rowc * S256) + (((lane & 31) ^ (r & 15)) << 4);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                       (void __attribute__((address_space(3)))*)(stg + i * 1024), 16, 0, 0);
    }
    asm volatile("" ::: "memory");                                   // (the gathers below stay behind the DMA issues: counted wait)
    xcur = min(max(xcur, 0), S256 - 1);
    xj = min(max(xj, 0), S256 - 1);
    u32x4 iqb[2][8];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int xp = __shfl(xj, j16 + 16 * p, WAVE);                 // rate-state of row 16 p + j16
      const unsigned short* qrow = invq16 + (size_t)xp * S256 + 128 * (q4 & 1) + 8 * (q4 >> 1);
#pragma unroll
      for (int i = 0; i < 8; ++i) iqb[p][i] = *(const u32x4*)(qrow + 16 * i);
    }
    // the rows do not hang on x: wait for them alone (the 16 gathers are the youngest operations), take max / exp / sum while
    // the gathers -- a second memory latency behind the states' -- are still in flight, multiply when they are there
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    B16_STAMP(5)
    float ev[2][64];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      u32x4 raw[8];
      const unsigned char* myrow = stg + p * 8192 + j16 * 512;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int f = 16 * (q4 & 1) + 2 * i + (q4 >> 1);             // fragment piece = 16-byte piece of the bf16 row
        raw[i] = *(const u32x4*)(myrow + ((f ^ j16) << 4));
      }
      float m0 = bf_lo(raw[0][0]), m1 = bf_hi(raw[0][0]);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int d = 0; d < 4; ++d) { m0 = fmaxf(m0, bf_lo(raw[i][d])); m1 = fmaxf(m1, bf_hi(raw[i][d])); }
      float mx = fmaxf(m0, m1);
      mx = quad_max(mx);
      const float ms = mx * LOG2E;
      float z0 = 0.0f, z1 = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float e0 = __builtin_amdgcn_exp2f(fmaf(bf_lo(raw[i][d]), LOG2E, -ms));
          const float e1 = __builtin_amdgcn_exp2f(fmaf(bf_hi(raw[i][d]), LOG2E, -ms));
          z0 += e0; z1 += e1;
          ev[p][8 * i + 2 * d] = e0; ev[p][8 * i + 2 * d + 1] = e1;
        }
      float z = z0 + z1;
      z = quad_sum(z);
      zv = ((q4 & 1) == p) ? z : zv;                                  // the lane that ends up with row 16 p + j16
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const u32x4 qb = iqb[p][i];
        u32x4 w;
#pragma unroll
        for (int d = 0; d < 4; ++d) w[d] = pack_bf16(ev[p][8 * i + 2 * d] * bf_lo(qb[d]), ev[p][8 * i + 2 * d + 1] * bf_hi(qb[d]));
        bw[8 * p + i] = w;
      }
  } else {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = wrow0 + r;
    const int64_t rowc = row < a.R ? row : a.R - 1;
    // LDS position `lane` of row r receives global 16-byte piece lane ^ r
    const unsigned char* src = (const unsigned char*)(a.logits + (size_t)rowc * S256) + ((lane ^ r) << 4);
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)(stg + r * 1024), 16, 0, 0);
  }
  f32x4 st1[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = wrow0 + 16 + r;
    const int64_t rowc = row < a.R ? row : a.R - 1;
    st1[r] = *(const f32x4*)(a.logits + (size_t)rowc * S256 + ((lane ^ r) << 2));
  }
  xcur = min(max(xcur, 0), S256 - 1);
  xj = min(max(xj, 0), S256 - 1);
  u32x4 iqb[2][8];                     // (native vector type: a struct-typed array here went to scratch memory)
  {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int xp = __shfl(xj, j16 + 16 * p, WAVE);                 // rate-state of row 16 p + j16
      const unsigned short* qrow = invq16 + (size_t)xp * S256 + 128 * (q4 & 1) + 8 * (q4 >> 1);
#pragma unroll
      for (int i = 0; i < 8; ++i) iqb[p][i] = *(const u32x4*)(qrow + 16 * i);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // rows, gathers (and chunk 0) have landed
  B16_STAMP(5)
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    f32x4 raw[16];
    {
      const unsigned char* myrow = stg + j16 * 1024;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int piece = 32 * (q4 & 1) + 4 * i + 2 * (q4 >> 1) + e;   // 16-byte piece 2 f + e of the row
          raw[2 * i + e] = *(const f32x4*)(myrow + ((piece ^ j16) << 4));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (p == 0) {                                                   // the staging buffer is free again: pass 1's rows take its place
#pragma unroll
      for (int r = 0; r < 16; ++r) *(f32x4*)(stg + r * 1024 + lane * 16) = st1[r];
    }
    float m0 = raw[0].x, m1 = raw[0].y, m2 = raw[0].z, m3 = raw[0].w;
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      m0 = fmaxf(m0, raw[k].x); m1 = fmaxf(m1, raw[k].y); m2 = fmaxf(m2, raw[k].z); m3 = fmaxf(m3, raw[k].w);
    }
    float mx = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    mx = quad_max(mx);
    const float ms = mx * LOG2E;
    float z0 = 0.0f, z1 = 0.0f, z2 = 0.0f, z3 = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      raw[k].x = __builtin_amdgcn_exp2f(fmaf(raw[k].x, LOG2E, -ms));
      raw[k].y = __builtin_amdgcn_exp2f(fmaf(raw[k].y, LOG2E, -ms));
      raw[k].z = __builtin_amdgcn_exp2f(fmaf(raw[k].z, LOG2E, -ms));
      raw[k].w = __builtin_amdgcn_exp2f(fmaf(raw[k].w, LOG2E, -ms));
      z0 += raw[k].x; z1 += raw[k].y; z2 += raw[k].z; z3 += raw[k].w;
    }
    float z = (z0 + z1) + (z2 + z3);
    z = quad_sum(z);
    zv = ((q4 & 1) == p) ? z : zv;                                  // the lane that ends up with row 16 p + j16
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 e0 = raw[2 * i], e1 = raw[2 * i + 1];
      const u32x4 qb = iqb[p][i];
      u32x4 w;
      w[0] = pack_bf16(e0.x * bf_lo(qb[0]), e0.y * bf_hi(qb[0]));
      w[1] = pack_bf16(e0.z * bf_lo(qb[1]), e0.w * bf_hi(qb[1]));
      w[2] = pack_bf16(e1.x * bf_lo(qb[2]), e1.y * bf_hi(qb[2]));
      w[3] = pack_bf16(e1.z * bf_lo(qb[3]), e1.w * bf_hi(qb[3]));
      bw[8 * p + i] = w;
    }
  }
  }
  // even 16-lane rows computed K-steps 0..7, odd rows K-steps 8..15; pass 0 sits in bw[0..7], pass 1 in bw[8..15].
  // v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second: afterwards lane
  // j16 + 16 q' holds row 16 (q' & 1) + j16, K half q' >> 1, K-steps 0..7 in bw[0..7] and 8..15 in bw[8..15].
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const auto r = __builtin_amdgcn_permlane16_swap(bw[i][d], bw[8 + i][d], false, false);
      bw[i][d] = r[0];
      bw[8 + i][d] = r[1];
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();          // every wave is done with its staging buffer: slots 0..3 are free; chunk 0 has landed
  stage_chunk(1, slot_of(1));
  stage_chunk(2, slot_of(2));
  stage_chunk(3, slot_of(3));
  stage_chunk(4, slot_of(4));
  B16_STAMP(1)

  // ---- phase 2: acc[m] (32 s x 32 rows) += A(kk, m) . B(kk): 64 groups of two matrix instructions
  f32x16 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.0f;
  bf16x8 fh[4][2];
  // forward-rate pieces RT0[x_j][s] of the epilogue in three groups of m-tiles (0-2, 3-5, 6-7): the first is requested under
  // the last chunk of the contraction (its B registers are free by then)
  const float* frow = a.RT0 + (size_t)xj * S256 + 4 * g;
  f32x4 fA[3][4], fB[3][4];
#define B16_LOAD_F(dst, m0, nm)                                                             \
  _Pragma("unroll") for (int mm_ = 0; mm_ < (nm); ++mm_)                                    \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) dst[mm_][q_] = *(const f32x4*)(frow + 32 * ((m0) + mm_) + 8 * q_);
  auto read_group = [&](auto GG) {
    constexpr int G = decltype(GG)::value, c = G / 8, kl = (G / 4) % 2, i = G % 4, buf = G % 4;
    const unsigned char* ab = smem + slot_of(c) * SLOT_BYTES + kl * 8192 + (g * S256 + j) * 16 + (2 * i) * 512;
    fh[buf][0] = *(const bf16x8*)(ab);
    fh[buf][1] = *(const bf16x8*)(ab + 512);
  };
  read_group(std::integral_constant<int, 0>{});
  read_group(std::integral_constant<int, 1>{});
  read_group(std::integral_constant<int, 2>{});
  auto kgroup = [&](auto GG) {
    constexpr int G = decltype(GG)::value, c = G / 8, i = G % 4, buf = G % 4, kk = G / 4;
    if constexpr (G % 8 == 5 && c + 1 < NCHUNK) {
      wait_vmcnt<mid_wait(c)>();                  // my four pieces of chunk c+1 have landed
      __builtin_amdgcn_s_barrier();               // ... and everyone's; chunk c-1 is out of use
      if constexpr (c >= 1 && c + 4 < NCHUNK) stage_chunk(c + 4, slot_of(c - 1));
    }
    if constexpr (G + 3 < 64) read_group(std::integral_constant<int, G + 3>{});
    if constexpr (G == 48) { B16_LOAD_F(fA, 0, 3) }      // (the B fragments of K-steps 0..11 are out of use: 48 registers)
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8 b = __builtin_bit_cast(bf16x8, bw[kk]);
    acc[2 * i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[buf][0], b, acc[2 * i], 0, 0, 0);
    acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[buf][1], b, acc[2 * i + 1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto kchunk = [&](auto CC) {
    constexpr int c = decltype(CC)::value;
    kgroup(std::integral_constant<int, 8 * c + 0>{});
    kgroup(std::integral_constant<int, 8 * c + 1>{});
    kgroup(std::integral_constant<int, 8 * c + 2>{});
    kgroup(std::integral_constant<int, 8 * c + 3>{});
    kgroup(std::integral_constant<int, 8 * c + 4>{});
    kgroup(std::integral_constant<int, 8 * c + 5>{});
    kgroup(std::integral_constant<int, 8 * c + 6>{});
    kgroup(std::integral_constant<int, 8 * c + 7>{});
  };
#define KC(n) kchunk(std::integral_constant<int, n>{});
  KC(0) KC(1) KC(2) KC(3) KC(4) KC(5) KC(6) KC(7)
#undef KC
  B16_STAMP(2)

  // ---- phase 3: lane (j,g) holds out[s][row j] for s = 32m + 8q + 4g + p, reg = 4q + p
  const bool corrector = GENERAL && (a.flags & CTDD_STEP_CORRECTOR);
  const bool crm = GENERAL && (a.flags & CTDD_STEP_CRM);                     // see steps_s256.hip: acc[x] + 1e-35 Z in the place of Z
  float norm = zv, addn = 0.0f;
  if (crm) {
    float ox = 0.0f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) ox = (32 * m + 8 * (r >> 2) + 4 * g + (r & 3) == xj) ? acc[m][r] : ox;
    ox = pair_sum(ox);
    addn = 1e-35f * zv;
    norm = ox + addn;
  }
  const float invz = 1.0f / norm;
  const float scale = a.beta * invz;                            // true rate = scale * r
  const int64_t myrow = wrow0 + j;
  const bool live = myrow < a.R;
  const float* crow = a.R0 + (size_t)xj * S256 + 4 * g;
  __builtin_amdgcn_s_barrier();                                 // every wave has left the A ring
  float T0 = 0.0f, T1 = 0.0f, T2 = 0.0f, T3 = 0.0f;     // (four running sums: a per-block sum here would be kept alive for the
                                                          //  prefix pass of phase 4 -- 32 registers across the whole epilogue)
  auto apply = [&](auto MM, const f32x4 (&f)[4]) {              // one m-tile: acc <- masked rates (units of beta / Z)
    constexpr int m = decltype(MM)::value;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 fq = f[q];
      float v0, v1, v2, v3;
      if (GENERAL) {
        v0 = (acc[m][4 * q + 0] + addn) * fq.x; v1 = (acc[m][4 * q + 1] + addn) * fq.y;
        v2 = (acc[m][4 * q + 2] + addn) * fq.z; v3 = (acc[m][4 * q + 3] + addn) * fq.w;
      } else {
        v0 = acc[m][4 * q + 0] * fq.x; v1 = acc[m][4 * q + 1] * fq.y; v2 = acc[m][4 * q + 2] * fq.z; v3 = acc[m][4 * q + 3] * fq.w;
      }
      if (corrector) {                                          // r_s += Z * R[x][s]  (Z: the normaliser in use)
        const float4 c = *(const float4*)(crow + 32 * m + 8 * q);
        v0 = fmaf(norm, c.x, v0); v1 = fmaf(norm, c.y, v1); v2 = fmaf(norm, c.z, v2); v3 = fmaf(norm, c.w, v3);
      }
      T0 += v0; T1 += v1; T2 += v2; T3 += v3;
      acc[m][4 * q + 0] = v0; acc[m][4 * q + 1] = v1; acc[m][4 * q + 2] = v2; acc[m][4 * q + 3] = v3;
      if (GENERAL && a.out_rates && live)
        *(float4*)(a.out_rates + (size_t)myrow * S256 + 32 * m + 8 * q + 4 * g) =
            make_float4(scale * v0, scale * v1, scale * v2, scale * v3);
    }
  };
  B16_LOAD_F(fB, 3, 3)
  __builtin_amdgcn_sched_barrier(0);
  apply(std::integral_constant<int, 0>{}, fA[0]);
  apply(std::integral_constant<int, 1>{}, fA[1]);
  apply(std::integral_constant<int, 2>{}, fA[2]);
  __builtin_amdgcn_sched_barrier(0);
  B16_LOAD_F(fA, 6, 2)
  __builtin_amdgcn_sched_barrier(0);
  apply(std::integral_constant<int, 3>{}, fB[0]);
  apply(std::integral_constant<int, 4>{}, fB[1]);
  apply(std::integral_constant<int, 5>{}, fB[2]);
  __builtin_amdgcn_sched_barrier(0);
  apply(std::integral_constant<int, 6>{}, fA[0]);
  apply(std::integral_constant<int, 7>{}, fA[1]);
#undef B16_LOAD_F
  float T = (T0 + T1) + (T2 + T3);
  T = pair_sum(T);
  if (GENERAL && a.out_rates && !a.out_x) return;
  B16_STAMP(3)

  const float Lam = scale * T * a.h;
  const bool ordinal = a.flags & CTDD_STEP_ORDINAL;
  const uint64_t rngrow = (uint64_t)(live ? myrow : a.R - 1);
  int jump = 0, njumps = 0;               // njumps: jump events drawn for this dimension (sum_s k_s)
  const bool dense = Lam > SUPERPOSE_MAX_LAMBDA;
  const bool superp = Lam > 0.0f && Lam <= SUPERPOSE_MAX_LAMBDA;

  // ---- dense rows (Lambda > 64) first, on the raw rates: sub-blocks of 4 consecutive destinations (draw.hpp:
  // subblock_draw), the lane owns sub-blocks b = 8m + 2q + g.  The rates of four m-tiles at a time are parked in the
  // wave's own LDS region so that the draw is a rolled loop.
  if (__any(dense)) {
    float4* rl4 = (float4*)(smem + wave * 16384) + lane;       // [block 16][lane 64] float4: four m-tiles at a time
    const float sh = scale * a.h;
    int cnt = 0;
    long long jl = 0;
    for (int m = 0; m < 8; ++m) {
      if ((m & 3) == 0) {                                       // park m-tiles m .. m+3 (static register indices)
        if (m == 0) {
#pragma unroll
          for (int mm = 0; mm < 4; ++mm)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              rl4[(4 * mm + q) * 64] = make_float4(acc[mm][4 * q], acc[mm][4 * q + 1], acc[mm][4 * q + 2], acc[mm][4 * q + 3]);
        } else {
#pragma unroll
          for (int mm = 0; mm < 4; ++mm)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              rl4[(4 * mm + q) * 64] = make_float4(acc[4 + mm][4 * q], acc[4 + mm][4 * q + 1], acc[4 + mm][4 * q + 2], acc[4 + mm][4 * q + 3]);
        }
        asm volatile("" ::: "memory");                          // (no store-to-load forwarding: it would index acc[] at run time)
      }
      if (dense) {
        const int mm = m & 3;
        // each lane builds Philox block 2m+g and trades it with its partner, so both see blocks 2m and 2m+1
        const u4 mine = philox_row(a.seed, a.offset, rngrow, DENSE_DRAW0 + (uint32_t)(2 * m + g));
        u4 lo, hi;                                                         // blocks 2m, 2m+1
        pair_values(mine.x, lo.x, hi.x); pair_values(mine.y, lo.y, hi.y);
        pair_values(mine.z, lo.z, hi.z); pair_values(mine.w, lo.w, hi.w);
        const uint32_t w0 = g == 0 ? lo.x : lo.y, w1 = g == 0 ? lo.z : lo.w;
        const uint32_t w2 = g == 0 ? hi.x : hi.y, w3 = g == 0 ? hi.z : hi.w;
        const float4 v0 = rl4[(4 * mm + 0) * 64], v1 = rl4[(4 * mm + 1) * 64], v2 = rl4[(4 * mm + 2) * 64],
                     v3 = rl4[(4 * mm + 3) * 64];
        const int b0 = 8 * m + g;
        cnt += min(subblock_draw(v0.x, v0.y, v0.z, v0.w, sh, u01(w0), a.seed, a.offset, rngrow, b0, xj, 4, &jl), 1 << 20);
        cnt += min(subblock_draw(v1.x, v1.y, v1.z, v1.w, sh, u01(w1), a.seed, a.offset, rngrow, b0 + 2, xj, 4, &jl), 1 << 20);
        cnt += min(subblock_draw(v2.x, v2.y, v2.z, v2.w, sh, u01(w2), a.seed, a.offset, rngrow, b0 + 4, xj, 4, &jl), 1 << 20);
        cnt += min(subblock_draw(v3.x, v3.y, v3.z, v3.w, sh, u01(w3), a.seed, a.offset, rngrow, b0 + 6, xj, 4, &jl), 1 << 20);
      }
      asm volatile("" ::: "memory");
    }
    if (dense) {
      cnt = pair_sum(cnt);
      {
        unsigned l0, h0, l1, h1;
        pair_values((unsigned)(unsigned long long)jl, l0, h0);
        pair_values((unsigned)((unsigned long long)jl >> 32), l1, h1);
        jl = (long long)(((unsigned long long)l1 << 32) | l0) + (long long)(((unsigned long long)h1 << 32) | h0);
      }
      jl = jl > S256 ? S256 : (jl < -S256 ? -S256 : jl);        // |jump| >= S - 1 saturates the state clamp either way
      jump = (ordinal || cnt <= 1) ? (int)jl : 0;
      njumps = cnt;
    }
  }

  // ---- superposition rows: K ~ Poisson(Lambda), then K destinations by inverse CDF in s order
  int K = 0;
  PhiloxStream rng(a.seed, a.offset, rngrow, 0u);               // both lanes of the pair: same stream
  if (superp) {
    K = poisson_row(Lam, rng);
    njumps = K;
  }
  const bool picks = superp && K > 0 && (ordinal || K == 1);
  if (__any(picks)) {
    // cumulative rates in destination order (block of 8 = g0's four then g1's four), in place over the rates.
    // Lane-local group i = the lane's blocks 2i, 2i+1 (8 values): its start goes to LDS in fp32, its 8 cumulative values
    // as bf16 offsets from that start; the group's last value stays in acc[i >> 1][8 (i & 1) + 7].
    uint4* relv = (uint4*)(smem + wave * 16384) + lane;        // [group 16][lane 64] 8 bf16
    float* gsv = (float*)(smem + 65536 + wave * 4096) + lane;  // [group 16][lane 64] f32
    float run = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float gstart = 0.0f;
      float rel[8];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int m = i >> 1, q = 2 * (i & 1) + bb;
        const float v0 = acc[m][4 * q], v1 = acc[m][4 * q + 1], v2 = acc[m][4 * q + 2], v3 = acc[m][4 * q + 3];
        const float mine = (v0 + v1) + (v2 + v3);
        unsigned lo_, hi_;
        pair_values(__float_as_uint(mine), lo_, hi_);
        const float first = __uint_as_float(lo_), second = __uint_as_float(hi_);   // the block's g = 0 / g = 1 four
        float c = run + (g == 0 ? 0.0f : first);
        if (bb == 0) gstart = c;
        c += v0; rel[4 * bb + 0] = c - gstart;
        c += v1; rel[4 * bb + 1] = c - gstart;
        c += v2; rel[4 * bb + 2] = c - gstart;
        c += v3; rel[4 * bb + 3] = c - gstart;
        acc[m][4 * q + 3] = c;                                    // (only the bb = 1 value is read again)
        run += first + second;
      }
      uint4 pk;
      pk.x = pack_bf16(rel[0], rel[1]); pk.y = pack_bf16(rel[2], rel[3]);
      pk.z = pack_bf16(rel[4], rel[5]); pk.w = pack_bf16(rel[6], rel[7]);
      relv[i * 64] = pk;
      gsv[i * 64] = gstart;
    }
    if (picks) {
      // A pick is kept to ~45 instructions: no compare-to-mask / mask-to-register pairs (each costs its own wait states) --
      //  * group ends <= target: the sign bits of (target - end) shifted into one word, counted at the end (all values are
      //    finite and >= 0, and x - x = +0, so "sign clear" is exactly "end <= target");
      //  * the 7 bf16 offsets <= dlt: non-negative floats order as their bit patterns and a bf16 pattern has no low half, so
      //    b <= dlt  <=>  bits16(b) <= bits(dlt) >> 16 as integers: four packed 16-bit subtractions, sign bits counted
      //    (dlt < 0 -- the target lies in the partner's half of the block -- becomes -1: below every pattern);
      //  * the partner's count by v_permlane32_swap.
      using i16x2 = __attribute__((ext_vector_type(2))) short;
      for (int d = 0; d < K; ++d) {
        const float target = rng.next() * T;
        unsigned signs = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          signs = __builtin_amdgcn_alignbit(signs, __float_as_uint(target - acc[i >> 1][8 * (i & 1) + 7]), 31);
        const int nb = 16 - __builtin_popcount(signs);           // groups that end at or before the target
        int cnt = 8 * nb;
        if (nb < 16) {
          const uint4 pk = relv[nb * 64];
          const float dlt = target - gsv[nb * 64];
          const int di = max((int)__float_as_uint(dlt), -1) >> 16;
          const unsigned dd = ((unsigned)di & 0xFFFFu) * 0x10001u;
          const i16x2 d2 = __builtin_bit_cast(i16x2, dd);
          const unsigned s0 = __builtin_bit_cast(unsigned, d2 - __builtin_bit_cast(i16x2, pk.x));
          const unsigned s1 = __builtin_bit_cast(unsigned, d2 - __builtin_bit_cast(i16x2, pk.y));
          const unsigned s2 = __builtin_bit_cast(unsigned, d2 - __builtin_bit_cast(i16x2, pk.z));
          const unsigned s3 = __builtin_bit_cast(unsigned, d2 - __builtin_bit_cast(i16x2, pk.w));
          const unsigned neg = (s0 & 0x80008000u) | ((s1 & 0x80008000u) >> 1) | ((s2 & 0x80008000u) >> 2) | ((s3 & 0x00008000u) >> 3);
          cnt += 7 - __builtin_popcount(neg);
        }
        cnt = pair_sum(cnt);
        jump += min(cnt, S256 - 1) - xj;
      }
    }
  }
  B16_STAMP(4)
#ifdef CTDD_S256_STAMPS
  if (lane == 0) {
    unsigned long long* o = (unsigned long long*)a.out_changed + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int i = 0; i < 5; ++i) o[i] = stamps[i];
    o[5] = __builtin_amdgcn_s_memrealtime();
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    o[6] = rt0_;   (void)xcc; (void)hwid;
    o[7] = ((unsigned long long)xcc << 32) | hwid;
  }
  return;
#endif
  bool moved = false;
  if (live && g == 0) {
    const int xn = min(max(xcur + jump, 0), S256 - 1);
    a.out_x[myrow] = xn;
    moved = (a.flags & CTDD_STEP_COUNT_RAW) ? (jump != 0) : (xn != xcur);
  }
  if (a.out_changed) {
    // ONE atomic per workgroup and counter: the waves leave their counts in their own (by now idle) LDS regions and wave 0
    // adds them up behind a barrier.  (One atomic per wave -- 6 272 adds on one address per 256-sample launch, ~12 ns each
    // at the memory side -- took as long as the rest of the kernel.)
    const bool cj = a.flags & CTDD_STEP_COUNT_JUMPS;
    const int nmoved = __builtin_popcountll(__ballot(moved));
    const int n1 = cj ? __builtin_popcountll(__ballot(live && g == 0 && njumps > 0)) : 0;   // sampling.py:489-495: dimensions with
    const int n2 = cj ? __builtin_popcountll(__ballot(live && g == 0 && njumps > 1)) : 0;   // >= 1 and with > 1 jump events
    int* slot = (int*)(smem + wave * 16384);
    if (lane == 0) { slot[0] = nmoved; slot[1] = n1; slot[2] = n2; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x < 3) {
      int tot = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) tot += ((const int*)(smem + w * 16384))[threadIdx.x];
      if (tot && (threadIdx.x == 0 || cj)) atomicAdd(a.out_changed + threadIdx.x, tot);
    }
  }
}

}  // namespace b16

int launch_tauleap_s256_b16(const S256Args& a, hipStream_t stream) {
  const int64_t grid = (a.R + b16::TILE_ROWS - 1) / b16::TILE_ROWS;
  CTDD_REQUIRE(grid < (1ll << 31), CTDD_ERANGE, "too many rows");
  const bool general = a.x_base || a.out_rates || (a.flags & (CTDD_STEP_CORRECTOR | CTDD_STEP_CRM));
  const bool l16 = a.flags & CTDD_STEP_LOGITS_BF16;
  static bool attr_done4[4][16] = {};
  auto go = [&](auto kernel, int slot) {
    ensure_lds_ceiling((const void*)kernel, attr_done4[slot]);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(256), b16::LDS_BYTES, stream, a);
  };
  if (general && l16) go(b16::k_tauleap_s256_b16<true, true>, 0);
  else if (general) go(b16::k_tauleap_s256_b16<true, false>, 1);
  else if (l16) go(b16::k_tauleap_s256_b16<false, true>, 2);
  else go(b16::k_tauleap_s256_b16<false, false>, 3);
  return finish_launch("k_tauleap_s256_b16");
}

}  // namespace ctdd
