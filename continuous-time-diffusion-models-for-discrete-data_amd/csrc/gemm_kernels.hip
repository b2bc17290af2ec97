// gemm_kernels.hip -- plain bf16 GEMM for the hollow transformer's linear layers (reference: every nn.Linear of
// lib/networks/hollow_networks.py:90-447 in forward, and its data gradient dX = dY W in backward).
//
//   out[M][N] = act( [A_0 | A_1 | A_2][M][nseg K] . W[N][nseg K]^T + bias ) + res          (fp32 accumulate)
//
// The U-Net's slab-convolution kernel run as a GEMM (k_conv_patch with 1x1 segments) carries its halo / unit / tap machinery
// along: 1.5-2.8 TB/s and 70-210 TFLOP/s on these skinny shapes (rows = batch x tokens ~ 3e4, K = 128 .. 1024, N = 128 .. 1024),
// which are bandwidth-bound (K = 128: 2 flops per output byte).  This kernel is only the GEMM: a workgroup owns a
// (64 TM) x (64 TN) output tile, its four waves a 2 x 2 arrangement of (32 TM) x (32 TN) sub-tiles on
// v_mfma_f32_32x32x16_bf16; K in chunks of 64 through double-buffered LDS (rows padded to 72 bf16), TWO chunks of 16-byte global
// loads in flight in registers (with one, a chunk's loads had 256 cycles of matrix work to land behind: every chunk exposed the
// whole memory latency, 3.5 us per chunk measured), one barrier per chunk; the epilogue goes through a wave-private fp32 image
// in LDS so that a lane stores 16 bytes of one row (straight from the accumulators a lane holds one output COLUMN: 4- or
// 2-byte stores ran at 2.2 / 1.2 TB/s).  Up to three A segments against a concatenated weight serve the hi / lo
// split-precision mode of the inference engine ([x_hi | x_lo | x_hi] . [w_hi | w_hi | w_lo]).
// Tried and dropped: an A-stationary variant for K <= 256 (a workgroup keeps its 64 rows of A in LDS and walks the N tiles, next
// tile's weights in registers): 21.0 / 46.9 us against 18.7 / 41.6 us at 28800 x 128 x {384, 1024} -- 450 workgroups hide the
// per-tile latencies worse than 1800 independent tiles do; no difference at K = 256.
#include "common.hpp"

namespace ctdd {

struct GemmArgs {
  const unsigned short* a[3]; int nseg;        // A segments, each [M][K] row-major bf16
  const unsigned short* w;                     // [N][nseg * K] bf16
  const float* bias;                           // [N] or null
  const float* res;                            // [M][N] fp32 or null (added after the activation)
  float* out_f32; unsigned short* out_hi; unsigned short* out_lo;   // any subset: fp32, bf16, bf16(v - hi)
  int M, N, K, act;                            // act 0 none, 1 ReLU, 2 GELU (erf)
  // training epilogues (hollow_train_kernels.hip's streams and rules, so that the separate passes they replace and their backward
  // regenerate / read the same masks):
  float drop_p; const uint64_t* rng; uint64_t layer;   // drop_p > 0: dropout of act(.) BEFORE the residual, keep flags
                                               // Philox(rng[0], rng[1] * 4096 + layer, element / 4) as k_hollow_dropout / k_hollow_act
  const unsigned short* mask_u;                // [M][N] bf16 or null: v <- mask_u != 0 ? v / (1 - drop_p) : 0 (the ReLU + dropout
                                               // backward of k_hollow_relu_bf16: the saved output is the mask), no random numbers
};

using gbf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using gf32x16 = __attribute__((ext_vector_type(16))) float;
using gu32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ inline unsigned short g_bf16(float v) {
  using v2f = __attribute__((ext_vector_type(2))) float;
  using v2b = __attribute__((ext_vector_type(2))) __bf16;
  v2f t = {v, 0.0f};
  return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(t, v2b)) & 0xFFFFu);
}

// KC = 32 with one chunk in flight: 128 x 128 tiles at three workgroups per CU -- best for the K = 128 layers with wide N (the A
// tile is re-read once per 128 columns); KC = 64 with two chunks in flight on 64 x 128 tiles -- best from K = 256 up.
// SPLIT: the hi / lo split product of the inference engine, [x_hi | x_lo | x_hi] . [w_hi | w_hi | w_lo] (three segments whose first
// and third A pointers coincide), with each of its FOUR distinct operand tiles staged once per K chunk: x_hi and w_hi served two of
// the three products from separate passes over global memory and LDS before -- a third of the loads, LDS stores and fragment reads
// were repeats, and a barrier pair covered one product's chunk instead of all three.
template <int TM, int TN, int KC, bool DEEP, bool SPLIT = false>
__global__ __launch_bounds__(256) void k_gemm_bf16(const GemmArgs a) {
  constexpr int BM = 64 * TM, BN = 64 * TN, LDK = KC + 8, NP = SPLIT ? 2 : 1;   // NP: operand planes (hi, lo)
  constexpr int TPR = KC / 8, RPP = 256 / TPR;                  // threads per staged row piece, rows per pass
  constexpr int AV = BM / RPP, WV = BN / RPP;                   // 16-byte vectors per thread, plane and chunk
  constexpr int ILD = 32 * TN + 4;                               // epilogue image row (floats)
  constexpr int STAGE_B = 2 * NP * (BM + BN) * LDK * 2, IMG_B = 4 * 32 * ILD * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_B > IMG_B ? STAGE_B : IMG_B];
  unsigned short (*As)[NP * BM * LDK] = (unsigned short (*)[NP * BM * LDK])smem;                       // [buffer][plane][row][k]
  unsigned short (*Ws)[NP * BN * LDK] = (unsigned short (*)[NP * BN * LDK])(smem + 2 * NP * BM * LDK * 2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int K = a.K, Ktot = a.nseg * K, cps = K / KC, nchunks = SPLIT ? cps : a.nseg * cps;
  const int vrow = tid / TPR, vc8 = (tid % TPR) * 8;            // staging: TPR threads per row piece of KC bf16
  gu32x4 ra[DEEP ? 2 : 1][NP * AV], rw[DEEP ? 2 : 1][NP * WV];  // chunks in flight
  auto fetch = [&](int c, gu32x4 (&pa)[NP * AV], gu32x4 (&pw)[NP * WV]) {
    const int seg = SPLIT ? 0 : c / cps, kk = (c - seg * cps) * KC;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      const unsigned short* ap = a.a[SPLIT ? pl : seg];           // SPLIT: x_hi, x_lo
      const int wcol = SPLIT ? 2 * pl * K : seg * K;              // SPLIT: w_hi at columns [0, K), w_lo at [2K, 3K) of the packed rows
#pragma unroll
      for (int u = 0; u < AV; ++u) {
        const int64_t row = m0 + vrow + RPP * u;
        pa[pl * AV + u] = row < a.M ? *(const gu32x4*)(ap + (size_t)row * K + kk + vc8) : gu32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < WV; ++u) {
        const int n = n0 + vrow + RPP * u;
        pw[pl * WV + u] = n < a.N ? *(const gu32x4*)(a.w + (size_t)n * Ktot + wcol + kk + vc8) : gu32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  auto stash = [&](int buf, const gu32x4 (&pa)[NP * AV], const gu32x4 (&pw)[NP * WV]) {
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
      for (int u = 0; u < AV; ++u) *(gu32x4*)(&As[buf][pl * BM * LDK + (vrow + RPP * u) * LDK + vc8]) = pa[pl * AV + u];
#pragma unroll
      for (int u = 0; u < WV; ++u) *(gu32x4*)(&Ws[buf][pl * BN * LDK + (vrow + RPP * u) * LDK + vc8]) = pw[pl * WV + u];
    }
  };
  gf32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  auto compute = [&](int buf) {
    const unsigned short* Ab = &As[buf][(wm * 32 * TM + li) * LDK + 8 * kh];
    const unsigned short* Wb = &Ws[buf][(wn * 32 * TN + li) * LDK + 8 * kh];
#pragma unroll
    for (int s_ = 0; s_ < KC / 16; ++s_) {
      gbf16x8 af[NP][TM], bf[NP][TN];
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[pl][i] = *(const gbf16x8*)(Ab + (size_t)pl * BM * LDK + (size_t)i * 32 * LDK + 16 * s_);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[pl][j] = *(const gbf16x8*)(Wb + (size_t)pl * BN * LDK + (size_t)j * 32 * LDK + 16 * s_);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
          if constexpr (SPLIT) {                                 // + x_lo w_hi + x_hi w_lo  (x_lo w_lo ~ 2^-17 of the product: dropped)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
          }
        }
    }
  };
  // chunk c lives in registers set c & 1 until it is stashed into LDS buffer c & 1; chunk c + 2 is requested before chunk c is
  // worked on, so a chunk's loads have the matrix work of two chunks (and, at K = 128, nothing else at all) to land behind
  if (DEEP) {
    fetch(0, ra[0], rw[0]);
    if (nchunks > 1) fetch(1, ra[DEEP], rw[DEEP]);
    stash(0, ra[0], rw[0]);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {                      // nchunks is even (the launcher picks KC so): no exit in mid-body --
      if (c + 2 < nchunks) fetch(c + 2, ra[0], rw[0]);         // with one, the compiler kept two copies of the accumulators
      compute(0);
      stash(1, ra[DEEP], rw[DEEP]);
      __syncthreads();
      if (c + 3 < nchunks) fetch(c + 3, ra[DEEP], rw[DEEP]);
      compute(1);
      if (c + 2 < nchunks) stash(0, ra[0], rw[0]);
      __syncthreads();
    }
  } else {
    fetch(0, ra[0], rw[0]);
    stash(0, ra[0], rw[0]);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks) fetch(c + 1, ra[0], rw[0]);
      compute(c & 1);
      if (c + 1 < nchunks) stash((c & 1) ^ 1, ra[0], rw[0]);
      __syncthreads();
    }
  }
  // Epilogue through a wave-private fp32 image in the (now idle) staging memory: the accumulators hold one output COLUMN per
  // lane (a direct store is 4 or 2 bytes per lane: 2.2 TB/s in fp32, 1.2 in bf16 measured); read back row-major, a lane owns
  // four consecutive columns -> 16-byte residual loads and stores (8-byte in bf16).
  float* img = (float*)smem + wave * (32 * ILD);
  const float inv_keep = a.drop_p > 0.0f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const uint64_t dseed = (a.drop_p > 0.0f && !a.mask_u) ? a.rng[0] : 0, dctr = (a.drop_p > 0.0f && !a.mask_u) ? a.rng[1] * 4096u + a.layer : 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) img[((r & 3) + 8 * (r >> 2) + 4 * kh) * ILD + 32 * j + li] = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4 * TN; ++q) {
      const int v = lane + 64 * q, row = v / (8 * TN), c4 = (v % (8 * TN)) * 4;
      const int col = n0 + 32 * wn * TN + c4;
      const int64_t grow = m0 + 32 * (wm * TM + i) + row;
      const float4 t = *(const float4*)(img + row * ILD + c4);
      if (grow >= a.M || col >= a.N) continue;
      float x[4] = {t.x, t.y, t.z, t.w};
      if (a.bias) { const float4 b4 = *(const float4*)(a.bias + col); x[0] += b4.x; x[1] += b4.y; x[2] += b4.z; x[3] += b4.w; }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (a.act == 1) x[e] = fmaxf(x[e], 0.0f);
        else if (a.act == 2) x[e] = 0.5f * x[e] * (1.0f + erff(x[e] * 0.70710678118654752f));
      }
      const size_t o = (size_t)grow * a.N + col;
      if (a.mask_u) {                                            // (uniform branches: kernel arguments)
        const uint2 mu = *(const uint2*)(a.mask_u + o);
        x[0] = (mu.x & 0x7FFFu) ? x[0] * inv_keep : 0.0f; x[1] = (mu.x & 0x7FFF0000u) ? x[1] * inv_keep : 0.0f;
        x[2] = (mu.y & 0x7FFFu) ? x[2] * inv_keep : 0.0f; x[3] = (mu.y & 0x7FFF0000u) ? x[3] * inv_keep : 0.0f;
      } else if (a.drop_p > 0.0f) {
        const u4 r = philox_row(dseed, dctr, (uint64_t)(o >> 2), 0x44524F50u);
        x[0] = u01(r.x) >= a.drop_p ? x[0] * inv_keep : 0.0f; x[1] = u01(r.y) >= a.drop_p ? x[1] * inv_keep : 0.0f;
        x[2] = u01(r.z) >= a.drop_p ? x[2] * inv_keep : 0.0f; x[3] = u01(r.w) >= a.drop_p ? x[3] * inv_keep : 0.0f;
      }
      if (a.res) { const float4 r4 = *(const float4*)(a.res + o); x[0] += r4.x; x[1] += r4.y; x[2] += r4.z; x[3] += r4.w; }
      if (a.out_f32) *(float4*)(a.out_f32 + o) = make_float4(x[0], x[1], x[2], x[3]);
      if (a.out_hi) {
        const unsigned short h0 = g_bf16(x[0]), h1 = g_bf16(x[1]), h2 = g_bf16(x[2]), h3 = g_bf16(x[3]);
        *(uint2*)(a.out_hi + o) = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
        if (a.out_lo) {
          const unsigned short l0 = g_bf16(x[0] - __uint_as_float((unsigned)h0 << 16)), l1 = g_bf16(x[1] - __uint_as_float((unsigned)h1 << 16)),
                               l2 = g_bf16(x[2] - __uint_as_float((unsigned)h2 << 16)), l3 = g_bf16(x[3] - __uint_as_float((unsigned)h3 << 16));
          *(uint2*)(a.out_lo + o) = make_uint2((unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}


}  // namespace ctdd
using namespace ctdd;

extern "C" int ctdd_gemm_bf16(const void* args_, void* stream) {
  const GemmArgs& a = *(const GemmArgs*)args_;
  CTDD_REQUIRE(a.nseg >= 1 && a.nseg <= 3 && a.w && (a.out_f32 || a.out_hi), CTDD_EINVAL, "gemm: bad arguments");
  for (int i = 0; i < a.nseg; ++i) CTDD_REQUIRE(a.a[i], CTDD_EINVAL, "gemm: null segment %d", i);
  CTDD_REQUIRE(a.M > 0 && a.N > 0 && a.N % 4 == 0 && a.K > 0 && a.K % 64 == 0, CTDD_ERANGE, "gemm: M=%d N=%d K=%d (K %% 64 == 0, N %% 4 == 0)", a.M, a.N, a.K);
  CTDD_REQUIRE(a.act >= 0 && a.act <= 2, CTDD_EINVAL, "gemm: act %d", a.act);
  CTDD_REQUIRE(a.drop_p >= 0.0f && a.drop_p < 1.0f && (a.drop_p == 0.0f || a.mask_u || a.rng), CTDD_EINVAL, "gemm: dropout %g without a stream", (double)a.drop_p);
  hipStream_t st = (hipStream_t)stream;
  // Tile / chunk selection from measurements (scratch bench over the training and inference shapes, rows ~ 3e4):
  //   N <= 128               64 x 128 tiles, two chunks of 64 in flight            (K = 1024: 21.5 us vs 29.3 with 128 x 128 tiles)
  //   K' <= 128, N <= 512    128 x 128 tiles, chunks of 32, one in flight           (qkv 16.8 vs 18.0)
  //   K' <= 256, wider N     64 x 128 tiles, chunks of 32                           (fc1 / du 31.1 vs 36.2)
  //   K' >= 512              128 x 128 tiles, two chunks of 64 (32) in flight       (the hi / lo split products: 450-500 TFLOP/s)
  // The two-in-flight loop needs an even chunk count: K % 64 == 0 makes it so at chunks of 32, at 64 only for even nseg K / 64.
  static const bool no_split = [] { const char* e = getenv("CTDD_GEMM_NO_SPLIT"); return e && e[0] == '1'; }();    // (A/B: three separate segments)
  if (!no_split && a.nseg == 3 && a.a[0] == a.a[2] && a.a[0] != a.a[1] && a.N > 64) {
    // the hi / lo split product: four operand tiles per K chunk instead of six (k_gemm_bf16<..., SPLIT>); K % 64 == 0 makes
    // the chunk count of 32-wide chunks even, as the two-in-flight loop needs
    // (64 x 128 tiles, two workgroups per CU, for K <= 128 -- four chunks: the maze sampler 12.97 -> 13.19 k sample-steps/s; 128 x 128
    //  from K = 256 up: the MNIST hollow sampler 4.34 -> 4.59 k)
    if (a.N <= 128 || a.K <= 128) hipLaunchKernelGGL((k_gemm_bf16<1, 2, 32, true, true>), dim3((unsigned)((a.M + 63) / 64), (a.N + 127) / 128), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_gemm_bf16<2, 2, 32, true, true>), dim3((unsigned)((a.M + 127) / 128), (a.N + 127) / 128), dim3(256), 0, st, a);
    return finish_launch("k_gemm_bf16<split>");
  }
  const int Kt = a.nseg * a.K;
  const bool even64 = (Kt / 64) % 2 == 0;
  const dim3 g11((unsigned)((a.M + 63) / 64), (a.N + 63) / 64), g12((unsigned)((a.M + 63) / 64), (a.N + 127) / 128),
             g22((unsigned)((a.M + 127) / 128), (a.N + 127) / 128);
  if (a.N <= 64) {
    hipLaunchKernelGGL((k_gemm_bf16<1, 1, 32, true>), g11, dim3(256), 0, st, a);
  } else if (a.N <= 128) {
    if (even64) hipLaunchKernelGGL((k_gemm_bf16<1, 2, 64, true>), g12, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_gemm_bf16<1, 2, 32, true>), g12, dim3(256), 0, st, a);
  } else if (Kt <= 128 && a.N <= 512) {
    hipLaunchKernelGGL((k_gemm_bf16<2, 2, 32, false>), g22, dim3(256), 0, st, a);
  } else if (Kt <= 256) {
    hipLaunchKernelGGL((k_gemm_bf16<1, 2, 32, true>), g12, dim3(256), 0, st, a);
  } else {
    if (even64) hipLaunchKernelGGL((k_gemm_bf16<2, 2, 64, true>), g22, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_gemm_bf16<2, 2, 32, true>), g22, dim3(256), 0, st, a);
  }
  return finish_launch("k_gemm_bf16");
}
