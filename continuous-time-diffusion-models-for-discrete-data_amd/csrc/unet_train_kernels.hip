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
  int nchunks;       // chunks in all; grid_x workgroups share them
  int grid_x;        // M-split of this entry (<= the launch's grid.x)
  int tap;           // WG_3x3_S2: the tap this entry computes (nine entries per stride-2 convolution)
  float* gb;         // [N] fp32 or null: += column sums of dy over all pixels (the bias gradient), float atomics
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

// Matrix loop of one chunk (bf16): `nsteps` 16-position steps of NT + 1 transposed fragments and NT matrix instructions.
// The wave arrangement is a template parameter so that the LDS bytes per step are constants: the loop runs the steps in
// PAIRS from one set of running addresses -- the odd step's reads carry the step stride in the instruction's offset field --
// with two fragment sets (even / odd steps) that are each re-loaded where their previous contents were last used, so nothing
// is copied between registers.  (The first version rotated `next -> current` fragment registers and re-added the step
// offset per read: 40 register copies and 40 address adds per step, 17 vector instructions per matrix instruction;
// rocprofv3 SQ_INSTS_VALU / SQ_INSTS_MFMA.)  The taps go in groups of three: matrix instructions of group g, then the
// NEXT step's reads of group g -- at every wait the reads that may stay in flight number 12 or 14, which the 4-bit LDS
// counter can express (all twenty reads of a step issued in one block put 20 younger reads behind the ones a wait is for:
// the counter saturates and the wave waits for six reads it does not need yet, behind the other three waves' queues --
// measured 1.32 -> 1.80 ms for the 3x3 table of the MNIST step).
template <int NT, int NWN>
__device__ __attribute__((always_inline)) inline void wgrad_steps(const unsigned (&la)[2], const unsigned (&lb)[NT][2], int nsteps,
                                                                   f32x16 (&acc)[NT]) {
  constexpr unsigned STY = 16u * NWN * 64u, STX = 16u * (4 / NWN) * 64u;     // LDS bytes per 16-position step (dY / X tile)
  constexpr int GS = NT == 9 ? 3 : NT, NG = NT / GS;
  typedef __attribute__((address_space(3))) s16x4* lptr;
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  auto rd = [](unsigned addr0, unsigned addr1, unsigned imm) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(uintptr_t)(addr0 + imm));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(uintptr_t)(addr1 + imm));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  unsigned y0 = la[0], y1 = la[1], x0[NT], x1[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { x0[t] = lb[t][0]; x1[t] = lb[t][1]; }
  bf16x8 a0 = rd(y0, y1, 0), b0[NT], a1 = a0, b1[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { b0[t] = rd(x0[t], x1[t], 0); b1[t] = b0[t]; }
  int s = 0;
  for (; s + 2 <= nsteps; s += 2) {
    // even step on set 0; set 1 <- step s + 1
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = gq * GS; t < (gq + 1) * GS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0[t], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (gq == 0) a1 = rd(y0, y1, STY);
#pragma unroll
      for (int t = gq * GS; t < (gq + 1) * GS; ++t) b1[t] = rd(x0[t], x1[t], STX);
    }
    // odd step on set 1; set 0 <- step s + 2 (or, behind the last pair, the same positions again: not used)
    const bool more = s + 2 < nsteps;
    const unsigned iy = more ? 2 * STY : 0u, ix = more ? 2 * STX : 0u;
    y0 += iy; y1 += iy;
#pragma unroll
    for (int t = 0; t < NT; ++t) { x0[t] += ix; x1[t] += ix; }
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = gq * GS; t < (gq + 1) * GS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1[t], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (gq == 0) a0 = rd(y0, y1, 0);
#pragma unroll
      for (int t = gq * GS; t < (gq + 1) * GS; ++t) b0[t] = rd(x0[t], x1[t], 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (s < nsteps) {                                          // odd step count: the last step sits in set 0
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0[t], acc[t], 0, 0, 0);
  }
}

// One launch computes the weight gradients of MANY convolutions (table of WgradArgs, blockIdx.z = table entry): the
// backward plan defers every weight gradient to its end (operands stay alive), so a level's seven equal-shaped
// convolutions fill the chip together and each needs only a few M-split workgroups.  That matters because the M-split
// workgroups of a tile meet in global float atomics (~1.3 TB/s chip-wide): total atomic bytes = workgroups x accumulator
// bytes per workgroup -- at ~256 workgroups per convolution the atomics took 3-5x the matrix time.
//
// Staging: the slot -> (row, column, channel vector) decomposition of a thread's 16-byte vectors does not depend on the
// chunk, so it is done once (LDS offset, row, in-row byte offset per slot); per chunk and slot only the image / row of the
// extended row list changes (~10 instructions per vector; recomputing the decomposition per vector made a 64 KB chunk cost
// 6 us of address arithmetic at one wave per SIMD).  Loads of a round are issued together, addresses clamped not predicated.
//
// Matrix loop (bf16): the LDS byte address of a fragment read is (lane part) + (16-pixel step) x (row bytes): the XOR swizzle
// looks at position bits 0..1 only and a step moves the position by 16, so the lane parts -- 2 for the dY fragment, 2 per
// tap for X -- are computed once; per step one scalar is added.  The next step's twenty transposing reads are issued
// before the current step's nine MFMAs (a read -> wait -> MFMA chain per tap with ~10 address instructions in front of it
// ran the matrix pipe at 7 %).  NT = taps per entry (9: stride-1 3x3; 1: 1x1 and one tap of the stride-2 kind) is a
// template parameter: a run-time tap count put a branch between every two MFMAs.
// Operand pointers come out of the table (memory), i.e. as generic pointers: they are cast to the global address space --
// flat loads count on lgkmcnt too, so every LDS wait in the matrix loop would also wait for the prefetched chunk.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;   // (a native vector: address-space-qualified loads, no struct copies)
typedef const __attribute__((address_space(1))) u32x4* gptr16;
template <bool F32, int NT>
__global__ __launch_bounds__(256) void k_wgrad(const WgradArgs* __restrict__ tab) {
  constexpr int ESZ = F32 ? 4 : 2, EPV = 16 / ESZ, TB = F32 ? 128 : 64;    // TB: bytes of one 32-channel tile of a row
  constexpr int SY = 8, SX = 10;                                           // staging slots per thread (host sizes chunks to fit)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const WgradArgs a = tab[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwn = a.nwn, nwc = 4 / nwn;
  const int ngr_n = (a.N + 32 * nwn - 1) / (32 * nwn), ngr_c = (a.C + 32 * nwc - 1) / (32 * nwc);
  if ((int)blockIdx.x >= a.grid_x || (int)blockIdx.y >= ngr_n * ngr_c) return;      // (the launch grid is the table's maximum)
  const int grp_n = blockIdx.y % ngr_n, grp_c = blockIdx.y / ngr_n;
  const int nt = wave % nwn, ct = wave / nwn;
  const int n_base = grp_n * 32 * nwn, c_base = grp_c * 32 * nwc;         // first channel of the staged operand rows
  const int n0 = n_base + 32 * nt, c0 = c_base + 32 * ct;
  const bool active = n0 < a.N && c0 < a.C;
  constexpr bool three = NT == 9;                                          // (the host sorts entries by kind into two tables)
  const int Wp = a.W + 2;
  const int KP = three ? ((a.nlr * Wp + 15) & ~15) : a.nlr;                // contraction positions per chunk
  const int XP = three ? KP + 2 * Wp + 2 : KP;
  unsigned char* Ys = smem;                                                // [KP][32 nwn]
  unsigned char* Xs = smem + (size_t)KP * nwn * TB;                        // [XP][32 nwc]

  for (int i = tid * 16; i < (KP * nwn + XP * nwc) * TB; i += 256 * 16) *(uint4*)(smem + i) = make_uint4(0, 0, 0, 0);

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  const int vn = 32 * nwn / EPV, vc = 32 * nwc / EPV;                      // 16-byte vectors per staged row
  const int HW = a.H * a.W;
  const float inv_H1 = 1.0f / (float)(a.H + 1);
  const unsigned rowY = (unsigned)a.ldy * ESZ, rowX = (unsigned)a.C * ESZ; // bytes per pixel
  // Operands through buffer descriptors (32-bit byte offsets; the host keeps both tensors below 2 GiB): an out-of-range
  // offset reads zeros, which is how rows beyond the batch, the zero rows of the extended row list, a stride-2 tap outside
  // the (padded) input and unused staging slots become zero vectors -- no select per vector at the LDS write.
  constexpr unsigned OOB = 0x80000000u;
  const unsigned ybytes = (unsigned)a.B * (unsigned)HW * rowY, xbytes = (unsigned)a.B * (unsigned)(a.Hin * a.Win) * rowX;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, xbytes, 0x00020000);
  // ---- slot tables (chunk-invariant): LDS byte offset (unused slots write a 16-byte pad behind the tiles), row of the
  // extended list, byte offset inside the image row (OOB = unused slot)
  const int padY = (KP * nwn + XP * nwc) * TB, padX = padY - KP * nwn * TB;
  int yl[SY], yr[SY], xl[SX], xr[SX];
  unsigned yc[SY], xc_[SX];
  const int nY = three ? a.nlr * a.W * vn : KP * vn, nX = three ? (a.nlr + 2) * a.W * vc : KP * vc;
#pragma unroll
  for (int u = 0; u < SY; ++u) {
    const int v = tid + 256 * u;
    const int cv = v % vn, pi = v / vn, lr = three ? pi / a.W : 0, xx = three ? pi - lr * a.W : pi;
    const int ch = n_base + cv * EPV;
    const int pos = three ? lr * Wp + xx + 1 : pi;
    const bool used = v < nY && ch < a.ldy;
    yl[u] = used ? lds_off<F32>(pos, cv * EPV / 32, (cv * EPV % 32) * ESZ, nwn) : padY;
    yr[u] = lr;
    yc[u] = used ? (unsigned)xx * rowY + (unsigned)ch * ESZ : OOB;
  }
#pragma unroll
  for (int u = 0; u < SX; ++u) {
    const int v = tid + 256 * u;
    const int cv = v % vc, pi = v / vc, lr = three ? pi / a.W : 0, xx = three ? pi - lr * a.W : pi;
    const int ch = c_base + cv * EPV;
    const int pos = three ? lr * Wp + xx + 2 : pi;
    const bool used = v < nX && ch < a.C;
    xl[u] = used ? lds_off<F32>(pos, cv * EPV / 32, (cv * EPV % 32) * ESZ, nwc) : padX;
    xr[u] = lr;
    xc_[u] = used ? (unsigned)xx * rowX + (unsigned)ch * ESZ : OOB;
  }
  const int64_t M = (int64_t)a.B * HW;
  const int dyy = a.tap / 3, dxx = a.tap % 3;                              // stride-2 kind: this entry's tap
  // lane parts of the fragment read addresses (LDS byte addresses; bf16 path)
  unsigned la[2] = {0, 0}, lb[NT][2];
  {
    const int grp = lane >> 4, i16 = lane & 15, h = grp >> 1, cb = 16 * (grp & 1), q = i16 >> 2, pp = i16 & 3;
    const unsigned ybase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)Ys;
    const unsigned xbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)Xs;
    la[0] = ybase + lds_off<false>(8 * h + q, nt, (cb + 4 * pp) * 2, nwn);
    la[1] = ybase + lds_off<false>(8 * h + q + 4, nt, (cb + 4 * pp) * 2, nwn);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int off = three ? (t / 3) * Wp + (t % 3) : 0;
      lb[t][0] = xbase + lds_off<false>(off + 8 * h + q, ct, (cb + 4 * pp) * 2, nwc);
      lb[t][1] = xbase + lds_off<false>(off + 8 * h + q + 4, ct, (cb + 4 * pp) * 2, nwc);
    }
  }

  // Software pipeline (one workgroup per CU, one wave per SIMD): the next chunk's vectors are loaded into registers before
  // the matrix loop of the current one and written to LDS after it, so the global latency hides behind the MFMAs.
  u32x4 vy[SY], vx[SX];
  // Bias gradient (a.gb): the column sums of dY ride on the staging of the workgroups of channel group 0 -- every dY vector
  // passes through their registers exactly once, and a thread's vectors are always the same EPV columns (256 % vn == 0).
  // (It was a second table entry against an all-ones input: a re-read of dY with its own M-split atomics per convolution.)
  const bool do_bias = a.gb != nullptr && grp_c == 0;
  float bsum[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) bsum[j] = 0.0f;
  const bool plain = !three && a.kind == WG_1x1;
  auto load_chunk = [&](int chunk) {
    if (three) {
      // extended rows E0 .. E0 + nlr - 1 of the list (image b, row y) = (E / (H+1), E % (H+1)); y == H is the zero row
      const int E0 = chunk * a.nlr;
#pragma unroll
      for (int u = 0; u < SY; ++u) {
        const int E = E0 + yr[u];
        const int b = (int)(((float)E + 0.5f) * inv_H1), y = E - b * (a.H + 1);
        const bool ok = y < a.H && b < a.B;
        vy[u] = __builtin_amdgcn_raw_buffer_load_b128(yrs, ok ? (unsigned)(b * HW + y * a.W) * rowY + yc[u] : OOB, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < SX; ++u) {
        const int E = E0 - 1 + xr[u];
        const int b = E < 0 ? 0 : (int)(((float)E + 0.5f) * inv_H1), y = E < 0 ? a.H : E - b * (a.H + 1);
        const bool ok = y < a.H && b < a.B;
        vx[u] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? (unsigned)(b * HW + y * a.W) * rowX + xc_[u] : OOB, 0, 0);
      }
    } else if (plain) {
      // rows P0 .. P0 + KP - 1 of both operands: the descriptors are moved to the chunk (scalar work), the per-slot offsets
      // are the chunk-invariant ones of the tables and rows beyond the last fall out of range by themselves
      const unsigned P0 = (unsigned)chunk * (unsigned)KP;
      const unsigned oy = P0 * rowY, ox = P0 * rowX;
      const __amdgpu_buffer_rsrc_t yr_ = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)a.dy + oy), 0, ybytes - oy, 0x00020000);
      const __amdgpu_buffer_rsrc_t xr_ = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)a.x + ox), 0, xbytes - ox, 0x00020000);
#pragma unroll
      for (int u = 0; u < SY; ++u) vy[u] = __builtin_amdgcn_raw_buffer_load_b128(yr_, yc[u], 0, 0);
#pragma unroll
      for (int u = 0; u < SX; ++u) vx[u] = __builtin_amdgcn_raw_buffer_load_b128(xr_, xc_[u], 0, 0);
    } else {
      // output pixels P0 .. P0 + KP - 1 in flattened (b, oy, ox) order; X gathered at the tap's input pixel
      const int64_t P0 = (int64_t)chunk * KP;
#pragma unroll
      for (int u = 0; u < SY; ++u) {
        const int pi = (tid + 256 * u) / vn;
        const bool ok = P0 + pi < M;
        vy[u] = __builtin_amdgcn_raw_buffer_load_b128(yrs, ok && yc[u] != OOB ? (unsigned)P0 * rowY + yc[u] : OOB, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < SX; ++u) {
        const int pi = (tid + 256 * u) / vc;
        const int64_t p = P0 + pi;
        unsigned off = OOB;
        if (p < M && xc_[u] != OOB) {
          if (a.kind == WG_3x3_S2) {
            const int b = (int)(p / HW), r = (int)(p - (int64_t)b * HW), oy = r / a.W, ox = r - oy * a.W;
            const int yy = 2 * oy + dyy, xx = 2 * ox + dxx;
            if (yy < a.Hin && xx < a.Win) off = (unsigned)((b * a.Hin + yy) * a.Win + xx) * rowX + (xc_[u] - (unsigned)pi * rowX);
          } else {
            off = (unsigned)p * rowX + (xc_[u] - (unsigned)pi * rowX);
          }
        }
        vx[u] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
      }
    }
  };
  if ((int)blockIdx.x < a.nchunks) load_chunk(blockIdx.x);
  for (int chunk = blockIdx.x; chunk < a.nchunks; chunk += a.grid_x) {
    __syncthreads();                                                       // the previous chunk's fragments are read
#pragma unroll
    for (int u = 0; u < SY; ++u) *(u32x4*)(Ys + yl[u]) = vy[u];
#pragma unroll
    for (int u = 0; u < SX; ++u) *(u32x4*)(Xs + xl[u]) = vx[u];
    __syncthreads();
    if (do_bias) {                                                         // (absent vectors are zeros)
#pragma unroll
      for (int u = 0; u < SY; ++u) {
        if constexpr (F32) {
#pragma unroll
          for (int j = 0; j < 4; ++j) bsum[j] += __uint_as_float(vy[u][j]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) { bsum[2 * j] += bf_lo(vy[u][j]); bsum[2 * j + 1] += bf_hi(vy[u][j]); }
        }
      }
    }
    if (chunk + a.grid_x < a.nchunks) load_chunk(chunk + a.grid_x);
    if (!active) continue;
    if constexpr (F32) {
      const int li = lane & 31, kk = lane >> 5;
      for (int p0 = 0; p0 < KP; p0 += 2) {
        const float av = *(const float*)(Ys + lds_off<true>(p0 + kk, nt, li * 4, nwn));
        float bv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int off = three ? (t / 3) * Wp + (t % 3) : 0;
          bv[t] = *(const float*)(Xs + lds_off<true>(p0 + kk + off, ct, li * 4, nwc));
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
      }
    } else {
      const int nsteps = KP >> 4;
      if (nwn == 2) wgrad_steps<NT, 2>(la, lb, nsteps, acc);
      else if (nwn == 4) wgrad_steps<NT, 4>(la, lb, nsteps, acc);
      else wgrad_steps<NT, 1>(la, lb, nsteps, acc);
    }
  }
  if (do_bias) {                                                           // (block-uniform)
    __syncthreads();
    float* bs = (float*)smem;                                              // [256][EPV]
#pragma unroll
    for (int j = 0; j < EPV; ++j) bs[tid * EPV + j] = bsum[j];
    __syncthreads();
    if (tid < 32 * nwn) {                                                  // one thread per staged dY column
      const int cvq = tid / EPV, j = tid % EPV;
      float sacc = 0.0f;
      for (int k = cvq; k < 256; k += vn) sacc += bs[k * EPV + j];
      const int n = n_base + tid;
      if (n < a.N) __hip_atomic_fetch_add(a.gb + n, sacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!active) return;
  // ---- flush: lane column c = c0 + (lane & 31), register r row n = n0 + (r&3) + 8 (r>>2) + 4 (lane>>5): one register of
  // the wave is two 128-byte row segments, the float-atomic unit that runs at the full rate
  const int col = c0 + (lane & 31), g = lane >> 5;
  if (col < a.C) {
    __attribute__((address_space(1))) float* gw = (__attribute__((address_space(1))) float*)a.gw;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int tap = a.kind == WG_3x3_S2 ? a.tap : t;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (n < a.N) __hip_atomic_fetch_add(gw + (size_t)n * a.Ktot + a.koff + tap * a.C + col, acc[t][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// ============================================================================ GroupNorm (+Swish, +dropout) backward
// forward (unet.py:103-133): z = gamma xhat + beta, xhat = (x - mean_g) rstd_g ; a = swish(z) [* keep / (1 - p)]
// backward (k_gn_bwd_sums, k_gn_bwd_dx): dz = dA swish'(z) [* keep / (1 - p)]
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
  float* dsum_bn; int dsum_stride; float* dsum_n;            // optional (single source): sum_p dX per (sample, channel) -> [b*stride + c], and += over samples
};
// keep-mask of the 8 channels starting at element index e0 (multiple of 8): two Philox blocks
__device__ inline unsigned drop_keep8(uint64_t seed, uint64_t offset, uint64_t e0, float p) {
  const u4 r0 = philox_row(seed, offset, e0 >> 2, 0x44524F50u), r1 = philox_row(seed, offset, (e0 >> 2) + 1, 0x44524F50u);
  const uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
  unsigned m = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) m |= (u01(w[j]) >= p ? 1u : 0u) << j;
  return m;
}
// raw operands of one 8-channel vector: the forward input x and the incoming gradient dA (two 16-byte loads)
struct GnVecRaw { float x[8], da[8]; size_t oo; };
__device__ inline void gn_bwd_load(const GnBwdArgs& a, int b, int px, int c0, GnVecRaw& r) {
  const int C = a.C1 + a.C2;
  const bool first = c0 < a.C1;
  const int cc = first ? c0 : c0 - a.C1, Cs = first ? a.C1 : a.C2;
  load8(first ? a.s1_f32 : a.s2_f32, first ? a.s1_bf16 : a.s2_bf16, ((size_t)b * a.HW + px) * Cs + cc, r.x);
  r.oo = ((size_t)b * a.HW + px) * C + c0;
  load8(a.da_f32, a.da_bf16, r.oo, r.da);
}
// dz and xhat from them; tm / tr / tg / tb = mean, rstd, gamma, beta of the thread's eight channels (registers: a thread keeps
// one channel vector for all its pixels; reading them from LDS per element was most of this kernel's time)
__device__ inline void gn_bwd_math(const GnBwdArgs& a, const float (&tm)[8], const float (&tr)[8], const float (&tg)[8], const float (&tb)[8],
                                   const GnVecRaw& r, float (&dz)[8], float (&xh)[8]) {
  unsigned keep = 0xFFu;
  float inv_keep = 1.0f;
  if (a.drop_p > 0.0f) { keep = drop_keep8(a.rng[0], a.rng[1] * 4096u + a.layer, r.oo, a.drop_p); inv_keep = 1.0f / (1.0f - a.drop_p); }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    xh[j] = (r.x[j] - tm[j]) * tr[j];
    float d = r.da[j];
    if (a.drop_p > 0.0f) d = (keep >> j) & 1u ? d * inv_keep : 0.0f;
    if (a.swish) {
      const float z = fmaf(tg[j], xh[j], tb[j]);
      const float sg = a.da_bf16 ? __builtin_amdgcn_rcpf(1.0f + __expf(-z)) : 1.0f / (1.0f + expf(-z));
      d *= sg * (1.0f + z * (1.0f - sg));
    }
    dz[j] = d;
  }
}
// Two launches, both with a workgroup per (sample, slice of pixels) and ALL channels: a wave's 16-byte loads then walk whole
// NHWC pixel rows (a workgroup per channel block -- which would let one launch do everything -- reads 16 bytes of every
// C*2-byte row per lane: uncoalesced, 0.6-1.0 TB/s measured).  thread = (pixel lane, channel vector cv = tid % (C/8)): its
// eight channels are fixed, so mean / rstd / gamma / beta (and the group means in pass 2) sit in registers; four pixels'
// loads are in flight per thread; every channel's statistics are fetched in one round trip and summed per group from LDS.
//   pass 1  k_gn_bwd_sums: per-(sample, channel) sums of dz and dz*xhat: block reduction in LDS, one atomic per channel,
//           moment and workgroup into `sums` (zeroed by the caller)
//   pass 2  k_gn_bwd_dx:   group means from `sums`, dX (+= existing gradient); block x = 0 also writes the closed-form
//           per-sample sums of dX (time-projection / conv1-bias gradients)
struct GnRegs { float tm[8], tr[8], tg[8], tb[8]; };
__device__ inline void gn_bwd_setup(const GnBwdArgs& a, int b, float* tab, double* cst, int cv, GnRegs& R) {
  const int C = a.C1 + a.C2, cg = C / a.G;
  for (int c = threadIdx.x; c < C; c += 256) {
    const double* st = c < a.C1 ? a.st1 + ((size_t)b * a.C1 + c) * 2 : a.st2 + ((size_t)b * a.C2 + (c - a.C1)) * 2;
    cst[2 * c] = st[0]; cst[2 * c + 1] = st[1];
    tab[2 * C + c] = a.gamma[c];
    tab[3 * C + c] = a.beta[c];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int gl = (c / cg) * cg;
    double s = 0.0, q = 0.0;
    for (int jc = gl; jc < gl + cg; ++jc) { s += cst[2 * jc]; q += cst[2 * jc + 1]; }
    const double n = (double)cg * (double)a.HW, m = s / n, var = fmax(q / n - m * m, 0.0);
    tab[c] = (float)m;
    tab[C + c] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    R.tm[j] = tab[8 * cv + j]; R.tr[j] = tab[C + 8 * cv + j]; R.tg[j] = tab[2 * C + 8 * cv + j]; R.tb[j] = tab[3 * C + 8 * cv + j];
  }
}
__global__ __launch_bounds__(256) void k_gn_bwd_sums(const GnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = a.C1 + a.C2, b = blockIdx.y, vpp = C / 8;
  float* tab = sm;                                   // [4][C]
  double* cst = (double*)(sm + 4 * C);               // [C][2]
  float* red = sm + 8 * C;                           // [256][16]
  const int lanes = 256 / vpp, cv = threadIdx.x % vpp, pl = threadIdx.x / vpp;          // C <= 2048
  const bool worker = pl < lanes;
  const int per = (a.HW + gridDim.x - 1) / gridDim.x, p_lo = blockIdx.x * per, p_hi = min(p_lo + per, a.HW);
  GnVecRaw r[4];
  if (worker && p_lo < p_hi) {
#pragma unroll
    for (int u = 0; u < 4; ++u) gn_bwd_load(a, b, min(p_lo + pl + u * lanes, p_hi - 1), cv * 8, r[u]);
  }
  GnRegs R;
  gn_bwd_setup(a, b, tab, cst, cv, R);
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.0f; s2[j] = 0.0f; }
  if (worker) {
    for (int px0 = p_lo + pl; px0 < p_hi; px0 += 4 * lanes) {
      if (px0 != p_lo + pl) {
#pragma unroll
        for (int u = 0; u < 4; ++u) gn_bwd_load(a, b, min(px0 + u * lanes, p_hi - 1), cv * 8, r[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (px0 + u * lanes < p_hi) {
          float dz[8], xh[8];
          gn_bwd_math(a, R.tm, R.tr, R.tg, R.tb, r[u], dz, xh);
#pragma unroll
          for (int j = 0; j < 8; ++j) { s1[j] += dz[j]; s2[j] = fmaf(dz[j], xh[j], s2[j]); }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[threadIdx.x * 16 + j] = s1[j]; red[threadIdx.x * 16 + 8 + j] = s2[j]; }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * C; t += 256) {                     // output (which, channel): sum over the pixel lanes
    const int which = t / C, c = t % C, vv = c / 8, j = c % 8;
    float acc = 0.0f;
    for (int p = 0; p < lanes; ++p) acc += red[(p * vpp + vv) * 16 + which * 8 + j];
    atomicAdd(a.sums + ((size_t)b * C + c) * 2 + which, acc);
  }
}
__global__ __launch_bounds__(256) void k_gn_bwd_dx(const GnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = a.C1 + a.C2, b = blockIdx.y, vpp = C / 8, cg = C / a.G;
  float* tab = sm;                                   // [4][C]
  double* cst = (double*)(sm + 4 * C);               // [C][2]
  float* ss = sm + 8 * C;                            // [C][2] this sample's sums, then m1[C], m2[C]
  const int lanes = 256 / vpp, cv = threadIdx.x % vpp, pl = threadIdx.x / vpp;
  const bool worker = pl < lanes;
  const int per = (a.HW + gridDim.x - 1) / gridDim.x, p_lo = blockIdx.x * per, p_hi = min(p_lo + per, a.HW);
  const int c0 = cv * 8;
  const bool first = c0 < a.C1;
  const int cc = first ? c0 : c0 - a.C1, Cs = first ? a.C1 : a.C2;
  float* const df = first ? a.d1_f32 : a.d2_f32;
  unsigned short* const dh = first ? a.d1_bf16 : a.d2_bf16;
  const bool accum = first ? a.acc1 : a.acc2;
  GnVecRaw r[4];
  float old[4][8];
  if (worker && p_lo < p_hi) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int px = min(p_lo + pl + u * lanes, p_hi - 1);
      gn_bwd_load(a, b, px, c0, r[u]);
      if (accum) load8(df, dh, ((size_t)b * a.HW + px) * Cs + cc, old[u]);
    }
  }
  for (int i = threadIdx.x; i < 2 * C; i += 256) ss[i] = a.sums[(size_t)b * C * 2 + i];
  GnRegs R;
  gn_bwd_setup(a, b, tab, cst, cv, R);               // (its barriers also publish ss)
  float* m1 = ss + 2 * C;
  float* m2 = m1 + C;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int gl = (c / cg) * cg;
    float t1 = 0.0f, t2 = 0.0f;
    for (int jc = gl; jc < gl + cg; ++jc) { t1 = fmaf(tab[2 * C + jc], ss[2 * jc], t1); t2 = fmaf(tab[2 * C + jc], ss[2 * jc + 1], t2); }
    const float inv = 1.0f / ((float)cg * (float)a.HW);
    m1[c] = t1 * inv; m2[c] = t2 * inv;
  }
  __syncthreads();
  if (blockIdx.x == 0 && (a.dsum_bn || a.dsum_n)) {
    // sum over the sample's pixels of dX, in closed form from the sums (the input is conv1's output + bias + time projection:
    // this IS the gradient of the time projection, unet.py:110,131, and its sum over samples the bias gradient):
    //   sum_p dX = rstd (gamma S1 - HW m1 - m2 sum_p xhat),   sum_p xhat = (sum_p x - HW mean) rstd
    for (int c = threadIdx.x; c < a.C1; c += 256) {
      const float sxh = ((float)cst[2 * c] - (float)a.HW * tab[c]) * tab[C + c];
      const float val = tab[C + c] * (tab[2 * C + c] * ss[2 * c] - (float)a.HW * m1[c] - m2[c] * sxh);
      if (a.dsum_bn) a.dsum_bn[(size_t)b * a.dsum_stride + c] = val;
      if (a.dsum_n) atomicAdd(a.dsum_n + c, val);
    }
  }
  if (!worker) return;
  float sc[8], k1[8], k2[8];                                       // dX = sc dz - k1 - xhat k2
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = R.tr[j] * R.tg[j]; k1[j] = R.tr[j] * m1[c0 + j]; k2[j] = R.tr[j] * m2[c0 + j]; }
  for (int px0 = p_lo + pl; px0 < p_hi; px0 += 4 * lanes) {
    if (px0 != p_lo + pl) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int px = min(px0 + u * lanes, p_hi - 1);
        gn_bwd_load(a, b, px, c0, r[u]);
        if (accum) load8(df, dh, ((size_t)b * a.HW + px) * Cs + cc, old[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int px = px0 + u * lanes;
      if (px >= p_hi) continue;
      float dz[8], xh[8], dx[8];
      gn_bwd_math(a, R.tm, R.tr, R.tg, R.tb, r[u], dz, xh);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        dx[j] = fmaf(-xh[j], k2[j], fmaf(sc[j], dz[j], -k1[j]));
        if (accum) dx[j] += old[u][j];
      }
      store8(df, dh, ((size_t)b * a.HW + px) * Cs + cc, dx);
    }
  }
}

// The same backward in ONE launch for bf16 tensors (round 3): a workgroup owns (sample, slab of whole groups) -- everything the
// two group means need -- and keeps the slab's x and dz in registers between the sums and dX.  The slab is 24-96 channels
// wide (48-192 contiguous bytes per pixel and lane group, not the 16 bytes per row the note above measured), and the slabs
// of one sample are placed on one XCD (blockIdx.x = sample, a multiple-of-8 batch), so its rows cross the fabric once.
// Thread t = (pixel lane pl = t / noct, octet oct = t % noct); pixels pl, pl + npl, ...  `sums` are plain stores here.
template <int MAXV>
__global__ __launch_bounds__(512) void k_gn_bwd_onepass(const GnBwdArgs a, int slabC, int noct, int npl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  const int C = a.C1 + a.C2, cg = C / a.G, b = blockIdx.x, cs0 = blockIdx.y * slabC, HW = a.HW;
  const int t = threadIdx.x, T = noct * npl;
  constexpr int PST = 20;                                      // floats per thread record (16 used): 80-byte records spread the 16-byte
                                                               // stores of neighbouring threads over all banks (as k_gn_onepass)
  float* part = (float*)gsm;                                   // [T][PST]
  float* tab = part + (size_t)T * PST;                         // mean, rstd, gamma, beta, S1, S2, m1, m2: [8][slabC]
  const bool act = t < T;
  const int oct = act ? t % noct : 0, pl = act ? t / noct : 0;
  const int c0 = cs0 + oct * 8;
  const bool first = c0 < a.C1;
  const int cc = first ? c0 : c0 - a.C1, Cs = first ? a.C1 : a.C2;
  const unsigned short* xs = (first ? a.s1_bf16 : a.s2_bf16) + (size_t)b * HW * Cs + cc;
  const unsigned short* das = a.da_bf16 + (size_t)b * HW * C + c0;
  unsigned short* const dh = (first ? a.d1_bf16 : a.d2_bf16) + (size_t)b * HW * Cs + cc;
  const bool accum = first ? a.acc1 : a.acc2;
  uint4 ux[MAXV], ud[MAXV];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int p = pl + k * npl;
    const bool ok = act && p < HW;
    ux[k] = ok ? *(const uint4*)(xs + (size_t)p * Cs) : make_uint4(0, 0, 0, 0);
    ud[k] = ok ? *(const uint4*)(das + (size_t)p * C) : make_uint4(0, 0, 0, 0);
  }
  for (int cl = t; cl < slabC; cl += blockDim.x) {             // statistics of the forward (fp64 sums of the producers' epilogues)
    const int c = cs0 + cl, gl = (c / cg) * cg;
    double s = 0.0, q = 0.0;
    for (int jc = gl; jc < gl + cg; ++jc) {
      const double* st = jc < a.C1 ? a.st1 + ((size_t)b * a.C1 + jc) * 2 : a.st2 + ((size_t)b * a.C2 + (jc - a.C1)) * 2;
      s += st[0]; q += st[1];
    }
    const double n = (double)cg * (double)HW, m = s / n, var = fmax(q / n - m * m, 0.0);
    tab[cl] = (float)m;
    tab[slabC + cl] = (float)(1.0 / sqrt(var + (double)a.eps));
    tab[2 * slabC + cl] = a.gamma[c];
    tab[3 * slabC + cl] = a.beta[c];
  }
  __syncthreads();
  float tm[8], tr[8], tg[8], tb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    tm[j] = tab[oct * 8 + j]; tr[j] = tab[slabC + oct * 8 + j]; tg[j] = tab[2 * slabC + oct * 8 + j]; tb[j] = tab[3 * slabC + oct * 8 + j];
  }
  float dzv[MAXV][8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.0f; s2[j] = 0.0f; }
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int p = pl + k * npl;
    if (act && p < HW) {
      GnVecRaw r;
      const unsigned wx[4] = {ux[k].x, ux[k].y, ux[k].z, ux[k].w}, wd[4] = {ud[k].x, ud[k].y, ud[k].z, ud[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r.x[2 * j] = bf_lo(wx[j]); r.x[2 * j + 1] = bf_hi(wx[j]);
        r.da[2 * j] = bf_lo(wd[j]); r.da[2 * j + 1] = bf_hi(wd[j]);
      }
      r.oo = ((size_t)b * HW + p) * C + c0;
      float xh[8];
      gn_bwd_math(a, tm, tr, tg, tb, r, dzv[k], xh);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += dzv[k][j]; s2[j] = fmaf(dzv[k][j], xh[j], s2[j]); }
      if (accum) ud[k] = *(const uint4*)(dh + (size_t)p * Cs);               // the gradient already there: in flight under the reductions
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) dzv[k][j] = 0.0f;
    }
  }
  if (act) {
    float4* pt = (float4*)(part + (size_t)t * PST);
    pt[0] = make_float4(s1[0], s1[1], s1[2], s1[3]); pt[1] = make_float4(s1[4], s1[5], s1[6], s1[7]);
    pt[2] = make_float4(s2[0], s2[1], s2[2], s2[3]); pt[3] = make_float4(s2[4], s2[5], s2[6], s2[7]);
  }
  __syncthreads();
  for (int r = t; r < 2 * slabC; r += blockDim.x) {
    const int which = r / slabC, cl = r - which * slabC;
    const float* pp = part + (size_t)(cl >> 3) * PST + which * 8 + (cl & 7);
    float acc = 0.0f;
    for (int q = 0; q < npl; ++q) acc += pp[(size_t)q * noct * PST];
    tab[(4 + which) * slabC + cl] = acc;
    a.sums[((size_t)b * C + cs0 + cl) * 2 + which] = acc;      // (sole owner of these entries: no atomics)
  }
  __syncthreads();
  for (int cl = t; cl < slabC; cl += blockDim.x) {
    const int c = cs0 + cl, gl = (c / cg) * cg - cs0;
    float t1 = 0.0f, t2 = 0.0f;
    for (int jc = gl; jc < gl + cg; ++jc) { t1 = fmaf(tab[2 * slabC + jc], tab[4 * slabC + jc], t1); t2 = fmaf(tab[2 * slabC + jc], tab[5 * slabC + jc], t2); }
    const float inv = 1.0f / ((float)cg * (float)HW);
    tab[6 * slabC + cl] = t1 * inv; tab[7 * slabC + cl] = t2 * inv;
  }
  __syncthreads();
  if (a.dsum_bn || a.dsum_n) {                                  // closed-form per-sample sums of dX (as k_gn_bwd_dx)
    for (int cl = t; cl < slabC; cl += blockDim.x) {
      const int c = cs0 + cl;
      if (c >= a.C1) continue;
      const float sx = (float)a.st1[((size_t)b * a.C1 + c) * 2];
      const float sxh = (sx - (float)HW * tab[cl]) * tab[slabC + cl];
      const float val = tab[slabC + cl] * (tab[2 * slabC + cl] * tab[4 * slabC + cl] - (float)HW * tab[6 * slabC + cl] - tab[7 * slabC + cl] * sxh);
      if (a.dsum_bn) a.dsum_bn[(size_t)b * a.dsum_stride + c] = val;
      if (a.dsum_n) atomicAdd(a.dsum_n + c, val);
    }
  }
  if (!act) return;
  float sc[8], k1[8], k2[8];                                       // dX = sc dz - k1 - xhat k2
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = tr[j] * tg[j]; k1[j] = tr[j] * tab[6 * slabC + oct * 8 + j]; k2[j] = tr[j] * tab[7 * slabC + oct * 8 + j]; }
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int p = pl + k * npl;
    if (p >= HW) break;
    const unsigned wx[4] = {ux[k].x, ux[k].y, ux[k].z, ux[k].w}, wo[4] = {ud[k].x, ud[k].y, ud[k].z, ud[k].w};
    float dx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (j & 1) ? bf_hi(wx[j >> 1]) : bf_lo(wx[j >> 1]);
      const float xh = (x - tm[j]) * tr[j];
      dx[j] = fmaf(-xh, k2[j], fmaf(sc[j], dzv[k][j], -k1[j]));
      if (accum) dx[j] += (j & 1) ? bf_hi(wo[j >> 1]) : bf_lo(wo[j >> 1]);
    }
    *(uint4*)(dh + (size_t)p * Cs) = make_uint4(pack2_bf16(dx[0], dx[1]), pack2_bf16(dx[2], dx[3]), pack2_bf16(dx[4], dx[5]), pack2_bf16(dx[6], dx[7]));
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

// out[j] (+)= sum_b in[b * bstride + j * jstride]  (GroupNorm dgamma / dbeta from the per-sample sums, ...): workgroup = 64
// outputs x 4 batch lanes
__global__ __launch_bounds__(256) void k_sum_batch(const float* in, int B, int64_t bstride, int jstride, int n, float* out, int accumulate) {
  __shared__ float part[4][64];
  const int jl = threadIdx.x & 63, bl = threadIdx.x >> 6, j = blockIdx.x * 64 + jl;
  float s = 0.0f;
  if (j < n)
    for (int b = bl; b < B; b += 4) s += in[(size_t)b * bstride + (size_t)j * jstride];
  part[bl][jl] = s;
  __syncthreads();
  if (bl == 0 && j < n) {
    const float t = (part[0][jl] + part[1][jl]) + (part[2][jl] + part[3][jl]);
    out[j] = accumulate ? out[j] + t : t;
  }
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

// The same reduction for a TABLE of jobs in one launch (the backward plan's ~80 small sums: GroupNorm dgamma / dbeta from the
// per-sample sums, bias gradients out of the weight-gradient launch's ones-column, copies for parameters that share a
// gradient): workgroup = one job's 64 outputs x 4 batch lanes (a thread per output walking B samples serially was 10 us of
// dependent-load latency per launch, ~1 ms per training step).
struct SumJob { const float* in; float* out; int64_t bstride; int B, jstride, n, accumulate, j0, pad_; };
__global__ __launch_bounds__(256) void k_sum_jobs(const SumJob* __restrict__ jobs) {
  __shared__ float part[4][64];
  const SumJob q = jobs[blockIdx.x];
  const int jl = threadIdx.x & 63, bl = threadIdx.x >> 6, j = q.j0 + jl;
  float s = 0.0f;
  if (j < q.n)
    for (int b = bl; b < q.B; b += 4) s += q.in[(size_t)b * q.bstride + (size_t)j * q.jstride];
  part[bl][jl] = s;
  __syncthreads();
  if (bl == 0 && j < q.n) {
    const float t = (part[0][jl] + part[1][jl]) + (part[2][jl] + part[3][jl]);
    q.out[j] = q.accumulate ? q.out[j] + t : t;
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

// The same on 4 x 4 register tiles (as k_attn_small_t4 of the forward): q, k, v, dO TRANSPOSED in LDS for the two token x token
// products (scores and dW), row-major for the three token x channel products, dS kept in both orientations -- one 16-byte read per
// operand and 16 (32, 48) multiply-adds; the thread-per-score kernel above read k and v rows at a stride of ch floats across
// neighbouring lanes (bank conflicts: 58 % of its LDS cycles) and ran the softmax rows on T of its 256 threads.
__global__ __launch_bounds__(256) void k_attn_small_bwd_t4(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x / a.heads, hd = blockIdx.x % a.heads, ch = a.C / a.heads, T = a.T, Tp = (T + 3) & ~3;
  float* qT = sm; float* kT = qT + ch * Tp; float* vT = kT + ch * Tp; float* dOT = vT + ch * Tp;      // [ch][Tp]
  float* q = dOT + ch * Tp; float* k = q + T * ch; float* dO = k + T * ch;                             // [T][ch]
  float* w = dO + T * ch; float* dS = w + T * Tp; float* dST = dS + T * Tp;                            // [T][Tp]
  const float sc = 1.0f / sqrtf(sqrtf((float)ch));
  for (int i = threadIdx.x; i < 4 * ch * Tp + 3 * T * ch + 3 * T * Tp; i += 256) sm[i] = 0.0f;        // (pad tokens: zeros everywhere)
  __syncthreads();
  for (int i = threadIdx.x; i < T * ch; i += 256) {
    const int t = i / ch, c = i % ch;
    const float* src = a.qkv + ((size_t)b * T + t) * 3 * a.C + hd * 3 * ch;
    const float qv = src[c] * sc, kv = src[ch + c] * sc, vv = src[2 * ch + c];
    const size_t oo = ((size_t)b * T + t) * a.C + hd * ch + c;
    const float dv_ = a.d_out_f32 ? a.d_out_f32[oo] : bf_lo((unsigned)a.d_out_bf16[oo]);
    q[i] = qv; k[i] = kv; dO[i] = dv_;
    qT[c * Tp + t] = qv; kT[c * Tp + t] = kv; vT[c * Tp + t] = vv; dOT[c * Tp + t] = dv_;
  }
  __syncthreads();
  const int nbt = Tp / 4;
  for (int blk = threadIdx.x; blk < nbt * nbt; blk += 256) {          // scores w[t][s] and dW[t][s] (into dS)
    const int tb = blk / nbt, sb = blk % nbt;
    float aw[4][4], ad[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { aw[i][j] = 0.0f; ad[i][j] = 0.0f; }
    for (int c = 0; c < ch; ++c) {
      const float4 qa = *(const float4*)(qT + c * Tp + 4 * tb), kb = *(const float4*)(kT + c * Tp + 4 * sb);
      const float4 da = *(const float4*)(dOT + c * Tp + 4 * tb), vb = *(const float4*)(vT + c * Tp + 4 * sb);
      const float qv[4] = {qa.x, qa.y, qa.z, qa.w}, kv[4] = {kb.x, kb.y, kb.z, kb.w}, dv_[4] = {da.x, da.y, da.z, da.w}, vv[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { aw[i][j] = fmaf(qv[i], kv[j], aw[i][j]); ad[i][j] = fmaf(dv_[i], vv[j], ad[i][j]); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (4 * tb + i < T) {
        *(float4*)(w + (4 * tb + i) * Tp + 4 * sb) = make_float4(aw[i][0], aw[i][1], aw[i][2], aw[i][3]);
        *(float4*)(dS + (4 * tb + i) * Tp + 4 * sb) = make_float4(ad[i][0], ad[i][1], ad[i][2], ad[i][3]);
      }
  }
  __syncthreads();
  for (int t4 = threadIdx.x; t4 < ((T + 63) / 64) * 256; t4 += 256) {  // softmax rows and dS = w (dW - sum_s dW w): four lanes per row
    const int t = t4 >> 2, l4 = t4 & 3;
    float m = -INFINITY;
    if (t < T)
      for (int s_ = l4; s_ < T; s_ += 4) m = fmaxf(m, w[t * Tp + s_]);
    m = fmaxf(m, __shfl_xor(m, 1, WAVE));
    m = fmaxf(m, __shfl_xor(m, 2, WAVE));
    float z = 0.0f;
    if (t < T)
      for (int s_ = l4; s_ < T; s_ += 4) { const float e = expf(w[t * Tp + s_] - m); w[t * Tp + s_] = e; z += e; }
    z += __shfl_xor(z, 1, WAVE);
    z += __shfl_xor(z, 2, WAVE);
    const float iz = 1.0f / z;
    float dot = 0.0f;
    if (t < T)
      for (int s_ = l4; s_ < T; s_ += 4) { const float p = w[t * Tp + s_] * iz; w[t * Tp + s_] = p; dot = fmaf(dS[t * Tp + s_], p, dot); }
    dot += __shfl_xor(dot, 1, WAVE);
    dot += __shfl_xor(dot, 2, WAVE);
    if (t < T)
      for (int s_ = l4; s_ < T; s_ += 4) {
        const float v_ = w[t * Tp + s_] * (dS[t * Tp + s_] - dot);
        dS[t * Tp + s_] = v_;
        dST[s_ * Tp + t] = v_;
      }
  }
  __syncthreads();
  const int nbc = ch / 4;
  for (int blk = threadIdx.x; blk < nbt * nbc; blk += 256) {          // dq, dk, dv for a block of 4 tokens x 4 channels
    const int tb = blk / nbc, cb = blk % nbc;
    float aq[4][4], ak[4][4], av[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { aq[i][j] = 0.0f; ak[i][j] = 0.0f; av[i][j] = 0.0f; }
    for (int s_ = 0; s_ < T; ++s_) {
      const float4 d1 = *(const float4*)(dST + s_ * Tp + 4 * tb);      // dS[t0..t0+3][s]
      const float4 d2 = *(const float4*)(dS + s_ * Tp + 4 * tb);       // dS[s][t0..t0+3]
      const float4 w2 = *(const float4*)(w + s_ * Tp + 4 * tb);        // w[s][t0..t0+3]
      const float4 kb = *(const float4*)(k + s_ * ch + 4 * cb), qb = *(const float4*)(q + s_ * ch + 4 * cb), ob = *(const float4*)(dO + s_ * ch + 4 * cb);
      const float x1[4] = {d1.x, d1.y, d1.z, d1.w}, x2[4] = {d2.x, d2.y, d2.z, d2.w}, x3[4] = {w2.x, w2.y, w2.z, w2.w};
      const float y1[4] = {kb.x, kb.y, kb.z, kb.w}, y2[4] = {qb.x, qb.y, qb.z, qb.w}, y3[4] = {ob.x, ob.y, ob.z, ob.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          aq[i][j] = fmaf(x1[i], y1[j], aq[i][j]); ak[i][j] = fmaf(x2[i], y2[j], ak[i][j]); av[i][j] = fmaf(x3[i], y3[j], av[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = 4 * tb + i;
      if (t >= T) break;
      const size_t d0 = ((size_t)b * T + t) * 3 * a.C + hd * 3 * ch + 4 * cb;
      if (a.d_qkv) {
        *(float4*)(a.d_qkv + d0) = make_float4(aq[i][0] * sc, aq[i][1] * sc, aq[i][2] * sc, aq[i][3] * sc);
        *(float4*)(a.d_qkv + d0 + ch) = make_float4(ak[i][0] * sc, ak[i][1] * sc, ak[i][2] * sc, ak[i][3] * sc);
        *(float4*)(a.d_qkv + d0 + 2 * ch) = make_float4(av[i][0], av[i][1], av[i][2], av[i][3]);
      }
      if (a.d_qkv_bf16) {
        *(uint2*)(a.d_qkv_bf16 + d0) = make_uint2(pack2_bf16(aq[i][0] * sc, aq[i][1] * sc), pack2_bf16(aq[i][2] * sc, aq[i][3] * sc));
        *(uint2*)(a.d_qkv_bf16 + d0 + ch) = make_uint2(pack2_bf16(ak[i][0] * sc, ak[i][1] * sc), pack2_bf16(ak[i][2] * sc, ak[i][3] * sc));
        *(uint2*)(a.d_qkv_bf16 + d0 + 2 * ch) = make_uint2(pack2_bf16(av[i][0], av[i][1]), pack2_bf16(av[i][2], av[i][3]));
      }
    }
  }
}

// ============================================================================ first conv weight gradient (unet.py:343, C_in = 1..4)
// dW0[n][ci][tap] = sum_{b,p} dY[b,p,n] xc[b,ci,p + off(tap)] with xc the centred integer state.  A workgroup takes a band of
// image rows of one sample: the centred input band (+ halo, zero border) is converted once into LDS, thread = (channel n,
// pixel lane) reads dY coalesced over n and the nine taps as LDS broadcasts; per-workgroup partial sums leave as plain
// stores ([workgroup][Cout*K | Cout]) and k_sum_batch adds the workgroups up (no atomics on the 960-word result).
struct FirstWgradArgs { const int64_t* x64; const int32_t* x32; float lo, hi; const float* dy_f32; const unsigned short* dy_bf16;
                        int B, Cin, H, W, Cout; float* gw; float* gbias; float* partial; };
__host__ __device__ inline int first_wgrad_bands(int H) { return H >= 16 ? 8 : H >= 8 ? 4 : 1; }
__global__ __launch_bounds__(256) void k_first_conv_wgrad(const FirstWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = a.Cin * 9, HW = a.H * a.W, b = blockIdx.y, Wp = a.W + 2;
  const int rows_per = (a.H + gridDim.x - 1) / gridDim.x, y_lo = blockIdx.x * rows_per, y_hi = min(y_lo + rows_per, a.H);
  float* slab = sm;                                                       // [Cin][rows_per + 2][W + 2]
  float* acc_s = sm + a.Cin * (rows_per + 2) * Wp;                        // [Cout][K + 1]
  for (int i = threadIdx.x; i < a.Cout * (K + 1); i += 256) acc_s[i] = 0.0f;
  for (int idx = threadIdx.x; idx < a.Cin * (rows_per + 2) * Wp; idx += 256) {
    const int ci = idx / ((rows_per + 2) * Wp), rem = idx % ((rows_per + 2) * Wp), ry = rem / Wp, xc = rem % Wp;
    const int yy = y_lo - 1 + ry, xx = xc - 1;
    float v = 0.0f;
    if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
      const size_t o = (((size_t)b * a.Cin + ci) * a.H + yy) * a.W + xx;
      const float raw = a.x64 ? (float)a.x64[o] : (float)a.x32[o];
      v = 2.0f * ((raw - a.lo) / (a.hi - a.lo)) - 1.0f;
    }
    slab[idx] = v;
  }
  __syncthreads();
  const int lanes = max(256 / a.Cout, 1), n = threadIdx.x % a.Cout, pl = threadIdx.x / a.Cout;
  if (pl < lanes && y_lo < y_hi) {
    float acc[37];
#pragma unroll
    for (int i = 0; i < 37; ++i) acc[i] = 0.0f;
    const int p_lo = y_lo * a.W, p_hi = y_hi * a.W;
    for (int p = p_lo + pl; p < p_hi; p += lanes) {
      const size_t o = ((size_t)b * HW + p) * a.Cout + n;
      const float g = a.dy_f32 ? a.dy_f32[o] : bf_lo((unsigned)a.dy_bf16[o]);
      const int y = p / a.W - y_lo, x = p % a.W;
      acc[36] += g;
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        if (ci < a.Cin) {
          const float* row = slab + (ci * (rows_per + 2) + y) * Wp + x;   // (y, x) of the band = slab row y + 1, column x + 1
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[ci * 9 + t] = fmaf(g, row[(t / 3) * Wp + (t % 3)], acc[ci * 9 + t]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 36; ++i)
      if (i < K) atomicAdd(acc_s + n * (K + 1) + i, acc[i]);
    atomicAdd(acc_s + n * (K + 1) + K, acc[36]);
  }
  __syncthreads();
  float* out = a.partial + ((size_t)b * gridDim.x + blockIdx.x) * a.Cout * (K + 1);
  for (int i = threadIdx.x; i < a.Cout * (K + 1); i += 256) {
    const int nn = i / (K + 1), kk = i % (K + 1);
    out[kk < K ? nn * K + kk : a.Cout * K + nn] = acc_s[i];              // [Cout*K weights (torch layout) | Cout bias sums]
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

// table: `n` ctdd_wgrad_args in DEVICE memory (table_host: the same entries in host memory, for validation and sizing).
// All entries of one call are of one tap count: CTDD_WG_3x3 entries (nine taps), or CTDD_WG_1x1 / CTDD_WG_3x3_S2 entries (one).
extern "C" int ctdd_unet_wgrad(const void* table_dev, const void* table_host, int n, int f32, void* stream) {
  CTDD_REQUIRE(table_dev && table_host && n > 0 && n <= 65535, CTDD_EINVAL, "wgrad: empty table / n=%d", n);
  const WgradArgs* t = (const WgradArgs*)table_host;
  const int epv = f32 ? 4 : 8, tb = f32 ? 128 : 64;
  size_t lds = 0;
  int gx = 1, gy = 1;
  const bool nine = t[0].kind == WG_3x3;
  for (int i = 0; i < n; ++i) {
    const WgradArgs& a = t[i];
    CTDD_REQUIRE(a.x && a.dy && a.gw, CTDD_EINVAL, "wgrad[%d]: null operand", i);
    CTDD_REQUIRE(a.nwn == 1 || a.nwn == 2 || a.nwn == 4, CTDD_EINVAL, "wgrad[%d]: nwn=%d", i, a.nwn);
    CTDD_REQUIRE(a.C % epv == 0 && a.ldy % epv == 0 && a.N <= a.ldy, CTDD_EINVAL, "wgrad[%d]: C=%d ldy=%d N=%d need %d-element vectors", i, a.C,
                 a.ldy, a.N, epv);
    CTDD_REQUIRE(a.kind == WG_3x3 || a.kind == WG_1x1 || a.kind == WG_3x3_S2, CTDD_EINVAL, "wgrad[%d]: kind=%d", i, a.kind);
    CTDD_REQUIRE((a.kind == WG_3x3) == nine, CTDD_EINVAL, "wgrad[%d]: a table holds nine-tap entries or one-tap entries, not both", i);
    CTDD_REQUIRE(a.nlr > 0 && a.nchunks > 0 && a.grid_x > 0 && a.tap >= 0 && a.tap < 9, CTDD_EINVAL, "wgrad[%d]: nlr=%d nchunks=%d grid=%d", i,
                 a.nlr, a.nchunks, a.grid_x);
    CTDD_REQUIRE((size_t)a.B * a.Hin * a.Win * a.C * (f32 ? 4 : 2) < (1ull << 31) && (size_t)a.B * a.H * a.W * a.ldy * (f32 ? 4 : 2) < (1ull << 31),
                 CTDD_ERANGE, "wgrad[%d]: operand beyond 2 GiB (buffer offsets; 0x80000000 marks an absent vector)", i);
    const int nwc = 4 / a.nwn, Wp = a.W + 2, vn = 32 * a.nwn / epv, vc = 32 * nwc / epv;
    int KP, XP, nY, nX;
    if (a.kind == WG_3x3) {
      CTDD_REQUIRE(a.Hin == a.H && a.Win == a.W, CTDD_EINVAL, "wgrad[%d]: 3x3 kind is stride 1", i);
      KP = (a.nlr * Wp + 15) & ~15; XP = KP + 2 * Wp + 2; nY = a.nlr * a.W * vn; nX = (a.nlr + 2) * a.W * vc;
      CTDD_REQUIRE(a.nlr < 2048, CTDD_ERANGE, "wgrad[%d]: nlr=%d", i, a.nlr);
    } else {
      CTDD_REQUIRE(a.nlr % 16 == 0, CTDD_EINVAL, "wgrad[%d]: pixels per chunk must be a multiple of 16", i);
      KP = XP = a.nlr; nY = KP * vn; nX = KP * vc;
    }
    CTDD_REQUIRE(nY <= 256 * 8 && nX <= 256 * 10, CTDD_ERANGE, "wgrad[%d]: chunk of %d + %d vectors exceeds the staging slots (8 + 10 per thread)", i, nY, nX);
    const size_t l = ((size_t)KP * a.nwn + (size_t)XP * nwc) * tb + 16;     // (+ the pad unused staging slots write)
    CTDD_REQUIRE(l <= 160 * 1024 && l < (1u << 20), CTDD_ERANGE, "wgrad[%d]: %zu bytes of LDS", i, l);
    if (l > lds) lds = l;
    const int groups = ((a.N + 32 * a.nwn - 1) / (32 * a.nwn)) * ((a.C + 32 * nwc - 1) / (32 * nwc));
    if (a.grid_x > gx) gx = a.grid_x;
    if (groups > gy) gy = groups;
  }
  dim3 g((unsigned)gx, (unsigned)gy, (unsigned)n);
  hipStream_t st = (hipStream_t)stream;
  const WgradArgs* td = (const WgradArgs*)table_dev;
#define CTDD_WGRAD_LAUNCH(F_, NT_)                                          \
  {                                                                         \
    static bool done[16] = {};                                              \
    ensure_lds_ceiling((const void*)k_wgrad<F_, NT_>, done);                \
    hipLaunchKernelGGL((k_wgrad<F_, NT_>), g, dim3(256), lds, st, td);      \
  }
  if (f32) { if (nine) CTDD_WGRAD_LAUNCH(true, 9) else CTDD_WGRAD_LAUNCH(true, 1) }
  else { if (nine) CTDD_WGRAD_LAUNCH(false, 9) else CTDD_WGRAD_LAUNCH(false, 1) }
#undef CTDD_WGRAD_LAUNCH
  return finish_launch("k_wgrad");
}

extern "C" int ctdd_unet_gn_bwd(const void* args_, void* stream) {
  const GnBwdArgs& a = *(const GnBwdArgs*)args_;
  const int C = a.C1 + a.C2;
  CTDD_REQUIRE(C % 8 == 0 && a.C1 % 8 == 0 && C % a.G == 0 && C <= 2048 && a.sums, CTDD_EINVAL, "gn bwd: C=%d C1=%d G=%d", C, a.C1, a.G);
  CTDD_REQUIRE((a.s1_f32 || a.s1_bf16) && (a.da_f32 || a.da_bf16) && (a.d1_f32 || a.d1_bf16), CTDD_EINVAL, "gn bwd: null tensor");
  CTDD_REQUIRE(!(a.dsum_bn || a.dsum_n) || a.C2 == 0, CTDD_EINVAL, "gn bwd: per-sample sums of dX are for a single source");
  CTDD_REQUIRE(a.drop_p >= 0.0f && a.drop_p < 1.0f && (a.drop_p == 0.0f || a.rng), CTDD_EINVAL, "gn bwd: dropout p=%g", (double)a.drop_p);
  hipStream_t st = (hipStream_t)stream;
  // bf16 tensors: one launch, a workgroup per (sample, slab of whole groups) when such a slab fits the registers
  // (<= 8 vectors of 8 channels per thread at <= 512 threads); CTDD_GN_BWD_TWO_PASS=1 keeps the two launches (A/B runs)
  static const bool two_pass = [] { const char* e = getenv("CTDD_GN_BWD_TWO_PASS"); return e && e[0] == '1'; }();
  if (!two_pass && a.s1_bf16 && (a.C2 == 0 || a.s2_bf16) && a.da_bf16 && a.d1_bf16 && (a.C2 == 0 || a.d2_bf16) && !a.s1_f32 && !a.da_f32 &&
      !a.d1_f32 && !a.d2_f32) {
    const int cg = C / a.G;
    int L = cg;
    while (L % 8) L += cg;
    int best = 0, bo = 0, bp = 0, bv = 0;
    for (int sc = L; sc <= C; sc += L) {
      if (C % sc) continue;
      const int noct = sc / 8;
      if (noct > 512) break;
      const int npl = 512 / noct < a.HW ? 512 / noct : a.HW, nvec = (a.HW + npl - 1) / npl;
      if (nvec > 8) continue;
      const long wgs = (long)a.B * (C / sc);
      if (best == 0 || wgs >= 256) { best = sc; bo = noct; bp = npl; bv = nvec; }
      if (wgs < 256) break;
    }
    if (best > 0) {
      const int threads = max(64, ((bo * bp + 63) / 64) * 64);
      const size_t lds = (size_t)bo * bp * 80 + (size_t)8 * best * sizeof(float);      // (80-byte thread records: PST)
      const dim3 g((unsigned)a.B, (unsigned)(C / best));
      static bool attr_done[3][16] = {};
      auto go = [&](auto kernel, int slot) {
        ensure_lds_ceiling((const void*)kernel, attr_done[slot]);
        hipLaunchKernelGGL(kernel, g, dim3(threads), lds, st, a, best, bo, bp);
      };
      if (bv <= 2) go(k_gn_bwd_onepass<2>, 0);
      else if (bv <= 4) go(k_gn_bwd_onepass<4>, 1);
      else go(k_gn_bwd_onepass<8>, 2);
      return finish_launch("k_gn_bwd_onepass");
    }
  }
  const int vpp = C / 8, lanes = 256 / vpp;
  // ~8 pixels per thread and pass: slices of 8 * lanes pixels, at most ~1024 workgroups
  const int gx = max(1, min((a.HW + 8 * lanes - 1) / (8 * lanes), max(1, 1024 / max(a.B, 1))));
  hipLaunchKernelGGL(k_gn_bwd_sums, dim3(gx, a.B), dim3(256), (size_t)(8 * C + 256 * 16) * sizeof(float), st, a);
  if (int rc = finish_launch("k_gn_bwd_sums")) return rc;
  hipLaunchKernelGGL(k_gn_bwd_dx, dim3(gx, a.B), dim3(256), (size_t)(8 * C + 4 * C) * sizeof(float), st, a);
  return finish_launch("k_gn_bwd_dx");
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
  hipLaunchKernelGGL(k_sum_batch, dim3((n + 63) / 64), dim3(256), 0, (hipStream_t)stream, in, B, bstride, jstride, n, out, accumulate);
  return finish_launch("k_sum_batch");
}

extern "C" int ctdd_unet_sum_jobs(const void* jobs_dev, int njobs, void* stream) {
  CTDD_REQUIRE(jobs_dev && njobs > 0, CTDD_EINVAL, "sum_jobs: empty table");
  hipLaunchKernelGGL(k_sum_jobs, dim3(njobs), dim3(256), 0, (hipStream_t)stream, (const SumJob*)jobs_dev);
  return finish_launch("k_sum_jobs");
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
  {
    const int Tp = (a.T + 3) & ~3;
    const size_t lds4 = (size_t)(4 * ch * Tp + 3 * a.T * ch + 3 * a.T * Tp) * sizeof(float);
    static const bool old_attn = [] { const char* e = getenv("CTDD_ATTN_SMALL_OLD"); return e && e[0] == '1'; }();    // (A/B)
    if (!old_attn && ch % 4 == 0 && a.C % 4 == 0 && lds4 <= 160 * 1024) {
      static bool done4[16] = {};
      ensure_lds_ceiling((const void*)k_attn_small_bwd_t4, done4);
      hipLaunchKernelGGL(k_attn_small_bwd_t4, dim3(a.B * a.heads), dim3(256), lds4, (hipStream_t)stream, a);
      return finish_launch("k_attn_small_bwd_t4");
    }
  }
  const size_t lds = (size_t)(4 * a.T * ch + 2 * a.T * a.T) * sizeof(float);
  CTDD_REQUIRE(lds <= 160 * 1024, CTDD_ERANGE, "attention bwd tile too large (T=%d)", a.T);
  static bool done[16] = {};
  ensure_lds_ceiling((const void*)k_attn_small_bwd, done);
  hipLaunchKernelGGL(k_attn_small_bwd, dim3(a.B * a.heads), dim3(256), lds, (hipStream_t)stream, a);
  return finish_launch("k_attn_small_bwd");
}

extern "C" int64_t ctdd_unet_first_conv_wgrad_scratch(int B, int H, int Cin, int Cout) {
  return (int64_t)B * first_wgrad_bands(H) * Cout * (Cin * 9 + 1);
}
extern "C" int ctdd_unet_first_conv_wgrad(const void* args_, void* stream) {
  const FirstWgradArgs& a = *(const FirstWgradArgs*)args_;
  CTDD_REQUIRE((a.x64 || a.x32) && (a.dy_f32 || a.dy_bf16) && a.gw && a.partial, CTDD_EINVAL, "first conv wgrad: null operand");
  CTDD_REQUIRE(a.Cin >= 1 && a.Cin <= 4 && a.Cout <= 256, CTDD_ERANGE, "first conv wgrad: Cin=%d Cout=%d", a.Cin, a.Cout);
  const int gx = first_wgrad_bands(a.H), rows_per = (a.H + gx - 1) / gx, K = a.Cin * 9;
  const size_t lds = ((size_t)a.Cin * (rows_per + 2) * (a.W + 2) + (size_t)a.Cout * (K + 1)) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_first_conv_wgrad, dim3(gx, a.B), dim3(256), lds, st, a);
  if (int rc = finish_launch("k_first_conv_wgrad")) return rc;
  const int64_t row = (int64_t)a.Cout * (K + 1);
  hipLaunchKernelGGL(k_sum_batch, dim3((a.Cout * K + 63) / 64), dim3(256), 0, st, a.partial, gx * a.B, row, 1, a.Cout * K, a.gw, 0);
  if (a.gbias) hipLaunchKernelGGL(k_sum_batch, dim3((a.Cout + 63) / 64), dim3(256), 0, st, a.partial + (size_t)a.Cout * K, gx * a.B, row, 1, a.Cout, a.gbias, 0);
  return finish_launch("k_sum_batch");
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
