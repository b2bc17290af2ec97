// misc.hip -- error state, explicit-jump update (K6a), arg-max (K10), initial state (A7), RNG hook.
#include <stdarg.h>

#include "common.hpp"

namespace ctdd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// one thread per (n,d): counts are integer-valued floats (torch.poisson output)
__global__ void k_tauleap_apply(const int32_t* __restrict__ x, const int32_t* __restrict__ xb,
                                const float* __restrict__ jn, int ordinal, int64_t R, int S,
                                int32_t* __restrict__ out, int32_t* __restrict__ changed) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= R) return;
  const int xv = x[row], base = xb ? xb[row] : xv;
  const float* j = jn + (size_t)row * S;
  long tot = 0, mv = 0;
  for (int s = 0; s < S; ++s) {
    const long k = (long)j[s];
    tot += k;
    mv += k * (s - base);
  }
  if (!ordinal && tot > 1) mv = 0;
  long xn = (long)xv + mv;
  xn = xn < 0 ? 0 : (xn > S - 1 ? S - 1 : xn);
  out[row] = (int32_t)xn;
  if (changed && xn != xv) atomicAdd(changed, 1);
}

// G lanes per row; first maximal index (torch.max(...)[1] on CPU)
__global__ __launch_bounds__(256) void k_argmax(const float* __restrict__ logits, int64_t R, int S, int G,
                                                int32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & (G - 1), gi = lane / G;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * (WAVE / G) + gi;
  const bool live = row < R;
  const float* l = logits + (size_t)(live ? row : R - 1) * S;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int s = li; s < S; s += G) {
    const float v = l[s];
    if (v > best) { best = v; bi = s; }
  }
  for (int m = G >> 1; m >= 1; m >>= 1) {
    const float ov = __shfl_xor(best, m, WAVE);
    const int oi = __shfl_xor(bi, m, WAVE);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (live && li == 0) out[row] = bi == 0x7fffffff ? 0 : bi;
}

__global__ void k_initial(const float* __restrict__ cdf, uint64_t seed, uint64_t offset, int64_t R, int S,
                          int32_t* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= R) return;
  const float u = u01(philox_row(seed, offset, (uint64_t)row, 0u).x);
  int v;
  if (!cdf) {
    v = (int)(u * (float)S);
  } else {  // first s with cdf[s] > u (binary search)
    int lo = 0, hi = S - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    v = lo;
  }
  out[row] = min(max(v, 0), S - 1);
}

__global__ void k_philox_uniform(uint64_t seed, uint64_t offset, int64_t nrows, int nblk, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows * nblk) return;
  const int64_t row = i / nblk;
  const int j = (int)(i % nblk);
  const u4 r = philox_row(seed, offset, (uint64_t)row, (uint32_t)j);
  float* o = out + (size_t)i * 4;
  o[0] = u01(r.x); o[1] = u01(r.y); o[2] = u01(r.z); o[3] = u01(r.w);
}
}  // namespace ctdd
using namespace ctdd;

extern "C" int ctdd_abi_version(void) { return 1; }
extern "C" const char* ctdd_last_error(void) { return g_err; }

extern "C" int ctdd_tauleap_apply(const int32_t* x, const int32_t* x_base, const float* jump_nums, int is_ordinal,
                                  int N, int D, int S, int32_t* out_x, int32_t* out_changed, void* stream) {
  CTDD_REQUIRE(x && jump_nums && out_x, CTDD_EINVAL, "null x/jump_nums/out");
  CTDD_REQUIRE(N > 0 && D > 0 && S >= 2, CTDD_EINVAL, "bad sizes N=%d D=%d S=%d", N, D, S);
  const int64_t R = (int64_t)N * D;
  hipLaunchKernelGGL(k_tauleap_apply, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     x_base, jump_nums, is_ordinal, R, S, out_x, out_changed);
  return finish_launch("k_tauleap_apply");
}

namespace ctdd {
// S = 256 (MNIST / CIFAR logits): a row is ONE 16-byte load per lane; a wave keeps eight rows in flight, reduces each on DPP
// (no LDS, no shuffles) and stores the eight indices from eight lanes.  HBM-bound: 1 KiB read per row.
template <int CTRL>
__device__ inline int am_dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ inline float am_dpp_f(float v) { return __builtin_bit_cast(float, am_dpp_i<CTRL>(__builtin_bit_cast(int, v))); }
__device__ inline void am_pick(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__global__ __launch_bounds__(256) void k_argmax_s256(const float* __restrict__ logits, int64_t R, int32_t* __restrict__ out) {
  constexpr int RPW = 8;
  const int lane = threadIdx.x & 63;
  const int64_t wave_g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t base = wave_g * RPW; base < R; base += nwaves * RPW) {
    float4 v[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int64_t row = base + r < R ? base + r : R - 1;
      v[r] = *(const float4*)(logits + (size_t)row * 256 + lane * 4);
    }
    int mine = 0;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      float b = v[r].x;
      int bi = lane * 4;
      if (v[r].y > b) { b = v[r].y; bi = lane * 4 + 1; }
      if (v[r].z > b) { b = v[r].z; bi = lane * 4 + 2; }
      if (v[r].w > b) { b = v[r].w; bi = lane * 4 + 3; }
      am_pick(b, bi, am_dpp_f<0xB1>(b), am_dpp_i<0xB1>(bi));          // xor 1
      am_pick(b, bi, am_dpp_f<0x4E>(b), am_dpp_i<0x4E>(bi));          // xor 2
      am_pick(b, bi, am_dpp_f<0x141>(b), am_dpp_i<0x141>(bi));        // row_half_mirror
      am_pick(b, bi, am_dpp_f<0x140>(b), am_dpp_i<0x140>(bi));        // row_mirror
      float b0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 0));
      int i0 = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
      for (int q = 1; q < 4; ++q)
        am_pick(b0, i0, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 16 * q)),
                __builtin_amdgcn_readlane(bi, 16 * q));
      mine = lane == r ? i0 : mine;
    }
    if (lane < RPW && base + lane < R) out[base + lane] = mine;
  }
}
// S <= 8 (maze, synthetic): a row per lane
__global__ __launch_bounds__(256) void k_argmax_small(const float* __restrict__ logits, int64_t R, int S, int32_t* __restrict__ out) {
  for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < R; row += (int64_t)gridDim.x * 256) {
    const float* l = logits + (size_t)row * S;
    float best = l[0];
    int bi = 0;
    for (int s_ = 1; s_ < S; ++s_) {
      const float v = l[s_];
      if (v > best) { best = v; bi = s_; }
    }
    out[row] = bi;
  }
}
}  // namespace ctdd

extern "C" int ctdd_argmax(const float* logits, int N, int D, int S, int32_t* out_x, void* stream) {
  CTDD_REQUIRE(logits && out_x, CTDD_EINVAL, "null logits/out");
  CTDD_REQUIRE(N > 0 && D > 0 && S >= 1, CTDD_EINVAL, "bad sizes N=%d D=%d S=%d", N, D, S);
  const int64_t R = (int64_t)N * D;
  if (S == 256) {
    const int64_t want = (R + 31) / 32;                     // eight rows per wave and iteration
    hipLaunchKernelGGL(k_argmax_s256, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, (hipStream_t)stream, logits, R, out_x);
    return finish_launch("k_argmax_s256");
  }
  if (S <= 8) {
    const int64_t want = (R + 255) / 256;
    hipLaunchKernelGGL(k_argmax_small, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream, logits, R, S, out_x);
    return finish_launch("k_argmax_small");
  }
  int G = 1;
  while (G < S && G < 64) G <<= 1;
  const int rows_per_wg = 4 * (64 / G);
  hipLaunchKernelGGL(k_argmax, dim3((unsigned)((R + rows_per_wg - 1) / rows_per_wg)), dim3(256), 0,
                     (hipStream_t)stream, logits, R, S, G, out_x);
  return finish_launch("k_argmax");
}

extern "C" int ctdd_initial_samples(const float* cdf, uint64_t seed, uint64_t offset, int N, int D, int S,
                                    int32_t* out_x, void* stream) {
  CTDD_REQUIRE(out_x, CTDD_EINVAL, "null out");
  CTDD_REQUIRE(N > 0 && D > 0 && S >= 2, CTDD_EINVAL, "bad sizes N=%d D=%d S=%d", N, D, S);
  const int64_t R = (int64_t)N * D;
  hipLaunchKernelGGL(k_initial, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cdf, seed,
                     offset, R, S, out_x);
  return finish_launch("k_initial");
}

extern "C" int ctdd_philox_uniform(uint64_t seed, uint64_t offset, int64_t nrows, int nblk, float* out, void* stream) {
  CTDD_REQUIRE(out && nrows > 0 && nblk > 0, CTDD_EINVAL, "bad arguments");
  const int64_t n = nrows * nblk;
  hipLaunchKernelGGL(k_philox_uniform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed,
                     offset, nrows, nblk, out);
  return finish_launch("k_philox_uniform");
}
