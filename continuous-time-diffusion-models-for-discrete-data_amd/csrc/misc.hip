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

extern "C" int ctdd_argmax(const float* logits, int N, int D, int S, int32_t* out_x, void* stream) {
  CTDD_REQUIRE(logits && out_x, CTDD_EINVAL, "null logits/out");
  CTDD_REQUIRE(N > 0 && D > 0 && S >= 1, CTDD_EINVAL, "bad sizes N=%d D=%d S=%d", N, D, S);
  int G = 1;
  while (G < S && G < 64) G <<= 1;
  const int64_t R = (int64_t)N * D;
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
