// common.hpp -- shared host/device helpers of libctdd (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ctdd.h"

namespace ctdd {

void set_error(const char* fmt, ...);

#define CTDD_REQUIRE(cond, code, ...)       \
  do {                                      \
    if (!(cond)) {                          \
      ::ctdd::set_error(__VA_ARGS__);       \
      return (code);                        \
    }                                       \
  } while (0)

inline int finish_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return CTDD_EHIP;
  }
  return CTDD_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute of a kernel: raise it to the CU's 160 KiB once per
// (device, kernel instantiation) -- `done` is the call site's static flag array -- so that a later launch of the same
// instantiation with a larger LDS request (another resolution, a second model) or on another GPU of the process is covered.
// Called before the launch, never inside a stream capture for the first time (the engines warm every plan up eagerly).
inline void ensure_lds_ceiling(const void* kernel, bool (&done)[16]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (!done[dev]) {
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done[dev] = true;
  }
}

constexpr int WAVE = 64;

// ---------------------------------------------------------------- Philox4x32-10 (oracle/philox.py)
struct u4 {
  uint32_t x, y, z, w;
};

template <int ROUNDS>
__host__ __device__ inline u4 philox4x32_r(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return u4{c0, c1, c2, c3};
}

__host__ __device__ inline u4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return philox4x32_r<10>(c0, c1, c2, c3, k0, k1);
}
// Seven rounds (Random123's smallest Crush-resistant Philox4x32): the attention-probability dropout of the hollow training
// kernels only, where the generator is ~40 % of the instructions and nothing replays the stream outside those kernels.
__host__ __device__ inline u4 philox_row7(uint64_t seed, uint64_t offset, uint64_t row, uint32_t draw) {
  return philox4x32_r<7>((uint32_t)row, (uint32_t)(row >> 32), (uint32_t)offset, draw, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// counter = (row_lo, row_hi, offset, draw), key = seed
__host__ __device__ inline u4 philox_row(uint64_t seed, uint64_t offset, uint64_t row, uint32_t draw) {
  return philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), (uint32_t)offset, draw,
                       (uint32_t)seed, (uint32_t)(seed >> 32));
}

// uint32 -> (0,1), every step exact in fp32: ((r>>9)+0.5)*2^-23
__host__ __device__ inline float u01(uint32_t r) {
  return ((float)(r >> 9) + 0.5f) * 1.1920928955078125e-07f;
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, WAVE));
  return v;
}

}  // namespace ctdd
