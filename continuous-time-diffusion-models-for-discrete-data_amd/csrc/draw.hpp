// draw.hpp -- device draw rules shared by every sampling kernel (restated in oracle/philox.py).
#pragma once
#include "common.hpp"

namespace ctdd {

constexpr float POISSON_ICDF_MAX_LAMBDA = 12.0f;
constexpr int POISSON_ICDF_KMAX = 64;

// Sequential-search inverse CDF of Poisson(lam), lam <= ~12.  Same operation order as
// oracle/philox.py:poisson_icdf.
__device__ inline int poisson_icdf(float lam, float u) {
  int k = 0;
  float p = expf(-lam);
  float c = p;
  while (u > c && k < POISSON_ICDF_KMAX) {
    ++k;
    p = p * lam / (float)k;
    c += p;
  }
  return k;
}

// Poisson for any lam >= 0 from a private Philox stream (row, draws base..): inverse CDF for
// small lam, Hoermann's PTRS transformed rejection (ACM TOMS 1993) above.
struct PhiloxStream {
  uint64_t seed, offset, row;
  uint32_t draw;
  u4 blk;
  int used;
  __device__ PhiloxStream(uint64_t s, uint64_t o, uint64_t r, uint32_t d0)
      : seed(s), offset(o), row(r), draw(d0), used(4) {}
  __device__ float next() {
    if (used == 4) {
      blk = philox_row(seed, offset, row, draw++);
      used = 0;
    }
    const uint32_t r = used == 0 ? blk.x : used == 1 ? blk.y : used == 2 ? blk.z : blk.w;
    ++used;
    return u01(r);
  }
};

__device__ inline int poisson_any(float lam, PhiloxStream& rng) {
  if (!(lam > 0.0f)) return 0;  // also NaN / negative -> 0 (the reference raises ValueError there)
  if (lam <= POISSON_ICDF_MAX_LAMBDA) return poisson_icdf(lam, rng.next());
  if (!(lam < 1.0e9f)) return 1000000000;
  const float slam = sqrtf(lam), loglam = logf(lam);
  const float b = 0.931f + 2.53f * slam;
  const float a = -0.059f + 0.02483f * b;
  const float invalpha = 1.1239f + 1.1328f / (b - 3.4f);
  const float vr = 0.9277f - 3.6224f / (b - 2.0f);
  for (int it = 0; it < 32; ++it) {
    const float U = rng.next() - 0.5f;
    const float V = rng.next();
    const float us = 0.5f - fabsf(U);
    const float kf = floorf((2.0f * a / us + b) * U + lam + 0.43f);
    if (us >= 0.07f && V <= vr) return (int)kf;
    if (kf < 0.0f || (us < 0.013f && V > us)) continue;
    if (logf(V) + logf(invalpha) - logf(a / (us * us) + b) <= -lam + kf * loglam - lgammaf(kf + 1.0f))
      return (int)kf;
  }
  return (int)rintf(lam);
}

}  // namespace ctdd
