// draw.hpp -- device draw rules shared by every sampling kernel (restated in oracle/philox.py).
#pragma once
#include "common.hpp"

namespace ctdd {

constexpr float POISSON_ICDF_MAX_LAMBDA = 12.0f;   // inverse-CDF search up to here, split above
constexpr int POISSON_ICDF_KMAX = 64;
// Row rule (superposition of independent Poisson processes, exact in distribution):
//   Lambda = h * sum_s r_s <= SUPERPOSE_MAX_LAMBDA:  K ~ Poisson(Lambda) from the first uniform(s) of the
//     row's stream (Lambda > 12: sum of n = ceil(Lambda/12) draws of Poisson(Lambda/n), one uniform each),
//     then K destinations ~ Categorical(r) by inverse CDF from the following uniforms;
//   above: the same one level down -- every sub-block b of 4 consecutive destinations draws
//     K_b ~ Poisson(h * sum_{s in b} r_s) from uniform (b & 3) of Philox block DENSE_DRAW0 + (b >> 2)
//     and, if K_b > 0, K_b picks among its 4 destinations from the private stream PICK_DRAW0 + 16 b
//     (K_b for a rate > 12 is the sum of <= 64 equal parts from the stream SPLIT_DRAW0 + 16 b).
constexpr float SUPERPOSE_MAX_LAMBDA = 64.0f;
constexpr uint32_t DENSE_DRAW0 = 1024u;
constexpr uint32_t SPLIT_DRAW0 = 8192u;
constexpr uint32_t PICK_DRAW0 = 65536u;

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

// A private Philox stream: successive uniforms of (row, draw0, draw0+1, ...).
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

// Poisson draw of one (sub-block) rate.  u = its uniform.  lam > 12 is split into n <= 64 equal
// independent parts (a sum of Poissons is Poisson) drawn from the element's private stream; parts
// that are still > 12 (lam > 768: every such draw saturates the state clamp) are taken at their
// mean.  NaN / negative rates give 0 (the reference's torch.poisson raises there).
__device__ inline int poisson_element(float lam, float u, uint64_t seed, uint64_t offset, uint64_t row, int s) {
  if (!(lam > 0.0f)) return 0;
  if (u + 2.0e-7f < 1.0f - lam) return 0;                 // u < 1-lam <= exp(-lam): certainly no jump
  if (lam <= POISSON_ICDF_MAX_LAMBDA) return poisson_icdf(lam, u);
  const int n = (int)fminf(ceilf(lam / POISSON_ICDF_MAX_LAMBDA), 64.0f);
  const float lc = lam / (float)n;
  if (!(lc <= POISSON_ICDF_MAX_LAMBDA)) return (int)fminf(rintf(lam), 1.0e9f);
  PhiloxStream rs(seed, offset, row, SPLIT_DRAW0 + 16u * (uint32_t)s);   // s = sub-block index
  int k = 0;
  for (int i = 0; i < n; ++i) k += poisson_icdf(lc, rs.next());
  return k;
}

// K ~ Poisson(Lam), Lam <= SUPERPOSE_MAX_LAMBDA, from the row's stream (see the row rule above)
__device__ inline int poisson_row(float Lam, PhiloxStream& rng) {
  if (Lam <= POISSON_ICDF_MAX_LAMBDA) return poisson_icdf(Lam, rng.next());
  const int n = (int)ceilf(Lam / POISSON_ICDF_MAX_LAMBDA);
  const float lc = Lam / (float)n;
  int k = 0;
  for (int i = 0; i < n; ++i) k += poisson_icdf(lc, rng.next());
  return k;
}

// one sub-block of the dense regime: returns the number of jumps and adds sum_k (dest_k - base)
// to *move.  r0..r3 = masked rates of destinations 4b..4b+3 (same units; `sh` turns them into
// rate*h), u = the sub-block's uniform.
__device__ inline int subblock_draw(float r0, float r1, float r2, float r3, float sh, float u, uint64_t seed,
                                    uint64_t offset, uint64_t row, int b, int base, int nvalid, int* move) {
  const float tot = (r0 + r1) + (r2 + r3);
  const int K = poisson_element(sh * tot, u, seed, offset, row, b);
  if (K > 0) {
    PhiloxStream ps(seed, offset, row, PICK_DRAW0 + 16u * (uint32_t)b);
    const float c0 = r0, c1 = r0 + r1, c2 = (r0 + r1) + r2;
    for (int k = 0; k < K && k < 4096; ++k) {
      const float v = ps.next() * tot;
      int i = (v >= c0) + (v >= c1) + (v >= c2);
      i = i < nvalid - 1 ? i : nvalid - 1;
      *move += 4 * b + i - base;
    }
  }
  return K;
}


}  // namespace ctdd
