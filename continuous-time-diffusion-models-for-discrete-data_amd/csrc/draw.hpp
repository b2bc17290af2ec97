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
//   above: one level down -- sub-block b of 4 consecutive destinations with rate lam_b = h * sum_{s in b} r_s:
//     lam_b <= 12: K_b ~ Poisson(lam_b) from uniform (b & 3) of Philox block DENSE_DRAW0 + (b >> 2) and, if K_b > 0,
//       K_b picks among its 4 destinations from the private stream PICK_DRAW0 + 16 b;
//     lam_b > 12: the reference's own form, one independent Poisson(h r_s) per destination from the stream
//       SPLIT_DRAW0 + 16 b (inverse CDF up to 12, Hoermann's PTRS transformed rejection above: O(1) for any rate --
//       random-init logistic heads produce h r_s ~ 1e6, where splitting or picking one jump at a time never ends).
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
    // (bit selects: the chain of equality tests became four branches in the callers' loops)
    const uint32_t r01 = (used & 1) ? blk.y : blk.x, r23 = (used & 1) ? blk.w : blk.z;
    const uint32_t r = (used & 2) ? r23 : r01;
    ++used;
    return u01(r);
  }
};

// Poisson(lam), lam > 12: PTRS (W. Hoermann, "The transformed rejection method for generating Poisson random
// variables", 1993; the algorithm numpy and ATen use for large rates).  Two uniforms per trial, ~1.1-1.3 trials.
// The acceptance test runs in fp64 (k log(lam) - lgamma(k+1) cancels ~1e7-sized terms at lam ~ 1e6).
// log(k!) for an integral k >= 0 in fp64 without the library lgamma (whose register footprint -- ~170 VGPRs -- set the
// occupancy of every kernel that can reach the heavy-rate draw): a 16-entry table, Stirling's series above it
// (x = k + 1 >= 17: the first omitted term, 1/(1680 x^7), is below 3e-12).
__device__ const double CTDD_LOG_FACTORIAL[16] = {0.0, 0.0, 0.693147180559945, 1.7917594692280554, 3.178053830347945, 4.787491742782047, 6.579251212010102, 8.525161361065415, 10.604602902745249, 12.801827480081467, 15.104412573075514, 17.502307845873887, 19.987214495661885, 22.55216385312342, 25.191221182738683, 27.89927138384089};
__device__ inline double log_factorial(double kf) {
  if (kf < 16.0) return CTDD_LOG_FACTORIAL[(int)kf];
  const double x = kf + 1.0, xi = 1.0 / x, xi2 = xi * xi;
  return (x - 0.5) * log(x) - x + 0.91893853320467274178 + xi * (1.0 / 12.0 - xi2 * (1.0 / 360.0 - xi2 * (1.0 / 1260.0)));
}

__device__ inline int poisson_ptrs(float lamf, PhiloxStream& rs) {
  const double lam = (double)lamf, slam = sqrt(lam), loglam = log(lam);
  const double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
  const double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
  for (int it = 0; it < 32; ++it) {
    const double U = (double)rs.next() - 0.5, V = (double)rs.next();
    const double us = 0.5 - fabs(U);
    const double kf = floor((2.0 * a / us + b) * U + lam + 0.43);
    if (us >= 0.07 && V <= vr) return (int)fmin(kf, 1.0e9);
    if (kf < 0.0 || (us < 0.013 && V > us)) continue;
    if (log(V * invalpha / (a / (us * us) + b)) <= -lam + kf * loglam - log_factorial(kf)) return (int)fmin(kf, 1.0e9);
  }
  return (int)fmin(rint(lam), 1.0e9);                     // (32 rejections in a row: p < 1e-20)
}

// Sub-block count for lam <= 12 from its shared uniform; NaN / negative rates give 0 (torch.poisson raises there).
__device__ inline int poisson_element(float lam, float u) {
  if (!(lam > 0.0f)) return 0;
  if (u + 2.0e-7f < 1.0f - lam) return 0;                 // u < 1-lam <= exp(-lam): certainly no jump
  return poisson_icdf(lam, u);
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
// to *move (64-bit; the caller clamps the row total to +-S before the state update).  r0..r3 = masked rates of destinations 4b..4b+3 (same units; `sh` turns them into
// rate*h), u = the sub-block's uniform.
__device__ inline int subblock_draw(float r0, float r1, float r2, float r3, float sh, float u, uint64_t seed,
                                    uint64_t offset, uint64_t row, int b, int base, int nvalid, long long* move) {
  const float tot = (r0 + r1) + (r2 + r3);
  const float lam = sh * tot;
  if (!(lam > POISSON_ICDF_MAX_LAMBDA)) {
    const int K = poisson_element(lam, u);
    if (K > 0) {
      PhiloxStream ps(seed, offset, row, PICK_DRAW0 + 16u * (uint32_t)b);
      const float c0 = r0, c1 = r0 + r1, c2 = (r0 + r1) + r2;
      for (int k = 0; k < K; ++k) {
        const float v = ps.next() * tot;
        int i = (v >= c0) + (v >= c1) + (v >= c2);
        i = i < nvalid - 1 ? i : nvalid - 1;
        *move += 4 * b + i - base;
      }
    }
    return K;
  }
  // heavy sub-block: independent Poisson(h r_s) per destination (the reference's own draw)
  PhiloxStream rs(seed, offset, row, SPLIT_DRAW0 + 16u * (uint32_t)b);
  const float rr[4] = {r0, r1, r2, r3};
  int K = 0;
  long long mv = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float li = sh * rr[i];
    if (i >= nvalid || !(li > 0.0f)) continue;
    const int k = li <= POISSON_ICDF_MAX_LAMBDA ? poisson_icdf(li, rs.next()) : poisson_ptrs(li, rs);
    K = (int)min((long long)K + k, (long long)1 << 30);
    mv += (long long)k * (4 * b + i - base);
  }
  *move += mv;                                              // 64-bit: |k (s - x)| reaches 2.5e11 per destination
  return K;
}


}  // namespace ctdd
