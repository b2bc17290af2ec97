// steps_generic.hip -- reverse rates, SDDM log-probs and the sampler step kernels for ANY S.
//
// Layout: a row = one (n,d) pair = S contiguous fp32 logits.  G = min(64, pow2ceil(S)) lanes
// of a wave own one row (element s = li + k*G, k < EPT), so every row reduction is an
// in-wave xor-shuffle over masks < G and small-S problems (maze S=3, synthetic S=2) pack
// 16-32 rows per wave.  The S x S contraction broadcasts w[s0] by shuffle and reads the
// q_{t|0} table through L1/L2.  This is the correctness-first path for every S; S = 256 (MNIST,
// CIFAR) is overridden by the LDS/MFMA-tiled kernel in steps_s256.hip.
//
// Reference semantics: lib/sampling/sampling.py:31-78 (rates), 119-160 (tau-leap), 278-293
// (LBJF), 417-453 (midpoint); lib/models/model_utils.py:30-60 (log-probs).
#include "draw.hpp"

namespace ctdd {

enum Mode { MODE_RATES = 0, MODE_LOGPROB = 1, MODE_TAULEAP = 2, MODE_LBJF = 3, MODE_MIDPOINT = 4,
            MODE_DRAW_ONLY = 5, MODE_EXACT = 6 };

struct StepArgs {
  const float* logits;   // (N,D,S)   [MODE_DRAW_ONLY: the rates themselves]
  const int32_t* x;      // (N,D)
  const int32_t* x_base; // (N,D) or null
  const float* qt0;      // (nT,S,S) or null (direct)
  const float* rate;     // (nT,S,S) table, or (S,S) base rate scaled by beta
  const int32_t* tidx;   // (N) or null
  const float* E;        // (N*D,S) explicit exponential noise or null
  float beta, eps, h;
  uint32_t flags;
  uint64_t seed, offset;
  int N, D, S, G;
  int branch, logit_type, mode;
  int pre_rates;         // `logits` holds the masked reverse rates already (from the S = 256 matrix-core kernel): any tail mode
  float* out_a;          // rates | ll_all | probs
  float* out_b;          // ratio | ll_xt
  int32_t* out_x;
  int32_t* out_changed;
};

// ---- reductions / scans over the G lanes of a row on DPP (data-parallel primitives: one VALU op per stage) instead of
// ds_bpermute shuffles (an LDS-crossbar round trip per stage: six dependent ones per 64-lane reduction, ~10 reductions and
// a 28-stage scan per row made the shuffles, not memory, the bound of every mode).  G is uniform over the launch, so the
// stage tests are scalar branches.  Stages inside a 16-lane row: quad_perm xor 1 / xor 2, row_half_mirror, row_mirror (the
// operations are commutative, so a mirror does what an xor would); across rows: v_readlane of one lane per row.
template <int CTRL>
__device__ inline int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ inline float dpp_f(float v) { return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v))); }
__device__ inline float rdlane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;

__device__ inline float grp_sum(float v, int G) {
  if (G >= 2) v += dpp_f<DPP_XOR1>(v);
  if (G >= 4) v += dpp_f<DPP_XOR2>(v);
  if (G >= 8) v += dpp_f<DPP_HALF_MIRROR>(v);
  if (G >= 16) v += dpp_f<DPP_MIRROR>(v);
  if (G >= 32) {
    const float r0 = rdlane_f(v, 0), r1 = rdlane_f(v, 16), r2 = rdlane_f(v, 32), r3 = rdlane_f(v, 48);
    v = G == 64 ? (r0 + r1) + (r2 + r3) : ((threadIdx.x & 32) ? r2 + r3 : r0 + r1);
  }
  return v;
}
__device__ inline int grp_sum_i(int v, int G) {
  if (G >= 2) v += dpp_i<DPP_XOR1>(v);
  if (G >= 4) v += dpp_i<DPP_XOR2>(v);
  if (G >= 8) v += dpp_i<DPP_HALF_MIRROR>(v);
  if (G >= 16) v += dpp_i<DPP_MIRROR>(v);
  if (G >= 32) {
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16), r2 = __builtin_amdgcn_readlane(v, 32),
              r3 = __builtin_amdgcn_readlane(v, 48);
    v = G == 64 ? (r0 + r1) + (r2 + r3) : ((threadIdx.x & 32) ? r2 + r3 : r0 + r1);
  }
  return v;
}
__device__ inline float grp_max(float v, int G) {
  if (G >= 2) v = fmaxf(v, dpp_f<DPP_XOR1>(v));
  if (G >= 4) v = fmaxf(v, dpp_f<DPP_XOR2>(v));
  if (G >= 8) v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
  if (G >= 16) v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
  if (G >= 32) {
    const float r0 = rdlane_f(v, 0), r1 = rdlane_f(v, 16), r2 = rdlane_f(v, 32), r3 = rdlane_f(v, 48);
    v = G == 64 ? fmaxf(fmaxf(r0, r1), fmaxf(r2, r3)) : ((threadIdx.x & 32) ? fmaxf(r2, r3) : fmaxf(r0, r1));
  }
  return v;
}
// (value, index) arg-max with first-index tie-break (torch.argmax)
__device__ inline void argmax_pick(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__device__ inline void grp_argmax(float& v, int& i, int G) {
  if (G >= 2) { const float ov = dpp_f<DPP_XOR1>(v); const int oi = dpp_i<DPP_XOR1>(i); argmax_pick(v, i, ov, oi); }
  if (G >= 4) { const float ov = dpp_f<DPP_XOR2>(v); const int oi = dpp_i<DPP_XOR2>(i); argmax_pick(v, i, ov, oi); }
  if (G >= 8) { const float ov = dpp_f<DPP_HALF_MIRROR>(v); const int oi = dpp_i<DPP_HALF_MIRROR>(i); argmax_pick(v, i, ov, oi); }
  if (G >= 16) { const float ov = dpp_f<DPP_MIRROR>(v); const int oi = dpp_i<DPP_MIRROR>(i); argmax_pick(v, i, ov, oi); }
  if (G >= 32) {
    const int hi = (G == 64) ? 0 : (int)(threadIdx.x & 32);          // first row of this lane's group
    float bv = rdlane_f(v, 0), b1 = rdlane_f(v, 16), b2 = rdlane_f(v, 32), b3 = rdlane_f(v, 48);
    int bi = __builtin_amdgcn_readlane(i, 0), i1 = __builtin_amdgcn_readlane(i, 16), i2 = __builtin_amdgcn_readlane(i, 32),
        i3 = __builtin_amdgcn_readlane(i, 48);
    if (G == 64) {
      argmax_pick(bv, bi, b1, i1); argmax_pick(bv, bi, b2, i2); argmax_pick(bv, bi, b3, i3);
      v = bv; i = bi;
    } else {
      argmax_pick(bv, bi, b1, i1);
      argmax_pick(b2, i2, b3, i3);
      v = hi ? b2 : bv; i = hi ? i2 : bi;
    }
  }
}
// inclusive prefix sum over the G lanes of a row group, lane order (li = lane within the group)
__device__ inline float grp_scan(float v, int G, int li) {
  const int pos = li & 15;                                             // place inside the 16-lane DPP row
  if (G >= 2) { const float t = dpp_f<0x111>(v); v += pos >= 1 ? t : 0.0f; }    // row_shr:1 (lanes without a source read 0)
  if (G >= 4) { const float t = dpp_f<0x112>(v); v += pos >= 2 ? t : 0.0f; }
  if (G >= 8) { const float t = dpp_f<0x114>(v); v += pos >= 4 ? t : 0.0f; }
  if (G >= 16) { const float t = dpp_f<0x118>(v); v += pos >= 8 ? t : 0.0f; }
  if (G >= 32) {
    const float t0 = rdlane_f(v, 15), t1 = rdlane_f(v, 31), t2 = rdlane_f(v, 47);
    const int r = (threadIdx.x >> 4) & 3;
    if (G == 64) v += r == 0 ? 0.0f : r == 1 ? t0 : r == 2 ? t0 + t1 : (t0 + t1) + t2;
    else v += r == 1 ? t0 : r == 3 ? t2 : 0.0f;
  }
  return v;
}

constexpr int ROWS_PER_WAVE = 4;
template <int EPT>
__global__ __launch_bounds__(256) void k_rows(const StepArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int G = a.G, S = a.S;
  const int li = lane & (G - 1), gi = lane / G, gbase = lane - li;
  const int64_t R = (int64_t)a.N * a.D;
  // A wave walks ROWS_PER_WAVE consecutive row groups; the state and the logits of the NEXT group are requested before the
  // current one is worked on (one row per wave and kernel lifetime left every load latency of the x -> table -> logits
  // chain exposed: 7 us per row at S = 256).
  const int64_t first = ((int64_t)blockIdx.x * 4 + wave) * ROWS_PER_WAVE * (WAVE / G) + gi;
  int nx_x = 0, nx_xb = 0;
  float nx_l[EPT];
  auto prefetch = [&](int64_t r) {
    const int64_t rc = r < R ? r : R - 1;
    nx_x = a.x[rc];
    nx_xb = a.x_base ? a.x_base[rc] : 0;
    const float* lr = a.logits + (size_t)rc * S;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int s = li + k * G;
      nx_l[k] = s < S ? lr[s] : 0.0f;
    }
  };
  prefetch(first);
  for (int it = 0; it < ROWS_PER_WAVE; ++it) {
  const int64_t row = first + (int64_t)it * (WAVE / G);
  if (row - gi >= R) break;                                   // (uniform over the wave: the whole group is past the end)
  const bool live = row < R;
  const int64_t rowc = live ? row : R - 1;
  const int cur_x = nx_x, cur_xb = nx_xb;
  float raw[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) raw[k] = nx_l[k];
  if (it + 1 < ROWS_PER_WAVE) prefetch(row + (WAVE / G));
  const int n = (int)(rowc / a.D);
  const int tbl = a.tidx ? a.tidx[n] : 0;
  const float* qt0 = a.qt0 ? a.qt0 + (size_t)tbl * S * S : nullptr;
  const float* rate = a.rate ? a.rate + (size_t)tbl * S * S : nullptr;
  // xcur = state the move is added to; xv = state the rates are evaluated at and measured from
  // (x' in stage 2 of the midpoint sampler, sampling.py:459-503; otherwise the same state)
  const int xcur = min(max(cur_x, 0), S - 1);
  const int xv = a.x_base ? min(max(cur_xb, 0), S - 1) : xcur;

  float rr[EPT];     // reverse rates (own state NOT zeroed)
  float ratio[EPT];

  if (a.mode == MODE_DRAW_ONLY || a.pre_rates) {
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int s = li + k * G;
      rr[k] = s < S ? raw[k] : 0.0f;
      ratio[k] = 0.0f;
    }
  } else {
    // ---- softmax pieces
    float l[EPT], e[EPT];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int s = li + k * G;
      l[k] = s < S ? raw[k] : -INFINITY;
      m = fmaxf(m, l[k]);
    }
    m = grp_max(m, G);
    float z = 0.0f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int s = li + k * G;
      e[k] = s < S ? expf(l[k] - m) : 0.0f;
      z += e[k];
    }
    z = grp_sum(z, G);

    if (a.mode == MODE_EXACT) {
      // exact one-step posterior (sampling.py:975-1061): post[s] = (sum_s0 p[s0] q_lo[s0][s]) q_step[s][x]  (qt0 = q_{t-h|0},
      // rate = q_{t|t-h}), normalised, one categorical draw by the exponential race of the LBJF step
      float acc[EPT];
#pragma unroll
      for (int k = 0; k < EPT; ++k) acc[k] = 0.0f;
#pragma unroll
      for (int k0 = 0; k0 < EPT; ++k0) {
        for (int j = 0; j < G; ++j) {
          const int s0 = j + k0 * G;
          if (s0 >= S) break;
          const float ps = G == 1 ? e[k0] / z : __shfl(e[k0] / z, gbase + j, WAVE);
          const float* qrow = qt0 + (size_t)s0 * S;
#pragma unroll
          for (int k = 0; k < EPT; ++k) {
            const int s = li + k * G;
            if (s < S) acc[k] = fmaf(ps, qrow[s], acc[k]);
          }
        }
      }
      float tot = 0.0f;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int s = li + k * G;
        acc[k] = s < S ? acc[k] * rate[(size_t)s * S + xv] : 0.0f;
        tot += acc[k];
      }
      tot = grp_sum(tot, G);
      float best = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int s = li + k * G;
        if (s < S) {
          const float pr = acc[k] / tot;
          if (live && a.out_a) a.out_a[(size_t)row * S + s] = pr;
          float Ev;
          if (a.E) Ev = a.E[(size_t)rowc * S + s];
          else Ev = -logf(u01(philox_row(a.seed, a.offset, (uint64_t)rowc, (uint32_t)s).x));
          const float v = pr / Ev;
          if (v > best || (v == best && s < bi)) { best = v; bi = s; }
        }
      }
      grp_argmax(best, bi, G);
      if (live && li == 0) {
        a.out_x[row] = bi;
        if (a.out_changed && bi != xcur) atomicAdd(a.out_changed, 1);
      }
      continue;
    }
    if (a.branch == CTDD_BRANCH_CTELBO && a.mode != MODE_LOGPROB) {
      // w[s0] = softmax[s0] / (qt0[s0][x] + eps);  ratio[s] = sum_s0 w[s0] qt0[s0][s]
      float w[EPT];
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int s = li + k * G;
        w[k] = s < S ? (e[k] / z) / (qt0[(size_t)s * S + xv] + a.eps) : 0.0f;
        ratio[k] = 0.0f;
      }
#pragma unroll
      for (int k0 = 0; k0 < EPT; ++k0) {
        for (int j = 0; j < G; ++j) {
          const int s0 = j + k0 * G;
          if (s0 >= S) break;
          const float ws = G == 1 ? w[k0] : __shfl(w[k0], gbase + j, WAVE);
          const float* qrow = qt0 + (size_t)s0 * S;
#pragma unroll
          for (int k = 0; k < EPT; ++k) {
            const int s = li + k * G;
            if (s < S) ratio[k] = fmaf(ws, qrow[s], ratio[k]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int s = li + k * G;
        const float fwd = s < S ? a.beta * rate[(size_t)s * S + xv] : 0.0f;   // rate[s][x]
        rr[k] = fwd * ratio[k];
      }
    } else {
      // ---- SDDM branch: ll_all by logit_type, ratio = exp(ll_all - ll_xt), R^ = ratio * rate[x][s]
      float ll[EPT];
      const float logz = logf(z);
      if (a.logit_type == CTDD_LOGIT_DIRECT) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) ll[k] = l[k] - m - logz;
      } else if (a.logit_type == CTDD_LOGIT_REVERSE_PROB) {
        float acc[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) acc[k] = 0.0f;
#pragma unroll
        for (int k0 = 0; k0 < EPT; ++k0) {
          for (int j = 0; j < G; ++j) {
            const int s0 = j + k0 * G;
            if (s0 >= S) break;
            const float ps = G == 1 ? e[k0] / z : __shfl(e[k0] / z, gbase + j, WAVE);
            const float* qrow = qt0 + (size_t)s0 * S;
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
              const int s = li + k * G;
              if (s < S) acc[k] = fmaf(ps, qrow[s], acc[k]);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) ll[k] = logf(acc[k] + 1e-35f);
      } else {  // reverse_logscale: logsumexp_s0(log_p0t[s0] + log qt0[s0][s])
        float mx[EPT], sm[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) { mx[k] = -INFINITY; sm[k] = 0.0f; }
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
          for (int k0 = 0; k0 < EPT; ++k0) {
            for (int j = 0; j < G; ++j) {
              const int s0 = j + k0 * G;
              if (s0 >= S) break;
              const float lp = G == 1 ? l[k0] - m - logz : __shfl(l[k0] - m - logz, gbase + j, WAVE);
              const float* qrow = qt0 + (size_t)s0 * S;
#pragma unroll
              for (int k = 0; k < EPT; ++k) {
                const int s = li + k * G;
                if (s < S) {
                  const float q = qrow[s];
                  const float t = lp + (q <= 1e-35f ? -1e9f : logf(q));
                  if (pass == 0) mx[k] = fmaxf(mx[k], t);
                  else sm[k] += expf(t - mx[k]);
                }
              }
            }
          }
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) ll[k] = mx[k] + logf(sm[k]);
      }
      // ll_xt = ll_all at the current state
      float sel = 0.0f;
#pragma unroll
      for (int k = 0; k < EPT; ++k) sel = (k == xv / G) ? ll[k] : sel;
      const float ll_xt = G == 1 ? sel : __shfl(sel, gbase + (xv & (G - 1)), WAVE);
      if (a.mode == MODE_LOGPROB) {
        if (live) {
#pragma unroll
          for (int k = 0; k < EPT; ++k) {
            const int s = li + k * G;
            if (s < S) a.out_a[(size_t)row * S + s] = ll[k];
          }
          if (li == 0) a.out_b[row] = ll_xt;
        }
        continue;
      }
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int s = li + k * G;
        ratio[k] = expf(ll[k] - ll_xt);
        const float fwd = s < S ? a.beta * rate[(size_t)xv * S + s] : 0.0f;   // rate[x][s]
        rr[k] = s < S ? ratio[k] * fwd : 0.0f;
      }
    }
  }

  if (a.mode == MODE_RATES) {
    if (live) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int s = li + k * G;
        if (s < S) {
          a.out_a[(size_t)row * S + s] = rr[k];
          if (a.out_b) a.out_b[(size_t)row * S + s] = ratio[k];
        }
      }
    }
    continue;
  }

  // ---- mask the own state (sampling.py:127-128); corrector adds the x -> s forward rate first
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int s = li + k * G;
    if ((a.flags & CTDD_STEP_CORRECTOR) && s < S) rr[k] += a.beta * rate[(size_t)xv * S + s];
    if (s == xv || s >= S) rr[k] = 0.0f;
  }
  const int base = xv;

  if (a.mode == MODE_MIDPOINT) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) acc += rr[k] * (float)(li + k * G - xv);
    acc = grp_sum(acc, G);
    const int change = (int)rintf(a.h * acc);   // a.h carries float(0.5*h)
    if (live && li == 0) a.out_x[row] = min(max(xv + change, 0), S - 1);
    continue;
  }

  if (a.mode == MODE_LBJF) {
    // P = h*R^ + clip(1 - h*sum, 0) at own state; normalise; Categorical(logits = log(P + 1e-35))
    float off = 0.0f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) off += rr[k];
    off = grp_sum(off, G);
    const float diag = fmaxf(1.0f - a.h * off, 0.0f);
    float P[EPT], tot = 0.0f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int s = li + k * G;
      P[k] = s < S ? (s == xv ? diag : rr[k] * a.h) : 0.0f;
      tot += P[k];
    }
    tot = grp_sum(tot, G);
    float lg[EPT], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int s = li + k * G;
      lg[k] = s < S ? logf(P[k] / tot + 1e-35f) : -INFINITY;
      mx = fmaxf(mx, lg[k]);
    }
    mx = grp_max(mx, G);
    float se = 0.0f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) se += (li + k * G < S) ? expf(lg[k] - mx) : 0.0f;
    se = grp_sum(se, G);
    const float lse = mx + logf(se);
    // probs = softmax(lg - lse)
    float pr[EPT], mx2 = -INFINITY;
#pragma unroll
    for (int k = 0; k < EPT; ++k) { lg[k] -= lse; mx2 = fmaxf(mx2, lg[k]); }
    mx2 = grp_max(mx2, G);
    float se2 = 0.0f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) { pr[k] = (li + k * G < S) ? expf(lg[k] - mx2) : 0.0f; se2 += pr[k]; }
    se2 = grp_sum(se2, G);
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int s = li + k * G;
      if (s < S) {
        pr[k] /= se2;
        if (live && a.out_a) a.out_a[(size_t)row * S + s] = pr[k];
        float Ev;
        if (a.E) Ev = a.E[(size_t)rowc * S + s];
        else Ev = -logf(u01(philox_row(a.seed, a.offset, (uint64_t)rowc, (uint32_t)s).x));
        const float v = pr[k] / Ev;
        if (v > best || (v == best && s < bi)) { best = v; bi = s; }
      }
    }
    grp_argmax(best, bi, G);
    if (live && li == 0) {
      a.out_x[row] = bi;
      if (a.out_changed && bi != xv) atomicAdd(a.out_changed, 1);
    }
    continue;
  }

  // ---- MODE_TAULEAP / MODE_DRAW_ONLY: K ~ Poisson(h * sum rr), K destinations ~ Categorical(rr)
  float T = 0.0f;
#pragma unroll
  for (int k = 0; k < EPT; ++k) T += rr[k];
  T = grp_sum(T, G);
  const float Lam = T * a.h;
  const bool ordinal = a.flags & CTDD_STEP_ORDINAL;
  int jump = 0, njumps = 0;                                   // njumps: jump events drawn for this dimension (sum_s k_s)
  if (Lam > 0.0f && Lam <= SUPERPOSE_MAX_LAMBDA) {
    PhiloxStream rng(a.seed, a.offset, (uint64_t)rowc, 0u);   // identical in every lane of the row
    const int K = poisson_row(Lam, rng);
    njumps = K;
    if (K > 0 && (ordinal || K == 1)) {
      // inclusive prefix sums of rr in s order (s = li + k*G)
      float c[EPT], carry = 0.0f;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        c[k] = grp_scan(rr[k], G, li) + carry;
        carry = G == 64 ? rdlane_f(c[k], 63) : G == 1 ? c[k] : __shfl(c[k], gbase + G - 1, WAVE);
      }
      for (int j = 0; j < K; ++j) {
        const float target = rng.next() * T;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) cnt += (li + k * G < S && c[k] <= target) ? 1 : 0;
        cnt = grp_sum_i(cnt, G);
        jump += min(cnt, S - 1) - base;
      }
    }
  } else if (Lam > SUPERPOSE_MAX_LAMBDA) {
    // dense regime: per sub-block of 4 consecutive destinations (draw.hpp: subblock_draw).  The
    // lane owning destination 4b (element s = li + k*G with s % 4 == 0) gathers its three
    // neighbours by shuffle (G >= 4) or from its own slots (G < 4) and draws for the sub-block.
    int cnt = 0;
    long long jl = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int s = li + k * G;
      float r1, r2, r3;
      if (G >= 4) {                  // the owner of 4b sits at quad position 0: its neighbours are the quad's other three lanes
        r1 = dpp_f<0x55>(rr[k]); r2 = dpp_f<0xAA>(rr[k]); r3 = dpp_f<0xFF>(rr[k]);
      } else if (G == 2) {           // s, s+1 in lanes li, li+1 of slot k; s+2, s+3 in slot k+1
        r1 = __shfl(rr[k], lane + 1, WAVE);
        const float nx = k + 1 < EPT ? rr[k + 1] : 0.0f;
        r2 = nx; r3 = __shfl(nx, lane + 1, WAVE);
      } else {                       // G == 1: consecutive slots of this lane
        r1 = k + 1 < EPT ? rr[k + 1] : 0.0f; r2 = k + 2 < EPT ? rr[k + 2] : 0.0f; r3 = k + 3 < EPT ? rr[k + 3] : 0.0f;
      }
      if ((s & 3) == 0 && s < S) {
        if (s + 1 >= S) r1 = 0.0f;
        if (s + 2 >= S) r2 = 0.0f;
        if (s + 3 >= S) r3 = 0.0f;
        const int b = s >> 2;
        const u4 blk = philox_row(a.seed, a.offset, (uint64_t)rowc, DENSE_DRAW0 + (uint32_t)(b >> 2));
        const uint32_t w = (b & 3) == 0 ? blk.x : (b & 3) == 1 ? blk.y : (b & 3) == 2 ? blk.z : blk.w;
        cnt += min(subblock_draw(rr[k], r1, r2, r3, a.h, u01(w), a.seed, a.offset, (uint64_t)rowc, b, base,
                                 min(4, S - s), &jl), 1 << 20);
      }
    }
    cnt = grp_sum_i(cnt, G);
    for (int o = G >> 1; o >= 1; o >>= 1) jl += __shfl_xor(jl, o, WAVE);
    jl = jl > S ? S : (jl < -S ? -S : jl);                    // |jump| >= S - 1 saturates the state clamp either way
    jump = (ordinal || cnt <= 1) ? (int)jl : 0;
    njumps = cnt;
  }
  if (live && li == 0) {
    const int xn = min(max(xcur + jump, 0), S - 1);
    a.out_x[row] = xn;
    const bool moved = (a.flags & CTDD_STEP_COUNT_RAW) ? (jump != 0) : (xn != xcur);
    if (a.out_changed && moved) atomicAdd(a.out_changed, 1);
    if (a.out_changed && (a.flags & CTDD_STEP_COUNT_JUMPS)) {  // sampling.py:489-495: dimensions with >= 1 and with > 1 jump events
      if (njumps > 0) atomicAdd(a.out_changed + 1, 1);
      if (njumps > 1) atomicAdd(a.out_changed + 2, 1);
    }
  }
  }  // rows of this wave
}


// ============================================================================ S <= 8 (maze S = 3, synthetic S = 2): a row per lane, FOUR rows per thread
// The fused sampler steps (one q_{t|0} / rate table for the whole launch) at small S move 4 S + 8 bytes per row: the generic
// kernel above spends ~1000 instructions on each (run-time S / mode / group-size tests, per-row table addressing, 4-byte loads).
// Here S and the mode are compile-time, a thread owns four CONSECUTIVE rows (their 4 S logits are S aligned 16-byte loads, the
// states one int4 each way), both tables sit in LDS, and every row loop is unrolled straight-line code in the operation order
// of k_rows with G = 1 (sequential sums in s order -- for S <= 4 bit-identical to it, for 5 <= S <= 8 the oracle's own order
// where k_rows uses a lane tree).  Same Philox streams, same draw rules (draw.hpp).
template <int S, int MODE>
__global__ __launch_bounds__(256) void k_rows_small(const StepArgs a) {
  __shared__ float Tq[S * S], Tr[S * S];
  for (int i = threadIdx.x; i < S * S; i += 256) {
    Tq[i] = a.qt0 ? a.qt0[i] : 0.0f;
    Tr[i] = a.rate ? a.rate[i] : 0.0f;
  }
  __syncthreads();
  const int64_t R = (int64_t)a.N * a.D;
  const int64_t row0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  int n_changed = 0, n_j1 = 0, n_j2 = 0;
  if (row0 < R) {
    float L[4 * S], RB[MODE == MODE_RATES ? 4 * S : 1];
    int X[4], XB[4], XO[4];
    const bool full = row0 + 3 < R;
    if (full) {
#pragma unroll
      for (int i = 0; i < S; ++i) {
        const float4 v = *(const float4*)(a.logits + (size_t)row0 * S + 4 * i);
        L[4 * i] = v.x; L[4 * i + 1] = v.y; L[4 * i + 2] = v.z; L[4 * i + 3] = v.w;
      }
      const int4 xv4 = *(const int4*)(a.x + row0);
      X[0] = xv4.x; X[1] = xv4.y; X[2] = xv4.z; X[3] = xv4.w;
      if (a.x_base) { const int4 b4 = *(const int4*)(a.x_base + row0); XB[0] = b4.x; XB[1] = b4.y; XB[2] = b4.z; XB[3] = b4.w; }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t rc = row0 + r < R ? row0 + r : R - 1;
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) L[r * S + s_] = a.logits[(size_t)rc * S + s_];
        X[r] = a.x[rc];
        if (a.x_base) XB[r] = a.x_base[rc];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + r;
      const bool live = row < R;
      const int64_t rowc = live ? row : R - 1;
      const int xcur = min(max(X[r], 0), S - 1);
      const int xv = a.x_base ? min(max(XB[r], 0), S - 1) : xcur;
      XO[r] = xcur;
      float rr[S], ratio[S], l[S], e[S];
      float m = -INFINITY;
#pragma unroll
      for (int k = 0; k < S; ++k) { l[k] = L[r * S + k]; m = fmaxf(m, l[k]); }
      float z = 0.0f;
#pragma unroll
      for (int k = 0; k < S; ++k) { e[k] = expf(l[k] - m); z += e[k]; }
      if (a.branch == CTDD_BRANCH_CTELBO) {
        float w[S];
#pragma unroll
        for (int k = 0; k < S; ++k) { w[k] = (e[k] / z) / (Tq[k * S + xv] + a.eps); ratio[k] = 0.0f; }
#pragma unroll
        for (int s0 = 0; s0 < S; ++s0)
#pragma unroll
          for (int k = 0; k < S; ++k) ratio[k] = fmaf(w[s0], Tq[s0 * S + k], ratio[k]);
#pragma unroll
        for (int k = 0; k < S; ++k) rr[k] = (a.beta * Tr[k * S + xv]) * ratio[k];
      } else {
        float ll[S];
        const float logz = logf(z);
        if (a.logit_type == CTDD_LOGIT_DIRECT) {
#pragma unroll
          for (int k = 0; k < S; ++k) ll[k] = l[k] - m - logz;
        } else if (a.logit_type == CTDD_LOGIT_REVERSE_PROB) {
          float acc[S];
#pragma unroll
          for (int k = 0; k < S; ++k) acc[k] = 0.0f;
#pragma unroll
          for (int s0 = 0; s0 < S; ++s0) {
            const float ps = e[s0] / z;
#pragma unroll
            for (int k = 0; k < S; ++k) acc[k] = fmaf(ps, Tq[s0 * S + k], acc[k]);
          }
#pragma unroll
          for (int k = 0; k < S; ++k) ll[k] = logf(acc[k] + 1e-35f);
        } else {
          float mx[S], sm[S];
#pragma unroll
          for (int k = 0; k < S; ++k) { mx[k] = -INFINITY; sm[k] = 0.0f; }
#pragma unroll
          for (int pass = 0; pass < 2; ++pass)
#pragma unroll
            for (int s0 = 0; s0 < S; ++s0) {
              const float lp = l[s0] - m - logz;
#pragma unroll
              for (int k = 0; k < S; ++k) {
                const float q = Tq[s0 * S + k];
                const float t = lp + (q <= 1e-35f ? -1e9f : logf(q));
                if (pass == 0) mx[k] = fmaxf(mx[k], t);
                else sm[k] += expf(t - mx[k]);
              }
            }
#pragma unroll
          for (int k = 0; k < S; ++k) ll[k] = mx[k] + logf(sm[k]);
        }
        float ll_xt = ll[0];
#pragma unroll
        for (int k = 1; k < S; ++k) ll_xt = (k == xv) ? ll[k] : ll_xt;
#pragma unroll
        for (int k = 0; k < S; ++k) {
          ratio[k] = expf(ll[k] - ll_xt);
          rr[k] = ratio[k] * (a.beta * Tr[xv * S + k]);
        }
      }
      if (MODE == MODE_RATES) {                              // (L is dead from here: the rates / ratios of the four rows go out as S 16-byte stores)
#pragma unroll
        for (int k = 0; k < S; ++k) { L[r * S + k] = rr[k]; RB[r * S + k] = ratio[k]; }
        continue;
      }
#pragma unroll
      for (int k = 0; k < S; ++k) {
        if (a.flags & CTDD_STEP_CORRECTOR) rr[k] += a.beta * Tr[xv * S + k];
        if (k == xv) rr[k] = 0.0f;
      }
      const int base = xv;
      if (MODE == MODE_MIDPOINT) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < S; ++k) acc += rr[k] * (float)(k - xv);
        XO[r] = min(max(xv + (int)rintf(a.h * acc), 0), S - 1);
        continue;
      }
      if (MODE == MODE_LBJF) {
        float off = 0.0f;
#pragma unroll
        for (int k = 0; k < S; ++k) off += rr[k];
        const float diag = fmaxf(1.0f - a.h * off, 0.0f);
        float P[S], tot = 0.0f;
#pragma unroll
        for (int k = 0; k < S; ++k) { P[k] = (k == xv) ? diag : rr[k] * a.h; tot += P[k]; }
        float lg[S], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < S; ++k) { lg[k] = logf(P[k] / tot + 1e-35f); mx = fmaxf(mx, lg[k]); }
        float se = 0.0f;
#pragma unroll
        for (int k = 0; k < S; ++k) se += expf(lg[k] - mx);
        const float lse = mx + logf(se);
        float pr[S], mx2 = -INFINITY;
#pragma unroll
        for (int k = 0; k < S; ++k) { lg[k] -= lse; mx2 = fmaxf(mx2, lg[k]); }
        float se2 = 0.0f;
#pragma unroll
        for (int k = 0; k < S; ++k) { pr[k] = expf(lg[k] - mx2); se2 += pr[k]; }
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < S; ++k) {
          pr[k] /= se2;
          if (live && a.out_a) a.out_a[(size_t)row * S + k] = pr[k];
          float Ev;
          if (a.E) Ev = a.E[(size_t)rowc * S + k];
          else Ev = -logf(u01(philox_row(a.seed, a.offset, (uint64_t)rowc, (uint32_t)k).x));
          const float v = pr[k] / Ev;
          if (v > best || (v == best && k < bi)) { best = v; bi = k; }
        }
        XO[r] = bi;
        if (live && bi != xv) ++n_changed;
        continue;
      }
      // ---- MODE_TAULEAP
      float T = 0.0f;
#pragma unroll
      for (int k = 0; k < S; ++k) T += rr[k];
      const float Lam = T * a.h;
      const bool ordinal = a.flags & CTDD_STEP_ORDINAL;
      int jump = 0, njumps = 0;
      if (Lam > 0.0f && Lam <= SUPERPOSE_MAX_LAMBDA) {
        PhiloxStream rng(a.seed, a.offset, (uint64_t)rowc, 0u);
        const int K = poisson_row(Lam, rng);
        njumps = K;
        if (K > 0 && (ordinal || K == 1)) {
          float c[S], carry = 0.0f;
#pragma unroll
          for (int k = 0; k < S; ++k) { c[k] = rr[k] + carry; carry = c[k]; }
          for (int j = 0; j < K; ++j) {
            const float target = rng.next() * T;
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < S; ++k) cnt += (c[k] <= target) ? 1 : 0;
            jump += min(cnt, S - 1) - base;
          }
        }
      } else if (Lam > SUPERPOSE_MAX_LAMBDA) {
        int cnt = 0;
        long long jl = 0;
#pragma unroll
        for (int k = 0; k < S; k += 4) {
          const float r1 = k + 1 < S ? rr[k + 1] : 0.0f, r2 = k + 2 < S ? rr[k + 2] : 0.0f, r3 = k + 3 < S ? rr[k + 3] : 0.0f;
          const int b = k >> 2;
          const u4 blk = philox_row(a.seed, a.offset, (uint64_t)rowc, DENSE_DRAW0 + (uint32_t)(b >> 2));
          const uint32_t w = (b & 3) == 0 ? blk.x : (b & 3) == 1 ? blk.y : (b & 3) == 2 ? blk.z : blk.w;
          cnt += min(subblock_draw(rr[k], r1, r2, r3, a.h, u01(w), a.seed, a.offset, (uint64_t)rowc, b, base, min(4, S - k), &jl), 1 << 20);
        }
        jl = jl > S ? S : (jl < -S ? -S : jl);
        jump = (ordinal || cnt <= 1) ? (int)jl : 0;
        njumps = cnt;
      }
      const int xn = min(max(xcur + jump, 0), S - 1);
      XO[r] = xn;
      if (live) {
        const bool moved = (a.flags & CTDD_STEP_COUNT_RAW) ? (jump != 0) : (xn != xcur);
        n_changed += moved ? 1 : 0;
        n_j1 += njumps > 0 ? 1 : 0;
        n_j2 += njumps > 1 ? 1 : 0;
      }
    }
    if (MODE == MODE_RATES) {
      if (full) {
#pragma unroll
        for (int i = 0; i < S; ++i) {
          *(float4*)(a.out_a + (size_t)row0 * S + 4 * i) = make_float4(L[4 * i], L[4 * i + 1], L[4 * i + 2], L[4 * i + 3]);
          if (a.out_b) *(float4*)(a.out_b + (size_t)row0 * S + 4 * i) = make_float4(RB[4 * i], RB[4 * i + 1], RB[4 * i + 2], RB[4 * i + 3]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int k = 0; k < S; ++k)
            if (row0 + r < R) {
              a.out_a[(size_t)(row0 + r) * S + k] = L[r * S + k];
              if (a.out_b) a.out_b[(size_t)(row0 + r) * S + k] = RB[r * S + k];
            }
      }
    } else {
      if (full) *(int4*)(a.out_x + row0) = make_int4(XO[0], XO[1], XO[2], XO[3]);
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (row0 + r < R) a.out_x[row0 + r] = XO[r];
      }
    }
  }
  if (MODE != MODE_RATES && MODE != MODE_MIDPOINT && a.out_changed) {   // one atomic per wave and counter
    n_changed = grp_sum_i(n_changed, 64);
    if ((threadIdx.x & 63) == 0 && n_changed) atomicAdd(a.out_changed, n_changed);
    if (MODE == MODE_TAULEAP && (a.flags & CTDD_STEP_COUNT_JUMPS)) {
      n_j1 = grp_sum_i(n_j1, 64); n_j2 = grp_sum_i(n_j2, 64);
      if ((threadIdx.x & 63) == 0) { if (n_j1) atomicAdd(a.out_changed + 1, n_j1); if (n_j2) atomicAdd(a.out_changed + 2, n_j2); }
    }
  }
}

template <int S>
static void launch_small_mode(const StepArgs& a, dim3 g, hipStream_t st) {
  switch (a.mode) {
    case MODE_RATES: hipLaunchKernelGGL((k_rows_small<S, MODE_RATES>), g, dim3(256), 0, st, a); break;
    case MODE_TAULEAP: hipLaunchKernelGGL((k_rows_small<S, MODE_TAULEAP>), g, dim3(256), 0, st, a); break;
    case MODE_LBJF: hipLaunchKernelGGL((k_rows_small<S, MODE_LBJF>), g, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((k_rows_small<S, MODE_MIDPOINT>), g, dim3(256), 0, st, a); break;
  }
}
// the fused steps with ONE table for the launch, rates not precomputed, 16-byte aligned buffers
static bool small_rows_ok(const StepArgs& a) {
  if (a.S > 8 || a.tidx || a.pre_rates) return false;
  if (a.mode != MODE_RATES && a.mode != MODE_TAULEAP && a.mode != MODE_LBJF && a.mode != MODE_MIDPOINT) return false;
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15u) == 0; };
  return al16(a.logits) && al16(a.x) && al16(a.x_base) && al16(a.out_x) && (a.mode != MODE_RATES || (al16(a.out_a) && al16(a.out_b)));
}
static int launch_rows_small(const StepArgs& a, void* stream) {
  const int64_t R = (int64_t)a.N * a.D;
  const int64_t grid = (R + 1023) / 1024;
  CTDD_REQUIRE(grid > 0 && grid < (1ll << 31), CTDD_ERANGE, "rows out of range: %lld", (long long)R);
  hipStream_t st = (hipStream_t)stream;
  const dim3 g((unsigned)grid);
  switch (a.S) {
    case 2: launch_small_mode<2>(a, g, st); break;
    case 3: launch_small_mode<3>(a, g, st); break;
    case 4: launch_small_mode<4>(a, g, st); break;
    case 5: launch_small_mode<5>(a, g, st); break;
    case 6: launch_small_mode<6>(a, g, st); break;
    case 7: launch_small_mode<7>(a, g, st); break;
    default: launch_small_mode<8>(a, g, st); break;
  }
  return finish_launch("k_rows_small");
}

static int launch_rows(const StepArgs& a0, void* stream) {
  if (small_rows_ok(a0)) return launch_rows_small(a0, stream);
  StepArgs a = a0;
  int G = 1;
  while (G < a.S && G < 64) G <<= 1;
  if (a.S <= 4) G = 1;          // maze (S = 3), synthetic (S = 2): a row per LANE -- 64 rows share a wave's instruction stream
  a.G = G;
  const int ept_need = (a.S + G - 1) / G;
  const int64_t R = (int64_t)a.N * a.D;
  const int rows_per_wg = 4 * ROWS_PER_WAVE * (64 / G);
  const int64_t grid = (R + rows_per_wg - 1) / rows_per_wg;
  CTDD_REQUIRE(grid > 0 && grid < (1ll << 31), CTDD_ERANGE, "rows out of range: %lld", (long long)R);
  hipStream_t st = (hipStream_t)stream;
  dim3 g((unsigned)grid), b(256);
  if (ept_need <= 1) hipLaunchKernelGGL(k_rows<1>, g, b, 0, st, a);
  else if (ept_need <= 2) hipLaunchKernelGGL(k_rows<2>, g, b, 0, st, a);
  else if (ept_need <= 4) hipLaunchKernelGGL(k_rows<4>, g, b, 0, st, a);
  else if (ept_need <= 8) hipLaunchKernelGGL(k_rows<8>, g, b, 0, st, a);
  else hipLaunchKernelGGL(k_rows<16>, g, b, 0, st, a);
  return finish_launch("k_rows");
}

static int check_common(const void* logits, const void* x, int N, int D, int S) {
  CTDD_REQUIRE(logits && x, CTDD_EINVAL, "null logits/x");
  CTDD_REQUIRE(N > 0 && D > 0, CTDD_EINVAL, "N=%d D=%d must be positive", N, D);
  CTDD_REQUIRE(S >= 2 && S <= CTDD_MAX_S, CTDD_ERANGE, "S=%d outside [2,%d]", S, CTDD_MAX_S);
  return CTDD_OK;
}

static int check_branch(int branch, int logit_type, const void* qt0, const void* rate) {
  CTDD_REQUIRE(branch == CTDD_BRANCH_CTELBO || branch == CTDD_BRANCH_CRM, CTDD_EINVAL, "unknown branch %d", branch);
  CTDD_REQUIRE(logit_type >= 0 && logit_type <= 2, CTDD_EINVAL, "unknown logit_type %d", logit_type);
  CTDD_REQUIRE(rate, CTDD_EINVAL, "null rate table");
  CTDD_REQUIRE(qt0 || (branch == CTDD_BRANCH_CRM && logit_type == CTDD_LOGIT_DIRECT), CTDD_EINVAL,
               "qt0 table required for this branch/logit_type");
  return CTDD_OK;
}

}  // namespace ctdd

using namespace ctdd;

// steps_s256.hip: fast path, returns 1 when it handled the call
namespace ctdd { int try_s256(const StepArgs& a, void* stream, int* status); }

extern "C" int ctdd_logprob(const float* logits, const int32_t* x, const float* qt0, const int32_t* tidx,
                            int logit_type, int N, int D, int S, float* out_ll_all, float* out_ll_xt,
                            void* stream) {
  if (int rc = check_common(logits, x, N, D, S)) return rc;
  CTDD_REQUIRE(out_ll_all && out_ll_xt, CTDD_EINVAL, "null output");
  CTDD_REQUIRE(logit_type >= 0 && logit_type <= 2, CTDD_EINVAL, "unknown logit_type %d", logit_type);
  CTDD_REQUIRE(qt0 || logit_type == CTDD_LOGIT_DIRECT, CTDD_EINVAL, "qt0 required");
  StepArgs a{};
  a.logits = logits; a.x = x; a.qt0 = qt0; a.tidx = tidx; a.N = N; a.D = D; a.S = S;
  a.branch = CTDD_BRANCH_CRM; a.logit_type = logit_type; a.mode = MODE_LOGPROB;
  a.out_a = out_ll_all; a.out_b = out_ll_xt;
  return launch_rows(a, stream);
}

extern "C" int ctdd_reverse_rates(int branch, int logit_type, const float* logits, const int32_t* x,
                                  const float* qt0, const float* rate, const int32_t* tidx, float eps,
                                  int N, int D, int S, float* out_rates, float* out_ratio, void* stream) {
  if (int rc = check_common(logits, x, N, D, S)) return rc;
  if (int rc = check_branch(branch, logit_type, qt0, rate)) return rc;
  CTDD_REQUIRE(out_rates, CTDD_EINVAL, "null output");
  StepArgs a{};
  a.logits = logits; a.x = x; a.qt0 = qt0; a.rate = rate; a.tidx = tidx; a.beta = 1.0f; a.eps = eps;
  a.N = N; a.D = D; a.S = S; a.branch = branch; a.logit_type = logit_type; a.mode = MODE_RATES;
  a.out_a = out_rates; a.out_b = out_ratio;
  int st;
  if (try_s256(a, stream, &st)) return st;
  return launch_rows(a, stream);
}

extern "C" int ctdd_tauleap_draw(const float* rates, const int32_t* x, const int32_t* x_base, float h,
                                 uint32_t flags, uint64_t seed, uint64_t offset, int N, int D, int S,
                                 int32_t* out_x, int32_t* out_changed, void* stream) {
  if (int rc = check_common(rates, x, N, D, S)) return rc;
  CTDD_REQUIRE(out_x, CTDD_EINVAL, "null output");
  StepArgs a{};
  a.logits = rates; a.x = x; a.x_base = x_base; a.h = h; a.flags = flags & CTDD_STEP_ORDINAL;
  a.seed = seed; a.offset = offset; a.N = N; a.D = D; a.S = S; a.mode = MODE_DRAW_ONLY;
  a.out_x = out_x; a.out_changed = out_changed;
  return launch_rows(a, stream);
}

extern "C" int ctdd_tauleap_step(int branch, int logit_type, const float* logits, const int32_t* x,
                                 const int32_t* x_base, const float* qt0, const float* base_rate,
                                 float beta, float eps, float h, uint32_t flags, uint64_t seed,
                                 uint64_t offset, int N, int D, int S, int32_t* out_x,
                                 int32_t* out_changed, void* stream) {
  if (int rc = check_common(logits, x, N, D, S)) return rc;
  if (int rc = check_branch(branch, logit_type, qt0, base_rate)) return rc;
  CTDD_REQUIRE(out_x, CTDD_EINVAL, "null output");
  StepArgs a{};
  a.logits = logits; a.x = x; a.x_base = x_base; a.qt0 = qt0; a.rate = base_rate; a.beta = beta;
  a.eps = eps; a.h = h; a.flags = flags; a.seed = seed; a.offset = offset; a.N = N; a.D = D; a.S = S;
  a.branch = branch; a.logit_type = logit_type; a.mode = MODE_TAULEAP;
  a.out_x = out_x; a.out_changed = out_changed;
  int st;
  if (try_s256(a, stream, &st)) return st;
  return launch_rows(a, stream);
}

extern "C" int ctdd_lbjf_step(int branch, int logit_type, const float* logits, const int32_t* x,
                              const float* qt0, const float* base_rate, float beta, float eps, float h,
                              uint32_t flags, const float* E, uint64_t seed, uint64_t offset,
                              int N, int D, int S, int32_t* out_x, float* out_probs,
                              int32_t* out_changed, void* stream) {
  if (int rc = check_common(logits, x, N, D, S)) return rc;
  if (int rc = check_branch(branch, logit_type, qt0, base_rate)) return rc;
  CTDD_REQUIRE(out_x, CTDD_EINVAL, "null output");
  StepArgs a{};
  a.logits = logits; a.x = x; a.qt0 = qt0; a.rate = base_rate; a.beta = beta; a.eps = eps; a.h = h;
  a.flags = flags; a.E = E; a.seed = seed; a.offset = offset; a.N = N; a.D = D; a.S = S;
  a.branch = branch; a.logit_type = logit_type; a.mode = MODE_LBJF;
  a.out_x = out_x; a.out_a = out_probs; a.out_changed = out_changed;
  return launch_rows(a, stream);
}

extern "C" int ctdd_midpoint_predict(int branch, int logit_type, const float* logits, const int32_t* x,
                                     const float* qt0, const float* base_rate, float beta, float eps,
                                     float h, int N, int D, int S, int32_t* out_x, void* stream) {
  if (int rc = check_common(logits, x, N, D, S)) return rc;
  if (int rc = check_branch(branch, logit_type, qt0, base_rate)) return rc;
  CTDD_REQUIRE(out_x, CTDD_EINVAL, "null output");
  StepArgs a{};
  a.logits = logits; a.x = x; a.qt0 = qt0; a.rate = base_rate; a.beta = beta; a.eps = eps;
  a.h = (float)(0.5 * (double)h);   // 0.5*h evaluated in double, then cast (sampling.py:437-439)
  a.N = N; a.D = D; a.S = S; a.branch = branch; a.logit_type = logit_type; a.mode = MODE_MIDPOINT;
  a.out_x = out_x;
  return launch_rows(a, stream);
}

extern "C" int ctdd_exact_step(const float* logits, const int32_t* x, const float* q_lo, const float* q_step, const float* E,
                               uint64_t seed, uint64_t offset, int N, int D, int S, int32_t* out_x, float* out_probs,
                               int32_t* out_changed, void* stream) {
  if (int rc = check_common(logits, x, N, D, S)) return rc;
  CTDD_REQUIRE(q_lo && q_step && out_x, CTDD_EINVAL, "exact step: null table / output");
  StepArgs a{};
  a.logits = logits; a.x = x; a.qt0 = q_lo; a.rate = q_step; a.E = E; a.seed = seed; a.offset = offset; a.N = N; a.D = D; a.S = S;
  a.branch = CTDD_BRANCH_CRM; a.logit_type = CTDD_LOGIT_DIRECT; a.mode = MODE_EXACT;
  a.out_x = out_x; a.out_a = out_probs; a.out_changed = out_changed;
  return launch_rows(a, stream);
}

/* LBJF / midpoint tails on reverse rates that are already computed and masked (rates (N,D,S): the out_rates of
 * ctdd_tauleap_step_s256) -- the S = 256 samplers get their S x S contraction from the matrix-core kernel this way. */
extern "C" int ctdd_lbjf_from_rates(const float* rates, const int32_t* x, float h, const float* E, uint64_t seed, uint64_t offset,
                                    int N, int D, int S, int32_t* out_x, float* out_probs, int32_t* out_changed, void* stream) {
  if (int rc = check_common(rates, x, N, D, S)) return rc;
  CTDD_REQUIRE(out_x, CTDD_EINVAL, "null output");
  StepArgs a{};
  a.logits = rates; a.x = x; a.h = h; a.E = E; a.seed = seed; a.offset = offset; a.N = N; a.D = D; a.S = S;
  a.mode = MODE_LBJF; a.pre_rates = 1;
  a.out_x = out_x; a.out_a = out_probs; a.out_changed = out_changed;
  return launch_rows(a, stream);
}

extern "C" int ctdd_midpoint_from_rates(const float* rates, const int32_t* x, float h, int N, int D, int S, int32_t* out_x,
                                        void* stream) {
  if (int rc = check_common(rates, x, N, D, S)) return rc;
  CTDD_REQUIRE(out_x, CTDD_EINVAL, "null output");
  StepArgs a{};
  a.logits = rates; a.x = x; a.h = (float)(0.5 * (double)h); a.N = N; a.D = D; a.S = S;
  a.mode = MODE_MIDPOINT; a.pre_rates = 1;
  a.out_x = out_x;
  return launch_rows(a, stream);
}
