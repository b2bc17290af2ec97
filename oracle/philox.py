"""Oracle (test infrastructure): Philox4x32-10 and the draw rules of the HIP sampling kernels.

Philox4x32-10 follows the published algorithm (Salmon et al., "Parallel random numbers: as easy
as 1, 2, 3", SC'11; constants M0=0xD2511F53, M1=0xCD9E8D57, W0=0x9E3779B9, W1=0xBB67AE85).  The
reference has no counter-based RNG (it uses the global torch generator); this file
restates the *build's* device RNG so GPU draws can be replayed bit-exactly on the CPU.

Counter layout used by every kernel (csrc/philox.hpp):
    counter = (row_lo, row_hi, offset, draw)   key = (seed_lo, seed_hi)
where row = flat row index (n*D+d, or b for per-batch draws), offset = host-supplied call index
(sampler step / corrector sub-step), draw = 0,1,2,... successive 4-word blocks for that row.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """All args broadcastable uint32 arrays.  Returns 4 uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint32).astype(np.uint64) for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0
            p1 = M1 * c2
            hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
            hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
            n0 = hi1 ^ c1 ^ np.uint64(k0)
            n1 = lo1
            n2 = hi0 ^ c3 ^ np.uint64(k1)
            n3 = lo0
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def u01(r):
    """uint32 -> float32 in the open interval (0,1): ((r>>9) + 0.5) * 2^-23.
    Every step is exact in fp32 (24 significant bits); min 2^-24, max 1-2^-24."""
    return (((r >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0**-23)).astype(np.float32)


def exp1(r):
    """uint32 -> Exp(1) float32: -log(u01(r))."""
    return (-np.log(u01(r).astype(np.float32))).astype(np.float32)


def row_uniforms(rows, offset, seed, ndraw_blocks=1):
    """uniforms[row, 4*ndraw_blocks] for flat row ids `rows` (uint64-able)."""
    rows = np.asarray(rows, dtype=np.uint64)
    lo = (rows & MASK32).astype(np.uint32)
    hi = (rows >> np.uint64(32)).astype(np.uint32)
    k0, k1 = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    out = []
    for j in range(ndraw_blocks):
        r = philox4x32_10(lo, hi, np.uint32(offset), np.uint32(j), k0, k1)
        out.extend(u01(x) for x in r)
    return np.stack(out, axis=-1)


# ------------------------------------------------------------------ draw rules of the kernels
POISSON_ICDF_MAX_LAMBDA = 12.0   # rows with total rate*h above this take the per-element path
POISSON_ICDF_KMAX = 64


def poisson_icdf(lam, u):
    """Sequential-search inverse CDF, float32 arithmetic, same operation order as the kernel:
    k=0; p=exp(-lam); c=p; while (u > c && k < KMAX) { k++; p = p*lam/k; c += p; }"""
    lam = np.asarray(lam, dtype=np.float32)
    u = np.asarray(u, dtype=np.float32)
    k = np.zeros(lam.shape, dtype=np.int32)
    p = np.exp(-lam).astype(np.float32)
    c = p.copy()
    margin = np.abs(u - c)                       # distance to the nearest tested boundary
    active = u > c
    it = 0
    while active.any() and it < POISSON_ICDF_KMAX:
        it += 1
        k = np.where(active, k + 1, k)
        p = np.where(active, (p * lam / k.clip(1).astype(np.float32)).astype(np.float32), p)
        c = np.where(active, (c + p).astype(np.float32), c)
        margin = np.where(active, np.minimum(margin, np.abs(u - c)), margin)
        active = active & (u > c)
    return k, margin


def categorical_icdf(w, u):
    """First index s with cumsum(w)[s] > u*sum(w) (w >= 0, float32, row-wise).
    Returns (index, margin) where margin is the relative distance of the target to the nearest
    cumulative boundary -- tests skip rows whose margin is below fp32 reassociation noise."""
    w = np.asarray(w, dtype=np.float64)
    cs = np.cumsum(w, axis=-1)
    tot = cs[..., -1:]
    target = np.asarray(u, dtype=np.float64)[..., None] * tot
    idx = np.sum(cs <= target, axis=-1)
    idx = np.minimum(idx, w.shape[-1] - 1)
    margin = np.min(np.abs(cs - target), axis=-1) / np.maximum(tot[..., 0], 1e-300)
    return idx.astype(np.int64), margin


def tauleap_draw_replay(rates, x, h, is_ordinal, seed, offset, x_base=None):
    """CPU replay of the kernels' jump draw (csrc/steps_generic.hip, MODE_TAULEAP):
    rates (N,D,S) float32 reverse rates evaluated at x_base (default x); its own state is masked here.
    K ~ Poisson(h*sum_s r_s) from uniform #0 of the row's stream; K destinations by inverse CDF
    over r in s order from uniforms #1.. ; non-ordinal rows with K>1 stay put.
    Rows with h*sum > POISSON_ICDF_MAX_LAMBDA use the per-element path and are flagged undecided.
    Returns (x_new int64 (N,D), decided bool (N,D)) -- `decided` is False where a float
    comparison sits within reassociation noise of a boundary (or on the dense path)."""
    r = np.array(rates, dtype=np.float32, copy=True)
    N, D, S = r.shape
    x = np.asarray(x).astype(np.int64)
    base = x if x_base is None else np.asarray(x_base).astype(np.int64)
    np.put_along_axis(r, base[..., None], 0.0, axis=-1)      # own state of the rate-state (x' if given)
    r = r.reshape(N * D, S)
    T = r.astype(np.float64).sum(-1)
    lam = (T * np.float64(np.float32(h))).astype(np.float32)
    rows = np.arange(N * D, dtype=np.uint64)
    nblk = 1 + (POISSON_ICDF_KMAX + 4) // 4
    U = row_uniforms(rows, offset, seed, nblk)                     # (R, 4*nblk)
    K, margin = poisson_icdf(lam, U[:, 0])
    K = np.where(lam > 0, K, 0)
    decided = (margin > 1e-5 * np.maximum(1.0, lam)) | (lam == 0)
    dense = lam > POISSON_ICDF_MAX_LAMBDA
    decided &= ~dense
    xf, bf = x.reshape(-1), base.reshape(-1)
    jump = np.zeros(N * D, dtype=np.int64)
    for j in range(int(K[~dense].max()) if (~dense).any() else 0):
        act = (K > j) & ~dense
        if not act.any():
            break
        idx, mg = categorical_icdf(r[act], U[act, 1 + j])
        jump[act] += idx - bf[act]
        d = decided[act]
        d &= mg > 1e-5
        decided[act] = d
    if not is_ordinal:
        jump = np.where(K > 1, 0, jump)
    xn = np.clip(xf + jump, 0, S - 1)
    return xn.reshape(N, D), decided.reshape(N, D)
