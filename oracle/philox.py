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


def row_uniforms(rows, offset, seed, ndraw_blocks=1, draw0=0):
    """uniforms[row, 4*ndraw_blocks] for flat row ids `rows` (uint64-able), blocks draw0.."""
    rows = np.asarray(rows, dtype=np.uint64)
    lo = (rows & MASK32).astype(np.uint32)
    hi = (rows >> np.uint64(32)).astype(np.uint32)
    k0, k1 = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    out = []
    for j in range(ndraw_blocks):
        r = philox4x32_10(lo, hi, np.uint32(offset), np.uint32(draw0 + j), k0, k1)
        out.extend(u01(x) for x in r)
    return np.stack(out, axis=-1)


# ------------------------------------------------------------------ draw rules of the kernels
POISSON_ICDF_MAX_LAMBDA = 12.0   # inverse-CDF search up to here, PTRS above
POISSON_ICDF_KMAX = 64
SUPERPOSE_MAX_LAMBDA = 64.0      # rows with total rate*h above this draw per sub-block of 4 destinations
DENSE_DRAW0 = 1024
SPLIT_DRAW0 = 8192
PICK_DRAW0 = 65536


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


def poisson_ptrs(lam, uu):
    """Hoermann's PTRS for Poisson(lam), lam > 12, exactly as csrc/draw.hpp:poisson_ptrs (fp64 arithmetic on the float32
    uniforms taken from the iterator `uu`).  Returns (k, decided)."""
    import math
    lam = float(np.float32(lam))
    slam, loglam = math.sqrt(lam), math.log(lam)
    b = 0.931 + 2.53 * slam
    a = -0.059 + 0.02483 * b
    invalpha = 1.1239 + 1.1328 / (b - 3.4)
    vr = 0.9277 - 3.6224 / (b - 2.0)
    ok = True
    for _ in range(32):
        U = float(next(uu)) - 0.5
        V = float(next(uu))
        us = 0.5 - abs(U)
        kf = math.floor((2.0 * a / us + b) * U + lam + 0.43)
        ok &= abs(us - 0.07) > 1e-9 and abs(V - vr) > 1e-9
        if us >= 0.07 and V <= vr:
            return int(min(kf, 1.0e9)), ok
        ok &= abs(us - 0.013) > 1e-9 and abs(V - us) > 1e-9
        if kf < 0.0 or (us < 0.013 and V > us):
            continue
        lhs = math.log(V) + math.log(invalpha) - math.log(a / (us * us) + b)
        rhs = -lam + kf * loglam - math.lgamma(kf + 1.0)
        ok &= abs(lhs - rhs) > 1e-7 * max(1.0, abs(rhs))
        if lhs <= rhs:
            return int(min(kf, 1.0e9)), ok
    return int(min(round(lam), 1.0e9)), ok


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
    """CPU replay of the kernels' jump draw (csrc/draw.hpp + steps_generic.hip / steps_s256.hip).
    rates (N,D,S) float32 reverse rates evaluated at x_base (default x); its own state is masked here.

    Row rule: Lambda = h*sum_s r_s.
      Lambda <= SUPERPOSE_MAX_LAMBDA: K ~ Poisson(Lambda) from the first uniform(s) of the row's stream
        (Lambda > 12: n = ceil(Lambda/12) draws of Poisson(Lambda/n), summed), then K destinations by
        inverse CDF over r in s order from the following uniforms;
      above: the same rule one level down: every sub-block b of 4 consecutive destinations draws
        K_b ~ Poisson(h*sum_{s in b} r_s) from uniform (b&3) of Philox block DENSE_DRAW0 + (b>>2) and
        K_b picks among its destinations from the stream PICK_DRAW0 + 16 b (a rate > 12 is the sum
        of <= 64 equal parts from the stream SPLIT_DRAW0 + 16 b).
    Non-ordinal rows with more than one jump stay put.
    Returns (x_new int64 (N,D), decided bool (N,D)) -- `decided` is False where a float
    comparison sits within reassociation noise of a boundary."""
    r = np.array(rates, dtype=np.float32, copy=True)
    N, D, S = r.shape
    x = np.asarray(x).astype(np.int64)
    base = x if x_base is None else np.asarray(x_base).astype(np.int64)
    np.put_along_axis(r, base[..., None], 0.0, axis=-1)      # own state of the rate-state (x' if given)
    r = r.reshape(N * D, S)
    h32 = np.float32(h)
    T = r.astype(np.float64).sum(-1)
    lam = (T * np.float64(h32)).astype(np.float32)
    rows = np.arange(N * D, dtype=np.uint64)
    xf, bf = x.reshape(-1), base.reshape(-1)
    jump = np.zeros(N * D, dtype=np.int64)
    decided = np.ones(N * D, dtype=bool)
    decided &= np.abs(lam - SUPERPOSE_MAX_LAMBDA) > 1e-4       # regime choice itself must be clear
    dense = lam > SUPERPOSE_MAX_LAMBDA
    # ---- superposition regime: K from the first uniform(s) of the row stream, picks from the next ones
    sp = ~dense & (lam > 0)
    if sp.any():
        lam_sp = lam[sp]
        npart = np.where(lam_sp <= POISSON_ICDF_MAX_LAMBDA, 1, np.ceil(lam_sp / np.float32(POISSON_ICDF_MAX_LAMBDA))).astype(np.int64)
        lc = (lam_sp / npart.astype(np.float32)).astype(np.float32)
        nblk = 2 + (int(npart.max()) + 6 * POISSON_ICDF_KMAX) // 4
        U = row_uniforms(rows[sp], offset, seed, min(nblk, 64))
        K = np.zeros(lam_sp.shape, dtype=np.int64)
        dsp = np.ones(lam_sp.shape, dtype=bool)
        for i in range(int(npart.max())):
            act = npart > i
            k_i, mg = poisson_icdf(lc[act], U[act, i])
            K[act] += k_i
            d = dsp[act]
            d &= mg > 1e-5
            dsp[act] = d
        jsp = np.zeros(sp.sum(), dtype=np.int64)
        rs, bs = r[sp], bf[sp]
        ridx = np.arange(lam_sp.shape[0])
        for j in range(int(K.max()) if K.size else 0):
            act = K > j
            col = npart[act] + j
            ok = col < U.shape[1]
            idx, mg = categorical_icdf(rs[act], U[ridx[act], np.minimum(col, U.shape[1] - 1)])
            jsp[act] += idx - bs[act]
            d = dsp[act]
            d &= (mg > 1e-5) & ok
            dsp[act] = d
        if not is_ordinal:
            jsp = np.where(K > 1, 0, jsp)
        jump[sp] = jsp
        tmp = decided[sp]
        tmp &= dsp
        decided[sp] = tmp
    # ---- dense regime: sub-blocks of 4 consecutive destinations (csrc/draw.hpp: subblock_draw)
    if dense.any():
        rd, rowsd, bd = r[dense], rows[dense], bf[dense]
        nd = rd.shape[0]
        nb = (S + 3) // 4
        rp = np.zeros((nd, nb * 4), dtype=np.float32)
        rp[:, :S] = rd
        rp = rp.reshape(nd, nb, 4)
        tot = ((rp[..., 0] + rp[..., 1]) + (rp[..., 2] + rp[..., 3])).astype(np.float32)
        lam_b = (h32 * tot).astype(np.float32)
        lo = (rowsd & MASK32).astype(np.uint32)
        hi = (rowsd >> np.uint64(32)).astype(np.uint32)
        k0, k1 = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
        U = np.empty((nd, nb), dtype=np.float32)
        for blk in range((nb + 3) // 4):
            w = philox4x32_10(lo, hi, np.uint32(offset), np.uint32(DENSE_DRAW0 + blk), k0, k1)
            for c in range(4):
                if 4 * blk + c < nb:
                    U[:, 4 * blk + c] = u01(w[c])
        Kb, margin = poisson_icdf(np.minimum(lam_b, np.float32(POISSON_ICDF_MAX_LAMBDA)), U)
        Kb = np.where(lam_b > 0, Kb, 0)
        margin = np.where(lam_b > 0, margin, 1.0)
        dd = np.ones(nd, dtype=bool)
        jl = np.zeros(nd, dtype=np.int64)
        # heavy sub-blocks (rate > 12): one independent Poisson(h r_s) per destination from the sub-block's SPLIT stream
        # (inverse CDF up to 12, PTRS above) -- csrc/draw.hpp: subblock_draw / poisson_ptrs
        heavy = lam_b > np.float32(POISSON_ICDF_MAX_LAMBDA)
        heavy_move = np.zeros(nd, dtype=np.int64)
        for ri, b in zip(*np.nonzero(heavy)):
            uu = iter(row_uniforms(np.array([rowsd[ri]]), offset, seed, 16, draw0=SPLIT_DRAW0 + 16 * int(b))[0])
            ktot = 0
            for i in range(min(4, S - 4 * int(b))):
                li = np.float32(h32 * rp[ri, b, i])
                if not li > 0:
                    continue
                if li <= POISSON_ICDF_MAX_LAMBDA:
                    kk, mg = poisson_icdf(np.array([li], dtype=np.float32), np.array([next(uu)], dtype=np.float32))
                    k, ok = int(kk[0]), bool(mg[0] > 1e-6)
                else:
                    k, ok = poisson_ptrs(li, uu)
                if not ok:
                    dd[ri] = False
                ktot += k
                heavy_move[ri] += k * (4 * int(b) + i - int(bd[ri]))
            Kb[ri, b] = min(ktot, 1 << 30)
            margin[ri, b] = 1.0
        dd &= margin.min(-1) > 1e-6
        # picks inside the active light sub-blocks
        for ri, b in zip(*np.nonzero((Kb > 0) & ~heavy)):
            K = int(min(Kb[ri, b], 4096))
            nblk = (K + 3) // 4
            uu = row_uniforms(np.array([rowsd[ri]]), offset, seed, nblk, draw0=PICK_DRAW0 + 16 * int(b))[0, :K]
            r4 = rp[ri, b]
            c0, c1, c2 = r4[0], np.float32(r4[0] + r4[1]), np.float32(np.float32(r4[0] + r4[1]) + r4[2])
            v = (uu * tot[ri, b]).astype(np.float32)
            i = (v >= c0).astype(int) + (v >= c1).astype(int) + (v >= c2).astype(int)
            i = np.minimum(i, min(4, S - 4 * b) - 1)
            gap = np.min(np.abs(np.stack([v - c0, v - c1, v - c2])), axis=0) / max(float(tot[ri, b]), 1e-30)
            if (gap < 1e-5).any():
                dd[ri] = False
            jl[ri] += int(np.sum(4 * b + i - bd[ri]))
        cnt = Kb.astype(np.int64).sum(-1)
        jl = np.clip(jl + heavy_move, -S, S)                     # |jump| >= S - 1 saturates the state clamp either way
        jump[dense] = jl if is_ordinal else np.where(cnt <= 1, jl, 0)
        tmp = decided[dense]
        tmp &= dd
        decided[dense] = tmp
    xn = np.clip(xf + jump, 0, S - 1)
    return xn.reshape(N, D), decided.reshape(N, D)
