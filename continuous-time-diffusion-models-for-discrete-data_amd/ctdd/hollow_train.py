"""Training path of the SDDM hollow transformer on hand-written HIP kernels (reference: forward + `l.backward()` through
TAUnSDDM/lib/networks/hollow_networks.py:668-755 and the blocks it is built from).

`HollowTrainer(model)(x, t)` computes the logits of a `BidirectionalTransformer2` with every operation a libctdd launch and
autograd as the tape only: each node is a `torch.autograd.Function` whose forward and backward call the C ABI
(include/ctdd_hollow.h, ctdd_hollow_train.h, ctdd_unet.h, ctdd_unet_train.h) -- one Function per prenorm attention block and
per feed-forward block (`AttnBlockFn`, `MlpBlockFn`: hand-scheduled backward), per-op Functions for the once-per-network readout:

  linear layers      forward / data gradient (the same GEMM with the packed transposed weight): `ctdd_gemm_bf16` (bf16 operands),
                     `ctdd_unet_conv` (exact fp32);  weight gradient `ctdd_unet_wgrad` (kind 1x1: tokens are the contraction index);
                     bias gradient on the same launch (`ctdd_wgrad_args.gb`);  dropout / ReLU-backward masks in the GEMM epilogue
                     (`ctdd_gemm_args.drop_p`, `mask_u`);  operands of all weights by one
                     `ctdd_unet_pack_weights` launch per forward
  LayerNorm (+FiLM)  `ctdd_hollow_layernorm` / `ctdd_hollow_layernorm_bwd` (adds the residual stream's gradient)
  attention          `ctdd_hollow_attention_train_bf16` / `_bwd_bf16` (matrix cores; dropout on the probabilities inside the
                     kernels), `ctdd_hollow_attention_train` / `_bwd` (fp32 FMA: the parity mode, other head dimensions)
  ReLU / GELU, dropout (+ residual, + bf16 copy), embedding, l2r + r2l
                     `ctdd_hollow_relu_bf16`, `ctdd_hollow_act`, `ctdd_hollow_dropout`, `ctdd_hollow_embed(_bwd)`, `ctdd_hollow_add`

Parameters enter the Functions directly, so their gradients are ordinary autograd gradients (DistributedDataParallel hooks
work unchanged).  Dropout masks are Philox(seed, step, layer, element): the backward of a step regenerates the forward's.
precision "fp32": exact-fp32 matrix instructions everywhere (parity mode); "bf16": bf16 GEMM / attention operands, fp32
accumulation, fp32 softmax / LayerNorm / residual streams, the (rows, mlp_dim) hidden tensor in bf16 only.
"""
import ctypes as C
import math

import torch

from . import native
from .unet_engine import SEG_1x1, _ConvArgs, _unwrap
from .hollow_engine import _AttnArgs as _InfAttnArgs, _EmbedArgs, _GemmArgs, _LnArgs, _lib as _hollow_lib, supports  # noqa: F401
from . import unet_train

_P, _I, _F, _I64, _U64 = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_uint64


class _LnBwdArgs(C.Structure):
    _fields_ = [("x", _P), ("y", _P), ("x_bs", _I64), ("y_bs", _I64), ("gamma", _P), ("beta", _P), ("eps", _F), ("film", _P),
                ("film_stride", _I), ("dout", _P), ("dout_bs", _I64), ("B", _I), ("T", _I), ("E", _I), ("rpw", _I), ("dx", _P),
                ("dx_bs", _I64), ("acc_dx", _I), ("dy", _P), ("dy_bs", _I64), ("acc_dy", _I), ("dgamma", _P), ("dbeta", _P), ("dfilm", _P),
                ("nrep", _I), ("rep_stride", _I), ("dres", _P), ("dres_bs", _I64)]


class _AttnTrainArgs(C.Structure):
    _fields_ = [("q", _P), ("k", _P), ("v", _P), ("q_bs", _I64), ("k_bs", _I64), ("v_bs", _I64), ("q_rs", _I), ("k_rs", _I), ("v_rs", _I),
                ("B", _I), ("Tq", _I), ("Tk", _I), ("H", _I), ("hd", _I), ("mode", _I), ("scale", _F), ("out", _P), ("out_rs", _I),
                ("stats", _P), ("drop_p", _F), ("rng", _P), ("layer", _U64), ("d_out", _P), ("dq", _P), ("dk", _P), ("dv", _P),
                ("dq_bs", _I64), ("dk_bs", _I64), ("dv_bs", _I64), ("dq_rs", _I), ("dk_rs", _I), ("dv_rs", _I),
                ("out_bf16", _P), ("dq_bf16", _P), ("dk_bf16", _P), ("dv_bf16", _P)]


class _EmbedBwdArgs(C.Structure):
    _fields_ = [("x64", _P), ("x32", _P), ("dl2r", _P), ("dr2l", _P), ("B", _I), ("D", _I), ("E", _I), ("S", _I), ("dw", _P), ("db", _P)]


_sigs_done = False
USE_GEMM_KERNEL = True          # bf16 linears on ctdd_gemm_bf16 (K % 64 == 0, N % 8 == 0); False: the U-Net slab kernel run as a GEMM


def lib():
    global _sigs_done
    l = _hollow_lib()
    unet_train.lib()
    if not _sigs_done:
        for name, argt in (("ctdd_hollow_layernorm_bwd", [_P, _P]), ("ctdd_hollow_attention_train", [_P, _P]),
                           ("ctdd_hollow_attention_bwd", [_P, _P]), ("ctdd_hollow_attention_train_bf16", [_P, _P]),
                           ("ctdd_hollow_attention_bwd_bf16", [_P, _P]),
                           ("ctdd_hollow_act", [_P, _P, _P, _P, _I64, _I, _F, _P, _U64, _P]), ("ctdd_hollow_embed_bwd", [_P, _P]),
                           ("ctdd_hollow_relu_bf16", [_P, _P, _P, _I64, _F, _P, _U64, _P]),
                           ("ctdd_hollow_colsum", [_P, _P, _I64, _I, _I, _P, _I, _I, _P]),
                           ("ctdd_hollow_dropout", [_P, _P, _P, _P, _I64, _F, _P, _U64, _P])):
            fn = getattr(l, name)
            fn.argtypes, fn.restype = argt, _I
        _sigs_done = True
    return l


def _st():
    return torch.cuda.current_stream().cuda_stream


def _ck(rc, what):
    if rc != 0:
        raise native.CtddError(f"{what} failed ({rc}): {native.load().ctdd_last_error().decode()}")


def _p(t):
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------- launch helpers (no autograd)
def _gemm_fusable(K, N, bf16):
    """The plain GEMM kernel takes this product (its epilogue can then carry a dropout or a ReLU-backward mask)."""
    return bool(bf16 and USE_GEMM_KERNEL and K % 64 == 0 and N % 8 == 0)


def _gemm(x, w, bias, res, rows, K, N, bf16, act=0, want_hi=False, want_f32=True, drop=None, mask_u=None):
    """out[rows][N] = act(x[rows][K] @ w[N][K]^T (+ bias)) (+ res) -> (fp32 out or None, bf16 copy or None).
    x, w: fp32, or bf16 when `bf16`.  drop = (p, rng, layer): dropout of act(.) before the residual, in the epilogue (the masks of
    _dropout / _act); mask_u (+ drop = (p, None, 0)): the ReLU + dropout backward against the saved output -- both only where
    _gemm_fusable says the plain GEMM kernel runs."""
    l = lib()
    dev = x.device
    out = torch.empty((rows, N), dtype=torch.float32, device=dev) if want_f32 else None
    out_hi = torch.empty((rows, N), dtype=torch.bfloat16, device=dev) if want_hi else None
    if _gemm_fusable(K, N, bf16):
        g = _GemmArgs()
        g.a[0], g.nseg, g.w, g.bias, g.res = x.data_ptr(), 1, w.data_ptr(), _p(bias), _p(res)
        g.out_f32, g.out_hi, g.M, g.N, g.K, g.act = _p(out), _p(out_hi), rows, N, K, act
        if drop is not None:
            g.drop_p, g.rng, g.layer = float(drop[0]), _p(drop[1]), int(drop[2])
        g.mask_u = _p(mask_u)
        _ck(l.ctdd_gemm_bf16(C.byref(g), _st()), "ctdd_gemm_bf16")
        return out, out_hi
    if drop is not None or mask_u is not None:
        raise native.CtddError("_gemm: dropout / mask epilogues exist in the plain GEMM kernel only")
    a = _ConvArgs()
    a.nseg = 1
    a.seg[0].C, a.seg[0].kind = K, SEG_1x1
    if bf16:
        a.seg[0].hi, a.w_hi = x.data_ptr(), w.data_ptr()
    else:
        a.seg[0].f32, a.w_f32 = x.data_ptr(), w.data_ptr()
    a.B, a.H, a.W, a.Hin, a.Win, a.N, a.Ktot = 1, rows, 1, rows, 1, N, K
    a.bias, a.res_f32, a.out_f32, a.out_hi, a.act = _p(bias), _p(res), _p(out), _p(out_hi), act
    if bf16 and N % 8 == 0:
        pbk = 64 if K % 64 == 0 else 48 if K % 48 == 0 else 32 if K % 32 == 0 else 16
        if pbk == 64:
            pbnt = 4 if N > 64 else 2 if N > 32 else 1
        elif pbk == 48:
            pbnt = 4 if N % 128 == 0 else 3 if N > 64 else 2 if N > 32 else 1
        elif pbk == 32:
            pbnt = 4 if N % 128 == 0 else 3 if N > 32 else 1
        else:
            pbnt = 1
        _ck(l.ctdd_unet_conv_patch(C.byref(a), pbk, pbnt, 32, _st()), "ctdd_unet_conv_patch")
    elif bf16:
        bk = 96 if K % 96 == 0 else 64 if K % 64 == 0 else 32 if K % 32 == 0 else 16
        _ck(l.ctdd_unet_conv(C.byref(a), bk, 1, 0, _st()), "ctdd_unet_conv")
    else:
        bk = 32 if K % 32 == 0 else 16
        bnt = 1 if (bk == 16 or rows <= 512) else (3 if N % 96 == 0 else 4 if N % 128 == 0 else 1)   # few rows (per-sample layers): narrow tiles, more workgroups
        _ck(l.ctdd_unet_conv(C.byref(a), bk, bnt, 1, _st()), "ctdd_unet_conv")
    return out, out_hi


def _cast(x, rows, n, ld_out, bf16):
    """fp32 rows of n values -> rows of ld_out (zero padded) in the GEMM operand type."""
    out = torch.empty((rows, ld_out), dtype=torch.bfloat16 if bf16 else torch.float32, device=x.device)
    _ck(lib().ctdd_unet_cast_rows(x.data_ptr(), rows, n, n, ld_out, out.data_ptr() if bf16 else None, None if bf16 else out.data_ptr(), _st()),
        "ctdd_unet_cast_rows")
    return out


def _w_op(w, bf16):
    """[N][K] weight as the forward GEMM's operand."""
    return w.detach().to(torch.bfloat16).contiguous() if bf16 else w.detach().contiguous()


def _wt_op(w, bf16, ld=None):
    """[K][ld] transposed weight (columns >= N zero): the data-gradient GEMM's operand."""
    N, K = w.shape
    dt = torch.bfloat16 if bf16 else torch.float32
    if ld is None or ld == N:
        return w.detach().t().to(dt).contiguous()
    wt = torch.zeros((K, ld), dtype=dt, device=w.device)
    wt[:, :N] = w.detach().t()
    return wt


_tables = {}


def _device_table(raw, dev):
    """Device copy of a launch table; the caching allocator hands a steady-state training step the same buffers every
    iteration, so the tables repeat and are uploaded once."""
    key = (raw, dev.index)
    t = _tables.get(key)
    if t is None:
        if len(_tables) > 8192:
            _tables.clear()
        t = _tables[key] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    return t


def _wgrad_geometry(rows, N, K, bf16, budget=144 * 1024):
    """(nwn, nlr) of a one-tap weight-gradient entry: as unet_train.TrainCtx.wgrad_geometry (kernel LDS + staging-slot limits)."""
    tb, epv = (64, 8) if bf16 else (128, 4)
    for nwn in ([1, 2, 4] if N <= 32 else ([4, 2, 1] if K <= 32 else [2, 4, 1])):
        nwc = 4 // nwn
        nlr = min(budget // ((nwn + nwc) * tb), 2048 // (32 * nwn // epv), 2560 // (32 * nwc // epv)) // 16 * 16
        if nlr >= 16:
            return nwn, min(nlr, -(-rows // 16) * 16)
    raise native.CtddError("weight gradient: no chunk fits the kernel's staging slots")


WGRAD_WORKGROUPS = 512          # M-split target per launch (two per CU: measured best over 256 / 512 / 768 / 1536 at 28800 rows;
                                # half of it for the small products -- up to 384 x 128 --, whose flush atomics weigh more: 11.1 vs 12.9 us)


COLSUM_REPLICAS = 1             # rows of the zeroed bias-gradient scratch a block hands to _wgrad (the column sums ride on the
                                # weight-gradient launch; _colsum's two-stage sum with 16 replicas remains for callers without one)


def _colsum(dy_op, rows, N, ld, bf16, rep=None):
    """db[N] = column sums of dY: 512 workgroups over the rows into COLSUM_REPLICAS partial rows, then their sum.
    rep: zeroed (COLSUM_REPLICAS, N rounded up to 8) scratch."""
    dev = dy_op.device
    n8 = -(-N // 8) * 8
    nblk = int(max(1, min(512, rows // 32)))
    if rep is None:
        rep = torch.zeros((16, n8), dtype=torch.float32, device=dev)
    out = torch.empty((n8,), dtype=torch.float32, device=dev)
    _ck(lib().ctdd_hollow_colsum(None if bf16 else dy_op.data_ptr(), dy_op.data_ptr() if bf16 else None, rows, n8, ld, rep.data_ptr(), nblk,
                                 rep.shape[0], _st()), "ctdd_hollow_colsum")
    _ck(lib().ctdd_unet_sum_batch(rep.data_ptr(), rep.shape[0], n8, 1, n8, out.data_ptr(), 0, _st()), "ctdd_unet_sum_batch")
    return out[:N]


def _wgrad(x_op, dy_op, rows, N, ld, K, bf16, bias, bufs=None):
    """dW[N][K] = dY^T X (tokens are the contraction index: ctdd_unet_wgrad, kind 1x1) and, with `bias`, db[N] = the column
    sums of dY from the same launch (ctdd_wgrad_args.gb: they ride on the staging of the channel-group-0 workgroups; a second
    table entry against an all-ones input cost 12-30 us per launch, the separate two-stage column sum 8.7 us + a reduction).
    bufs: zeroed (dW, (1, N) bias-gradient) scratch."""
    dev = dy_op.device
    dw = bufs[0] if bufs is not None else torch.zeros((N, K), dtype=torch.float32, device=dev)
    a = unet_train._WgradArgs()
    a.x, a.dy, a.gw = x_op.data_ptr(), dy_op.data_ptr(), dw.data_ptr()
    a.B, a.H, a.W, a.Hin, a.Win, a.N, a.ldy, a.C, a.Ktot, a.koff, a.kind = 1, rows, 1, rows, 1, N, ld, K, K, 0, unet_train.WG_1x1
    a.nwn, a.nlr = _wgrad_geometry(rows, N, K, bf16)
    a.nchunks = -(-rows // a.nlr)
    groups = -(-N // (32 * a.nwn)) * -(-K // (32 * (4 // a.nwn)))
    a.grid_x, a.tap = max(1, min(a.nchunks, -(-(WGRAD_WORKGROUPS if N * K > 384 * 128 else WGRAD_WORKGROUPS // 2) // groups))), 0
    db = None
    if bias:                                       # column sums of dY on the staging of the same launch (ctdd_wgrad_args.gb)
        rep = bufs[1] if bufs is not None and bufs[1] is not None else torch.zeros((1, -(-N // 8) * 8), dtype=torch.float32, device=dev)
        db = rep.view(-1)[:N]
        a.gb = db.data_ptr()
    tab = (unet_train._WgradArgs * 1)(a)
    _ck(lib().ctdd_unet_wgrad(_device_table(bytes(tab), dev).data_ptr(), C.addressof(tab), 1, 0 if bf16 else 1, _st()), "ctdd_unet_wgrad")
    return dw, db


def _layernorm(x, y, gamma, beta, film, eps, want_f32=True, want_hi=False):
    B, T, E = x.shape
    out = torch.empty_like(x) if want_f32 else None
    out_hi = torch.empty((B, T, E), dtype=torch.bfloat16, device=x.device) if want_hi else None
    a = _LnArgs()
    a.x, a.y, a.x_bs, a.y_bs, a.out_bs = x.data_ptr(), _p(y), T * E, T * E, T * E
    a.gamma, a.beta, a.eps = gamma.data_ptr(), beta.data_ptr(), float(eps)
    a.film, a.film_stride, a.B, a.T, a.E, a.out = _p(film), 0 if film is None else film.shape[1], B, T, E, _p(out)
    a.out_hi, a.out_hi_bs = _p(out_hi), T * E
    _ck(lib().ctdd_hollow_layernorm(C.byref(a), _st()), "ctdd_hollow_layernorm")
    return out, out_hi


LN_REPLICAS = 32


def _zeros(dev, *shapes):
    """fp32 zero tensors of the given shapes out of ONE fill."""
    sizes = [int(math.prod(sh)) for sh in shapes]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    out, o = [], 0
    for sh, n in zip(shapes, sizes):
        out.append(flat[o:o + n].view(sh))
        o += n
    return out


def _layernorm_bwd(x, y, gamma, beta, film, eps, dout, dres=None, rep=None):
    """-> (dx, dgamma, dbeta, dfilm); dres: dx = dres + gradient (the residual stream's incoming gradient, out of place).
    rep: zeroed (LN_REPLICAS, 2E) scratch for the replicated dgamma | dbeta accumulators."""
    B, T, E = x.shape
    dx = torch.empty_like(x)
    if rep is None:
        rep = torch.zeros((LN_REPLICAS, 2 * E), dtype=torch.float32, device=x.device)
    dfilm = None if film is None else torch.zeros_like(film)
    a = _LnBwdArgs()
    a.x, a.y, a.x_bs, a.y_bs = x.data_ptr(), _p(y), T * E, T * E
    a.gamma, a.beta, a.eps = gamma.data_ptr(), beta.data_ptr(), float(eps)
    a.film, a.film_stride = _p(film), 0 if film is None else film.shape[1]
    a.dout, a.dout_bs, a.B, a.T, a.E, a.rpw = dout.data_ptr(), T * E, B, T, E, 0      # (rows per wave: the launcher's choice)
    a.dx, a.dx_bs, a.acc_dx = dx.data_ptr(), T * E, 0
    a.dgamma, a.dbeta, a.dfilm = rep.data_ptr(), rep.data_ptr() + 4 * E, _p(dfilm)
    a.nrep, a.rep_stride = rep.shape[0], 2 * E
    a.dres, a.dres_bs = _p(dres), T * E
    _ck(lib().ctdd_hollow_layernorm_bwd(C.byref(a), _st()), "ctdd_hollow_layernorm_bwd")
    gb = torch.empty((2 * E,), dtype=torch.float32, device=x.device)
    _ck(lib().ctdd_unet_sum_batch(rep.data_ptr(), rep.shape[0], 2 * E, 1, 2 * E, gb.data_ptr(), 0, _st()), "ctdd_unet_sum_batch")
    return dx, gb[:E], gb[E:], dfilm


def _attn_args(q, k, v, B, Tq, Tk, H, hd, mode, drop_p, rng, layer):
    E = H * hd
    a = _AttnTrainArgs()
    if k is None:                                       # packed qkv rows (B*T, 3E)
        a.q, a.k, a.v = q.data_ptr(), q.data_ptr() + 4 * E, q.data_ptr() + 8 * E
        a.q_bs = a.k_bs = a.v_bs = Tq * 3 * E
        a.q_rs = a.k_rs = a.v_rs = 3 * E
    else:
        a.q, a.k, a.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
        a.q_bs, a.k_bs, a.v_bs, a.q_rs, a.k_rs, a.v_rs = Tq * E, Tk * E, Tk * E, E, E, E
    a.B, a.Tq, a.Tk, a.H, a.hd, a.mode, a.scale = B, Tq, Tk, H, hd, mode, 1.0 / math.sqrt(hd)
    a.out_rs, a.drop_p, a.rng, a.layer = E, float(drop_p), _p(rng), int(layer)
    return a


def _attention_fwd(q, k, v, B, Tq, Tk, H, hd, mode, drop_p, rng, layer, bf16):
    """-> (ctx fp32 (B*Tq, E), bf16 copy or None, stats)."""
    E = H * hd
    dev = q.device
    out = torch.empty((B * Tq, E), dtype=torch.float32, device=dev)
    stats = torch.empty((B, H, Tq, 4), dtype=torch.float32, device=dev)
    a = _attn_args(q, k, v, B, Tq, Tk, H, hd, mode, drop_p, rng, layer)
    a.out, a.stats = out.data_ptr(), stats.data_ptr()
    if bf16 and hd in (16, 32):
        out_hi = torch.empty((B * Tq, E), dtype=torch.bfloat16, device=dev)
        a.out_bf16 = out_hi.data_ptr()
        _ck(lib().ctdd_hollow_attention_train_bf16(C.byref(a), _st()), "ctdd_hollow_attention_train_bf16")
        return out, out_hi, stats
    _ck(lib().ctdd_hollow_attention_train(C.byref(a), _st()), "ctdd_hollow_attention_train")
    return out, (_cast(out, B * Tq, E, E, True) if bf16 else None), stats


def _attention_bwd(q, k, v, out, stats, dout, B, Tq, Tk, H, hd, mode, drop_p, rng, layer, bf16, want_f32=True):
    """-> (dq, dk, dv) fp32 (dqkv, None, None for packed rows) and their bf16 copies when the matrix-core kernels ran."""
    E = H * hd
    mfma = bf16 and hd in (16, 32)
    a = _attn_args(q, k, v, B, Tq, Tk, H, hd, mode, drop_p, rng, layer)
    a.out, a.stats, a.d_out = out.data_ptr(), stats.data_ptr(), dout.data_ptr()
    f32 = want_f32 or not mfma
    hi = [None, None, None]
    if k is None:
        dqkv = torch.empty_like(q) if f32 else None
        if f32:
            a.dq, a.dk, a.dv = dqkv.data_ptr(), dqkv.data_ptr() + 4 * E, dqkv.data_ptr() + 8 * E
        a.dq_bs = a.dk_bs = a.dv_bs = Tq * 3 * E
        a.dq_rs = a.dk_rs = a.dv_rs = 3 * E
        if mfma:
            hi[0] = torch.empty(q.shape, dtype=torch.bfloat16, device=q.device)
            a.dq_bf16, a.dk_bf16, a.dv_bf16 = hi[0].data_ptr(), hi[0].data_ptr() + 2 * E, hi[0].data_ptr() + 4 * E
        grads = (dqkv, None, None)
    else:
        grads = tuple(torch.empty_like(t) if f32 else None for t in (q, k, v))
        if f32:
            a.dq, a.dk, a.dv = (g.data_ptr() for g in grads)
        a.dq_bs, a.dk_bs, a.dv_bs, a.dq_rs, a.dk_rs, a.dv_rs = Tq * E, Tk * E, Tk * E, E, E, E
        if mfma:
            hi = [torch.empty(t.shape, dtype=torch.bfloat16, device=t.device) for t in (q, k, v)]
            a.dq_bf16, a.dk_bf16, a.dv_bf16 = (t.data_ptr() for t in hi)
    if mfma:
        _ck(lib().ctdd_hollow_attention_bwd_bf16(C.byref(a), _st()), "ctdd_hollow_attention_bwd_bf16")
    else:
        _ck(lib().ctdd_hollow_attention_bwd(C.byref(a), _st()), "ctdd_hollow_attention_bwd")
    return grads, hi


def _act(pre, dout, act, drop_p, rng, layer, want_f32=True, want_hi=False):
    out = torch.empty_like(pre) if want_f32 else None
    out_hi = torch.empty(pre.shape, dtype=torch.bfloat16, device=pre.device) if want_hi else None
    _ck(lib().ctdd_hollow_act(pre.data_ptr(), _p(dout), _p(out), _p(out_hi), pre.numel(), act, float(drop_p), _p(rng), int(layer), _st()),
        "ctdd_hollow_act")
    return out, out_hi


def _dropout(x, p, rng, layer, res=None, want_f32=True, want_hi=False):
    """dropout(x) (+ res) -> (fp32 or None, bf16 or None) in one pass (p = 0: an add / a cast)."""
    out = torch.empty_like(x) if want_f32 else None
    out_hi = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_hi else None
    _ck(lib().ctdd_hollow_dropout(x.data_ptr(), _p(res), _p(out), _p(out_hi), x.numel(), float(p), _p(rng) if p > 0 else None, int(layer), _st()),
        "ctdd_hollow_dropout")
    return out, out_hi


def _add(p, q):
    out = torch.empty_like(p)
    n = p.numel()
    _ck(lib().ctdd_hollow_add(p.data_ptr(), n, q.data_ptr(), n, out.data_ptr(), None, None, n, 1, n, _st()), "ctdd_hollow_add")
    return out


def _linear_bwd(x_op, w, dy, rows, bf16, has_bias, need_dx=True, dy_hi=None, wt=None, bufs=None):
    """Gradients of y = x W^T + b given dy (fp32, or its bf16 copy dy_hi): (dx fp32, dW, db).
    wt: the packed transposed weight [K][ld] when the trainer prepared it; bufs: zeroed (dW, [N][8]) targets."""
    N, K = w.shape
    ld = -(-N // 16) * 16
    if bf16 and dy_hi is not None and ld == N:
        dy_op = dy_hi
    else:
        dy_op = _cast(dy, rows, N, ld, bf16) if (bf16 or ld != N) else dy
    dw, db = _wgrad(x_op, dy_op, rows, N, ld, K, bf16, has_bias, bufs)
    dx = _gemm(dy_op, wt if wt is not None else _wt_op(w, bf16, ld), None, None, rows, ld, K, bf16)[0] if need_dx else None
    return dx, dw, db


# ---------------------------------------------------------------------- autograd Functions
class LinearFn(torch.autograd.Function):
    """y = x @ W^T + b (+ res); x (rows, K) fp32."""

    @staticmethod
    def forward(ctx, x, w, b, res, bf16, pack=None):
        rows, K = x.shape
        x = x.contiguous()
        xo = _cast(x, rows, K, K, True) if bf16 else x
        wf = pack[0] if pack is not None else _w_op(w, bf16)
        out, _ = _gemm(xo, wf, None if b is None else b.detach(), None if res is None else res.contiguous(), rows, K, w.shape[0], bf16)
        ctx.save_for_backward(xo, w)
        ctx.bf16, ctx.has_b, ctx.has_res, ctx.pack = bf16, b is not None, res is not None, pack
        return out

    @staticmethod
    def backward(ctx, dy):
        xo, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx, dw, db = _linear_bwd(xo, w, dy, xo.shape[0], ctx.bf16, ctx.has_b and ctx.needs_input_grad[2], need_dx=ctx.needs_input_grad[0],
                                 wt=None if ctx.pack is None else ctx.pack[1])
        return dx, dw, db, (dy if ctx.has_res else None), None, None


def linear(x, lin_w, lin_b, bf16, res=None, pack=None):
    return LinearFn.apply(x, lin_w, lin_b, res, bf16, pack)


class LayerNormFn(torch.autograd.Function):
    """out = FiLM_b(LayerNorm(x (+ y))): x, y (B, T, E) contiguous; film (B, 2E) or None."""

    @staticmethod
    def forward(ctx, x, y, gamma, beta, film, eps):
        x = x.contiguous()
        y = None if y is None else y.contiguous()
        out, _ = _layernorm(x, y, gamma, beta, film, eps)
        ctx.save_for_backward(x, y, gamma, beta, film)
        ctx.eps = float(eps)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y, gamma, beta, film = ctx.saved_tensors
        dx, dg, db, dfilm = _layernorm_bwd(x, y, gamma, beta, film, ctx.eps, dout.contiguous())
        return dx, (dx if y is not None else None), dg, db, dfilm, None


class AttentionFn(torch.autograd.Function):
    """Masked multi-head attention softmax(q k^T / sqrt(hd)) v with dropout on the probabilities.
    self-attention: qkv (B*T, 3E) with k = None;  readout: q (B*Tq, E), k, v (B*Tk, E)."""

    @staticmethod
    def forward(ctx, q, k, v, B, Tq, Tk, H, hd, mode, drop_p, rng, layer, bf16=False):
        q = q.contiguous()
        if k is not None:
            k, v = k.contiguous(), v.contiguous()
        out, _, stats = _attention_fwd(q, k, v, B, Tq, Tk, H, hd, mode, drop_p, rng, layer, bf16)
        ctx.save_for_backward(q, k, v, out, stats, rng)
        ctx.meta = (B, Tq, Tk, H, hd, mode, float(drop_p), int(layer), bf16)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, stats, rng = ctx.saved_tensors
        B, Tq, Tk, H, hd, mode, drop_p, layer, bf16 = ctx.meta
        grads, _ = _attention_bwd(q, k, v, out, stats, dout.contiguous(), B, Tq, Tk, H, hd, mode, drop_p, rng, layer, bf16)
        return grads + (None,) * 10


class AttnBlockFn(torch.autograd.Function):
    """h + dropout(out_proj(attention(in_proj(LayerNorm(h))))): one prenorm self-attention block (hollow_networks.py:311-340)
    with a hand-scheduled backward: bf16 operand copies come out of the producing kernels, the residual stream's gradient
    is accumulated by the LayerNorm backward."""

    @staticmethod
    def forward(ctx, h, ln_w, ln_b, w_in, b_in, w_out, b_out, rng, meta):
        B, D, H, hd, mode, p_att, p_drop, l_att, l_drop, bf16, eps, pk_in, pk_out = meta
        E, R = H * hd, B * D
        h = h.contiguous()
        z, z_hi = _layernorm(h, None, ln_w, ln_b, None, eps, want_f32=not bf16, want_hi=bf16)
        z_op = (z_hi if bf16 else z).view(R, E)
        qkv, _ = _gemm(z_op, pk_in[0], b_in.detach(), None, R, E, 3 * E, bf16)
        att, att_hi, stats = _attention_fwd(qkv, None, None, B, D, D, H, hd, mode, p_att, rng if p_att > 0 else None, l_att, bf16)
        att_op = att_hi if bf16 else att
        if p_drop > 0.0 and not _gemm_fusable(E, E, bf16):
            o, _ = _gemm(att_op, pk_out[0], b_out.detach(), None, R, E, E, bf16)
            out, _ = _dropout(o, p_drop, rng, l_drop, res=h.view(R, E))
        else:                                                  # h + dropout(out_proj(.)) in the epilogue
            out, _ = _gemm(att_op, pk_out[0], b_out.detach(), h.view(R, E), R, E, E, bf16, drop=(p_drop, rng, l_drop) if p_drop > 0.0 else None)
        ctx.save_for_backward(h, ln_w, ln_b, w_in, w_out, z_op, qkv, att, att_op, stats, rng)
        ctx.meta = meta
        return out.view(B, D, E)

    @staticmethod
    def backward(ctx, dout):
        h, ln_w, ln_b, w_in, w_out, z_op, qkv, att, att_op, stats, rng = ctx.saved_tensors
        B, D, H, hd, mode, p_att, p_drop, l_att, l_drop, bf16, eps, pk_in, pk_out = ctx.meta
        E, R = H * hd, B * D
        dout = dout.contiguous()
        zw_out, zb_out, zw_in, zb_in, rep = _zeros(dout.device, (E, E), (COLSUM_REPLICAS, E), (3 * E, E), (COLSUM_REPLICAS, 3 * E), (LN_REPLICAS, 2 * E))
        if bf16:                                             # dropout mask and bf16 operand copy in one pass over dout
            do, do_hi = None, _dropout(dout.view(R, E), p_drop, rng, l_drop, want_f32=False, want_hi=True)[1]
        else:
            do, do_hi = (_dropout(dout.view(R, E), p_drop, rng, l_drop)[0] if p_drop > 0.0 else dout.view(R, E)), None
        datt, dw_out, db_out = _linear_bwd(att_op, w_out, do, R, bf16, True, dy_hi=do_hi, wt=pk_out[1], bufs=(zw_out, zb_out))
        (dqkv, _, _), hi = _attention_bwd(qkv, None, None, att, stats, datt, B, D, D, H, hd, mode, p_att, rng if p_att > 0 else None, l_att, bf16,
                                          want_f32=False)
        dz, dw_in, db_in = _linear_bwd(z_op, w_in, dqkv, R, bf16, True, dy_hi=hi[0], wt=pk_in[1], bufs=(zw_in, zb_in))
        dh, dg, dbeta, _ = _layernorm_bwd(h, None, ln_w, ln_b, None, eps, dz.view(B, D, E), dres=dout, rep=rep)
        return dh, dg, dbeta, dw_in, db_in, dw_out, db_out, None, None


class MlpBlockFn(torch.autograd.Function):
    """h + dropout(fc2(dropout(relu(fc1(LayerNorm(h)))))): one prenorm feed-forward block (hollow_networks.py:343-420).
    bf16 mode: the (rows, mlp_dim) hidden tensor and its gradient exist in bf16 only (the fp32 copies were the largest
    streams of the block); the saved output u = dropout(relu(.)) is its own backward mask."""

    @staticmethod
    def forward(ctx, h, ln_w, ln_b, w1, b1, w2, rng, meta):
        B, D, E, p_drop, l1, l2, bf16, eps, pk1, pk2 = meta
        R, M = B * D, w1.shape[0]
        h = h.contiguous()
        z, z_hi = _layernorm(h, None, ln_w, ln_b, None, eps, want_f32=not bf16, want_hi=bf16)
        z_op = (z_hi if bf16 else z).view(R, E)
        if bf16:
            # ReLU in the GEMM epilogue; with dropout one in-place pass over the bf16 tensor (relu is idempotent)
            fuse1 = _gemm_fusable(E, M, True)                  # dropout in the epilogue too (same masks as the in-place pass)
            _, u_op = _gemm(z_op, pk1[0], b1.detach(), None, R, E, M, True, act=1, want_hi=True, want_f32=False,
                            drop=(p_drop, rng, l1) if (fuse1 and p_drop > 0.0) else None)
            if p_drop > 0.0 and not fuse1:
                _ck(lib().ctdd_hollow_relu_bf16(u_op.data_ptr(), None, u_op.data_ptr(), u_op.numel(), float(p_drop), rng.data_ptr(), int(l1), _st()),
                    "ctdd_hollow_relu_bf16")
            pre = u_op
        elif p_drop > 0.0:
            pre, _ = _gemm(z_op, pk1[0], b1.detach(), None, R, E, M, False)
            u_op, _ = _act(pre, None, 1, p_drop, rng, l1)
        else:
            pre, _ = _gemm(z_op, pk1[0], b1.detach(), None, R, E, M, False, act=1)     # relu(pre): same ReLU mask
            u_op = pre
        if p_drop > 0.0 and not _gemm_fusable(M, E, bf16):
            o, _ = _gemm(u_op, pk2[0], None, None, R, M, E, bf16)
            out, _ = _dropout(o, p_drop, rng, l2, res=h.view(R, E))
        else:                                                  # h + dropout(fc2(u)) in the epilogue
            out, _ = _gemm(u_op, pk2[0], None, h.view(R, E), R, M, E, bf16, drop=(p_drop, rng, l2) if p_drop > 0.0 else None)
        ctx.save_for_backward(h, ln_w, ln_b, w1, w2, z_op, pre, u_op, rng)
        ctx.meta = meta
        return out.view(B, D, E)

    @staticmethod
    def backward(ctx, dout):
        h, ln_w, ln_b, w1, w2, z_op, pre, u_op, rng = ctx.saved_tensors
        B, D, E, p_drop, l1, l2, bf16, eps, pk1, pk2 = ctx.meta
        R, M = B * D, w1.shape[0]
        dout = dout.contiguous()
        zw2, zw1, zb1, rep = _zeros(dout.device, (E, M), (M, E), (COLSUM_REPLICAS, M), (LN_REPLICAS, 2 * E))
        if bf16:
            do_hi = _dropout(dout.view(R, E), p_drop, rng, l2, want_f32=False, want_hi=True)[1]
            dw2, _ = _wgrad(u_op, do_hi, R, E, E, M, True, False, (zw2, None))
            if _gemm_fusable(E, M, True):                      # du * [u != 0] / (1 - p) in the epilogue of du = do W2
                _, du_hi = _gemm(do_hi, pk2[1], None, None, R, E, M, True, want_hi=True, want_f32=False, drop=(p_drop, None, 0), mask_u=u_op)
            else:
                _, du_hi = _gemm(do_hi, pk2[1], None, None, R, E, M, True, want_hi=True, want_f32=False)
                _ck(lib().ctdd_hollow_relu_bf16(du_hi.data_ptr(), u_op.data_ptr(), du_hi.data_ptr(), du_hi.numel(), float(p_drop), None, 0, _st()),
                    "ctdd_hollow_relu_bf16")
            dz, dw1, db1 = _linear_bwd(z_op, w1, None, R, True, True, dy_hi=du_hi, wt=pk1[1], bufs=(zw1, zb1))
        else:
            do = _dropout(dout.view(R, E), p_drop, rng, l2)[0] if p_drop > 0.0 else dout.view(R, E)
            du, dw2, _ = _linear_bwd(u_op, w2, do, R, False, False, wt=pk2[1], bufs=(zw2, None))
            dpre, _ = _act(pre, du, 1, p_drop, rng if p_drop > 0 else None, l1)
            dz, dw1, db1 = _linear_bwd(z_op, w1, dpre, R, False, True, wt=pk1[1], bufs=(zw1, zb1))
        dh, dg, dbeta, _ = _layernorm_bwd(h, None, ln_w, ln_b, None, eps, dz.view(B, D, E), dres=dout, rep=rep)
        return dh, dg, dbeta, dw1, db1, dw2, None, None


class ActFn(torch.autograd.Function):
    """dropout(act(x)): act 1 ReLU, 2 GELU (erf)."""

    @staticmethod
    def forward(ctx, pre, act, drop_p, rng, layer):
        pre = pre.contiguous()
        out, _ = _act(pre, None, act, drop_p, rng, layer)
        ctx.save_for_backward(pre, rng)
        ctx.meta = (act, float(drop_p), int(layer))
        return out

    @staticmethod
    def backward(ctx, dout):
        pre, rng = ctx.saved_tensors
        act, drop_p, layer = ctx.meta
        return _act(pre, dout.contiguous(), act, drop_p, rng, layer)[0], None, None, None, None


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, drop_p, rng, layer):
        ctx.save_for_backward(rng)
        ctx.meta = (float(drop_p), int(layer))
        return _dropout(x.contiguous(), drop_p, rng, layer)[0]

    @staticmethod
    def backward(ctx, dy):
        (rng,) = ctx.saved_tensors
        return _dropout(dy.contiguous(), ctx.meta[0], rng, ctx.meta[1])[0], None, None, None


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, q):
        return _add(p.contiguous(), q.contiguous())

    @staticmethod
    def backward(ctx, d):
        return d, d


class EmbedFn(torch.autograd.Function):
    """(l2r, r2l, temb) token sequences from the integer state (hollow_networks.py:729-753)."""

    @staticmethod
    def forward(ctx, x, t, w_in, b_in, pe, S, temb_scale):
        B, D = x.shape
        E = w_in.numel()
        dev = x.device
        l2r, r2l, temb = (torch.empty((B, D, E), dtype=torch.float32, device=dev), torch.empty((B, D, E), dtype=torch.float32, device=dev),
                          torch.empty((B, E), dtype=torch.float32, device=dev))
        x = x.contiguous()
        a = _EmbedArgs()
        if x.dtype == torch.int64:
            a.x64 = x.data_ptr()
        else:
            a.x32 = x.data_ptr()
        wv, bv, tv = w_in.detach().reshape(-1).contiguous(), b_in.detach().contiguous(), t.float().contiguous()
        a.t, a.w_in, a.b_in, a.pe = tv.data_ptr(), wv.data_ptr(), bv.data_ptr(), pe.data_ptr()
        a.B, a.D, a.E, a.S, a.temb_scale = B, D, E, S, float(temb_scale)
        a.l2r, a.r2l, a.temb = l2r.data_ptr(), r2l.data_ptr(), temb.data_ptr()
        _ck(lib().ctdd_hollow_embed(C.byref(a), _st()), "ctdd_hollow_embed")
        ctx.save_for_backward(x)
        ctx.meta = (S, w_in.shape)
        ctx.mark_non_differentiable(temb)
        return l2r, r2l, temb

    @staticmethod
    def backward(ctx, dl2r, dr2l, _dtemb):
        (x,) = ctx.saved_tensors
        S, wshape = ctx.meta
        B, D = x.shape
        E = dl2r.shape[-1]
        dl2r, dr2l = dl2r.contiguous(), dr2l.contiguous()
        dw = torch.zeros((E,), dtype=torch.float32, device=x.device)
        db = torch.zeros((E,), dtype=torch.float32, device=x.device)
        a = _EmbedBwdArgs()
        if x.dtype == torch.int64:
            a.x64 = x.data_ptr()
        else:
            a.x32 = x.data_ptr()
        a.dl2r, a.dr2l, a.B, a.D, a.E, a.S, a.dw, a.db = dl2r.data_ptr(), dr2l.data_ptr(), B, D, E, S, dw.data_ptr(), db.data_ptr()
        _ck(lib().ctdd_hollow_embed_bwd(C.byref(a), _st()), "ctdd_hollow_embed_bwd")
        return None, None, dw.view(wshape), db, None, None, None


# ---------------------------------------------------------------------- the network
def training_supported(model):
    net = _unwrap(getattr(model, "net", None))
    if net is None or getattr(net.config.model, "engine_train", "hip") != "hip" or not supports(model):
        return False
    m = net.config.model
    return (m.embed_dim // m.num_heads) in (4, 8, 16, 32) and m.embed_dim <= 256 and m.embed_dim % 16 == 0 and m.mlp_dim % 16 == 0


class HollowTrainer:
    def __init__(self, model, precision=None):
        self.model, self.net = model, _unwrap(model.net)
        m = self.net.config.model
        self.precision = precision or getattr(m, "engine_train_precision", "bf16")
        if self.precision not in ("fp32", "bf16"):
            raise ValueError(f"unknown training precision {self.precision}")
        self.dev = next(self.net.parameters()).device
        if self.dev.type != "cuda":
            raise native.CtddError("HollowTrainer needs the model on a GPU")
        lib()
        self.rng = torch.zeros(2, dtype=torch.int64, device=self.dev)
        self.rng[0] = native.dropout_seed()
        self.pe = None
        self._pack_key, self._packs, self._pack_tab, self._pack_total = None, {}, None, 0

    def _pack_weights(self, bump):
        """ONE launch converts every matrix-core linear's weight into the forward operand [N][K] and the data-gradient
        operand [K][N'] (transposed, N' = N rounded up to 16) in the mode's operand type; it also advances the dropout
        stream.  The table is rebuilt when a parameter's storage moves."""
        net, bf = self.net, self.precision == "bf16"
        ws = [p for n, p in net.named_parameters()
              if p.dim() == 2 and p.shape[1] % 16 == 0 and not n.startswith(("embedding", "temb_net", "input_embedding"))]
        key = tuple(p.data_ptr() for p in ws)
        if key != self._pack_key:
            dt = torch.bfloat16 if bf else torch.float32
            total = sum(2 * p.shape[1] * (-(-p.shape[0] // 16) * 16) for p in ws)
            arena = torch.zeros(total, dtype=dt, device=self.dev)
            tab = (unet_train._PackEntry * len(ws))()
            o, first, packs = 0, 0, {}
            for i, p in enumerate(ws):
                N, K = p.shape
                ld = -(-N // 16) * 16
                fwd, dg = arena[o:o + N * K].view(N, K), arena[o + ld * K:o + 2 * ld * K].view(K, ld)
                o += 2 * ld * K
                t = tab[i]
                t.w, t.fwd, t.dgrad = p.data_ptr(), fwd.data_ptr(), dg.data_ptr()
                t.N, t.Cin_tot, t.c_off, t.C, t.ntap, t.Ktot, t.koff, t.flip, t.ldd, t.first = N, K, 0, K, 1, K, 0, 0, ld, first
                first += N * K
                packs[id(p)] = (fwd, dg)
            self._pack_key, self._packs, self._pack_total, self._arena = key, packs, first, arena
            self._pack_tab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.dev)
            self._pack_n = len(ws)
        _ck(lib().ctdd_unet_pack_weights(self._pack_tab.data_ptr(), self._pack_n, self._pack_total, 0 if bf else 1,
                                         self.rng.data_ptr() if bump else None, _st()), "ctdd_unet_pack_weights")

    def pk(self, w):
        return self._packs[id(w)]

    def __call__(self, x, times):
        net = self.net
        m = net.config.model
        bf = self.precision == "bf16"
        E, H, S = m.embed_dim, m.num_heads, net.S
        hd = E // H
        B, D = x.shape
        training = bool(self.model.training)
        p_drop = float(m.dropout_rate) if training else 0.0
        p_att = float(m.attention_dropout_rate) if training else 0.0
        self._pack_weights(bump=training)                     # (+ one dropout stream per training forward)
        # every forward keeps its OWN {seed, step}: the Functions save this tensor and their backward kernels read the step
        # from it, so a second training forward before the first one's backward (two-forward-pass CT-ELBO, gradient
        # accumulation over micro-batches) must not move the first one's masks
        rng, pk = (self.rng.clone() if training else self.rng), self.pk
        layer = [0]

        def nxt():
            layer[0] += 1
            return layer[0]

        def drop(t, p):
            return DropoutFn.apply(t, p, rng, nxt()) if p > 0.0 else t

        if self.pe is None or self.pe.shape[0] < D:
            self.pe = net.module_l2r.pos_embed.pe[0, :D].to(self.dev).float().contiguous()
        l2r, r2l, temb = EmbedFn.apply(x, times, net.input_embedding.weight, net.input_embedding.bias, self.pe, S, float(net.temb_scale))
        R = B * D
        streams = []
        for seq, stack, mode in ((l2r, net.module_l2r, 0), (r2l, net.module_r2l, 1)):
            h = drop(drop(seq, p_drop), p_drop)               # PositionalEncoding's dropout, then the stack's (hollow_networks.py:562-563)
            for blk in stack.trans_block_layers:
                sa, ff = blk.self_attention_block, blk.feed_forward_block
                mha = sa.self_attention
                h = AttnBlockFn.apply(h, sa.norm.weight, sa.norm.bias, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                                      mha.out_proj.bias, rng,
                                      (B, D, H, hd, mode, p_att, p_drop, nxt(), nxt(), bf, sa.norm.eps, pk(mha.in_proj_weight), pk(mha.out_proj.weight)))
                h = MlpBlockFn.apply(h, ff.norm.weight, ff.norm.bias, ff.mlp.fc1.weight, ff.mlp.fc1.bias, ff.mlp.fc2.weight, rng,
                                     (B, D, E, p_drop, nxt(), nxt(), bf, ff.norm.eps, pk(ff.mlp.fc1.weight), pk(ff.mlp.fc2.weight)))
            streams.append(h)
        l2r, r2l = streams
        def dense(x_, w_, b_, use_bf, res=None):                # packed operands when the pack's type is the layer's
            return linear(x_, w_, b_, use_bf, res, pk(w_) if use_bf == bf else None)

        # ---- attention readout (prenorm): cross attention over [temb | ln1(l2r) | ln2(r2l)] + (l2r + r2l)
        ro = net.readout_module
        ca = ro.cross_attention
        Tk = 2 * D + 1
        a1 = LayerNormFn.apply(l2r, None, ro.ln1.weight, ro.ln1.bias, None, ro.ln1.eps)
        a2 = LayerNormFn.apply(r2l, None, ro.ln2.weight, ro.ln2.bias, None, ro.ln2.eps)
        allk = torch.cat([temb.unsqueeze(1), a1, a2], dim=1).reshape(B * Tk, E)
        qin = AddFn.apply(a1, a2).view(R, E)
        raw = AddFn.apply(l2r, r2l).view(R, E)
        qb = dense(qin, ca.dense_query.weight, None, bf)
        kb = dense(allk, ca.dense_key.weight, ca.dense_key.bias, bf)
        vb = dense(allk, ca.dense_val.weight, ca.dense_val.bias, bf)
        ctxv = AttentionFn.apply(qb, kb, vb, B, D, Tk, H, hd, 2, 0.0, None, nxt(), bf)
        xr = dense(ctxv, ca.out_linear.weight, ca.out_linear.bias, bf, res=raw)
        # ---- FiLM residual readout
        rr = ro.model
        E2 = 2 * E
        lin = [l_ for l_ in rr.mlp.layers if isinstance(l_, torch.nn.Linear)]
        tm = dense(ActFn.apply(dense(temb, lin[0].weight, lin[0].bias, False), 2, 0.0, None, 0), lin[1].weight, lin[1].bias, False)
        hh = dense(xr, rr.input_layer.weight, rr.input_layer.bias, bf)
        for i in range(rr.n_res):
            mlp_i, ln_i = rr.resid_layers[2 * i], rr.resid_layers[2 * i + 1]
            li = [l_ for l_ in mlp_i.layers if isinstance(l_, torch.nn.Linear)]
            r_ = dense(ActFn.apply(dense(hh, li[0].weight, li[0].bias, bf), 2, 0.0, None, 0), li[1].weight, li[1].bias, bf)
            fl = dense(tm, rr.film_layer[i].weight, rr.film_layer[i].bias, False)
            hh = LayerNormFn.apply(hh.view(B, D, E2), r_.view(B, D, E2), ln_i.weight, ln_i.bias, fl, ln_i.eps).view(R, E2)
        logits = dense(hh, rr.logits_layer.weight, rr.logits_layer.bias, bf)
        return logits.view(B, D, rr.out_dim)
