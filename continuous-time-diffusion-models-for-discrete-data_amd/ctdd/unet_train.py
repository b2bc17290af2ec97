"""Training plan of the tauLDR U-Net on hand-written HIP kernels: the forward pass that keeps what backward needs and
the backward pass itself (reference: `l.backward()` in TAUnSDDM/lib/training/training.py:27 through
lib/networks/unet.py:100-140, 152-200, 303-459).

`UNetEngine.train_forward(x, t)` (ctdd/unet_engine.py) returns logits attached to autograd through ONE
`torch.autograd.Function` per network; its backward runs a flat list of libctdd launches built here:

  * every convolution's data gradient is a convolution of the output gradient with tap-flipped, transposed weights, run by the
    forward kernels (ring / patch / generic implicit GEMM, csrc/unet_kernels.hip); the Downsample's transpose is segment kind
    CTDD_SEG_3x3_S2T; channel concatenations and the folded linear skip are per-segment gradients, never materialised;
  * weight gradients by `ctdd_unet_wgrad` (pixels as the contraction index, transposing LDS reads), packed [N][K] fp32
    accumulators un-packed to the torch parameter layout in one launch;
  * GroupNorm + Swish (+ Dropout) backward in two passes, attention / first conv / upsample / bias / time-projection kernels
    (csrc/unet_train_kernels.hip, include/ctdd_unet_train.h);
  * all convolution weights are re-packed from the fp32 master parameters by ONE launch per step.

The tiny time MLP (B x 384 tensors) and the logistic head stay differentiable device ops around the Function.
"""
import ctypes as C

import torch

from . import native
from .unet_engine import SEG_1x1, SEG_3x3, SEG_3x3_S2, _lib

SEG_3x3_S2T = 4
WG_3x3, WG_1x1, WG_3x3_S2 = 0, 1, 2
_P, _I, _F, _I64, _U64 = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_uint64


class _WgradArgs(C.Structure):
    _fields_ = [("x", _P), ("dy", _P), ("gw", _P), ("B", _I), ("H", _I), ("W", _I), ("Hin", _I), ("Win", _I), ("N", _I), ("ldy", _I),
                ("C", _I), ("Ktot", _I), ("koff", _I), ("kind", _I), ("nlr", _I), ("nwn", _I), ("nchunks", _I), ("grid_x", _I), ("tap", _I), ("gb", _P)]


class _GnBwdArgs(C.Structure):
    _fields_ = [("s1_f32", _P), ("s1_bf16", _P), ("st1", _P), ("C1", _I), ("s2_f32", _P), ("s2_bf16", _P), ("st2", _P), ("C2", _I),
                ("gamma", _P), ("beta", _P), ("B", _I), ("HW", _I), ("G", _I), ("eps", _F), ("swish", _I), ("da_f32", _P), ("da_bf16", _P),
                ("sums", _P), ("d1_f32", _P), ("d1_bf16", _P), ("acc1", _I), ("d2_f32", _P), ("d2_bf16", _P), ("acc2", _I),
                ("drop_p", _F), ("rng", _P), ("layer", _U64), ("dsum_bn", _P), ("dsum_stride", _I), ("dsum_n", _P)]


class _AttnBwdArgs(C.Structure):
    _fields_ = [("qkv", _P), ("d_out_f32", _P), ("d_out_bf16", _P), ("B", _I), ("T", _I), ("C", _I), ("heads", _I), ("d_qkv", _P),
                ("d_qkv_bf16", _P)]


class _FirstWgradArgs(C.Structure):
    _fields_ = [("x64", _P), ("x32", _P), ("lo", _F), ("hi", _F), ("dy_f32", _P), ("dy_bf16", _P), ("B", _I), ("Cin", _I), ("H", _I),
                ("W", _I), ("Cout", _I), ("gw", _P), ("gbias", _P), ("partial", _P)]


class _SumJob(C.Structure):
    _fields_ = [("inp", _P), ("out", _P), ("bstride", _I64), ("B", _I), ("jstride", _I), ("n", _I), ("accumulate", _I), ("j0", _I), ("pad_", _I)]


class _PackEntry(C.Structure):
    _fields_ = [("w", _P), ("fwd", _P), ("dgrad", _P), ("gw", _P), ("grad", _P), ("N", _I), ("Cin_tot", _I), ("c_off", _I), ("C", _I),
                ("ntap", _I), ("Ktot", _I), ("koff", _I), ("flip", _I), ("ldd", _I), ("pad_", _I), ("first", _I64)]


TRAIN_EXPORTS = ("ctdd_unet_wgrad", "ctdd_unet_gn_bwd", "ctdd_unet_dropout", "ctdd_unet_colsum", "ctdd_unet_sum_batch", "ctdd_unet_sum_jobs", "ctdd_unet_accumulate",
                 "ctdd_unet_downsum2x", "ctdd_unet_upsample2x_f32", "ctdd_unet_cast_rows", "ctdd_unet_attention_bwd",
                 "ctdd_unet_first_conv_wgrad", "ctdd_unet_first_conv_wgrad_scratch", "ctdd_unet_pack_weights", "ctdd_unet_unpack_grads")
_sigs_done = False


def lib():
    global _sigs_done
    l = _lib()
    if not _sigs_done:
        for name, argt in (("ctdd_unet_wgrad", [_P, _P, _I, _I, _P]), ("ctdd_unet_gn_bwd", [_P, _P]),
                           ("ctdd_unet_dropout", [_P, _P, _I64, _F, _P, _U64, _P]),
                           ("ctdd_unet_colsum", [_P, _P, _I, _I, _I, _I, _P, _I, _P, _P]),
                           ("ctdd_unet_sum_batch", [_P, _I, _I64, _I, _I, _P, _I, _P]),
                           ("ctdd_unet_sum_jobs", [_P, _I, _P]),
                           ("ctdd_unet_accumulate", [_P, _P, _P, _P, _I64, _I, _P]),
                           ("ctdd_unet_downsum2x", [_P, _P, _I, _I, _I, _I, _P, _P, _I, _P]),
                           ("ctdd_unet_upsample2x_f32", [_P, _I, _I, _I, _I, _P, _P]),
                           ("ctdd_unet_cast_rows", [_P, _I64, _I, _I, _I, _P, _P, _P]),
                           ("ctdd_unet_attention_bwd", [_P, _P]), ("ctdd_unet_first_conv_wgrad", [_P, _P]),
                           ("ctdd_unet_pack_weights", [_P, _I, _I64, _I, _P, _P]), ("ctdd_unet_unpack_grads", [_P, _I, _I64, _P])):
            fn = getattr(l, name)
            fn.argtypes, fn.restype = argt, _I
        l.ctdd_unet_first_conv_wgrad_scratch.argtypes, l.ctdd_unet_first_conv_wgrad_scratch.restype = [_I, _I, _I, _I], _I64
        _sigs_done = True
    return l


class _Grad:
    """Gradient buffer of one activation: same NHWC shape, bf16 or fp32 by the engine's mode; `has` = already written while the
    backward plan is being laid out (the next writer accumulates)."""

    def __init__(self, eng, rows, Cn, dev, ld=None):
        ld = ld or Cn
        self.C, self.ld, self.has = Cn, ld, False
        self.f32 = torch.zeros((rows, ld), dtype=torch.float32, device=dev) if eng.precise else None
        self.hi = None if eng.precise else torch.zeros((rows, ld), dtype=torch.bfloat16, device=dev)
        self.stats = None


class TrainCtx:
    """Collects, while `UNetEngine._build` lays out the forward plan, what the backward plan needs; `finish` lays the backward out."""

    def __init__(self, eng, B, dropout):
        self.eng, self.B, self.dropout = eng, B, bool(dropout)
        self.dev = eng.dev
        self.records = []
        self.entries = []            # dicts -> _PackEntry
        self.zero_parts = []         # (name, numel): fp32 regions zeroed at the start of every backward
        self.bias_jobs = []          # (buffer, [params]): folded biases refreshed in the forward plan
        self.layer_id = 0
        self.tproj = self.resblocks = None
        self._last_pack = None
        self.engine_params = []      # parameters whose gradients the backward plan produces, in net.parameters() order
        self.rng = torch.zeros(2, dtype=torch.int64, device=self.dev)          # {seed, step}

    # ------------------------------------------------------------------ forward-side hooks
    def packed_forward(self, wsrc, segs, N, Ktot):
        eng = self.eng
        dt = torch.float32 if eng.precise else torch.bfloat16
        fwd = torch.zeros((N, Ktot), dtype=dt, device=self.dev)
        gw = ("gw", N * Ktot)                                            # region of the zero arena, resolved in finish()
        self.zero_parts.append(gw)
        koff, per_seg = 0, []
        for (w, c_off), (_, cs, kind) in zip(wsrc, segs):
            if not (w.dtype == torch.float32 and w.is_contiguous()):
                raise native.CtddError("training plan: convolution weights must be contiguous fp32 parameters")
            ntap = 1 if kind == SEG_1x1 else 9
            ldd = -(-N // 16) * 16                                       # data-gradient operand: channel count padded to the MFMA K step
            dg = torch.zeros((cs, ntap * ldd), dtype=dt, device=self.dev)
            e = dict(w=w, fwd=fwd, dgrad=dg, gw=gw, N=N, Cin_tot=w.shape[1], c_off=c_off, C=cs, ntap=ntap, Ktot=Ktot, koff=koff,
                     flip=int(kind == SEG_3x3), ldd=ldd)
            self.entries.append(e)
            per_seg.append(e)
            koff += ntap * cs
        self._last_pack = (fwd, gw, per_seg)
        return (None, fwd) if eng.precise else (fwd, None)

    def summed_bias(self, params):
        buf = torch.zeros_like(params[0], dtype=torch.float32)
        self.bias_jobs.append((buf, list(params)))
        return buf

    def record_conv(self, conv, segs, wsrc, bias_params, N, Hout, Wout, Hin, Win, out, tb, res, logits_C, out_f32_tensor):
        fwd, gw, per_seg = self._last_pack
        self.records.append(("conv", dict(segs=list(segs), per_seg=per_seg, gw=gw, bias_params=bias_params or [], N=N, Hout=Hout, Wout=Wout,
                                          Hin=Hin, Win=Win, out=out, tb=tb, res=res, logits_C=logits_C, out_f32=out_f32_tensor)))

    def record_gn(self, srcs, norm, swish, eps, HW, out, drop_p, launch, stats_views):
        layer = self.layer_id
        self.layer_id += 1
        if drop_p > 0.0:                                                  # forward side of the dropout: in place on the activated tensor
            l = lib()
            launch(l.ctdd_unet_dropout, None if out.f32 is None else out.f32.data_ptr(), None if out.hi is None else out.hi.data_ptr(),
                   out.B * out.H * out.W * out.C, float(drop_p), self.rng.data_ptr(), layer, label="dropout")
        self.records.append(("gn", dict(srcs=list(srcs), norm=norm, swish=swish, eps=eps, HW=HW, out=out, drop_p=drop_p, layer=layer,
                                        stats_views=stats_views)))

    def record_first(self, conv0, fa, cur):
        self.records.append(("first", dict(conv=conv0, fa=fa, out=cur)))

    def begin_attention(self, att, x, qkv):
        self._attn = dict(att=att, x=x, qkv=qkv)

    def record_attention(self, att, qkv, ao, B, T, Cx):
        self.records.append(("attn", dict(att=att, qkv=qkv, ao=ao, T=T, Cx=Cx)))

    def record_upsample(self, cur, up):
        self.records.append(("up", dict(cur=cur, up=up)))

    # ------------------------------------------------------------------ backward plan
    def g(self, t):
        """Gradient buffer of activation `t` (a _Tensor of the forward plan)."""
        k = id(t)
        if k not in self._grads:
            self._grads[k] = _Grad(self.eng, t.B * t.H * t.W, t.C, self.dev)
            self._grads[k].B, self._grads[k].H, self._grads[k].W = t.B, t.H, t.W
        return self._grads[k]

    def wgrad_geometry(self, kind, B, H, W, N, Cseg):
        """(nwn, nlr, nchunks) of one weight-gradient table entry: wave arrangement and chunk size within the kernel's LDS
        (one workgroup per CU) and staging-slot limits (8 + 10 sixteen-byte vectors per thread)."""
        eng = self.eng
        tb, epv = (128, 4) if eng.precise else (64, 8)
        budget = int(getattr(eng.cfg.model, "wgrad_lds_bytes", 144 * 1024))
        prefer = [1, 2, 4] if N <= 32 else ([4, 2, 1] if Cseg <= 32 else [2, 4, 1])
        for nwn in prefer:                                 # first arrangement whose smallest chunk fits the staging slots
            nwc = 4 // nwn
            vn, vc = 32 * nwn // epv, 32 * nwc // epv
            if kind == WG_3x3:
                Wp = W + 2

                def fits(n_):
                    KP = -(-(n_ * Wp) // 16) * 16
                    return ((KP * nwn + (KP + 2 * Wp + 2) * nwc) * tb <= budget and n_ * W * vn <= 2048 and (n_ + 2) * W * vc <= 2560)
                if not fits(1):
                    continue
                nlr = 1
                while fits(nlr + 1) and nlr + 1 <= B * (H + 1):
                    nlr += 1
                return nwn, nlr, -(-(B * (H + 1)) // nlr)
            nlr = min(budget // ((nwn + nwc) * tb), 2048 // vn, 2560 // vc) // 16 * 16
            if nlr < 16:
                continue
            nlr = min(nlr, -(-(B * H * W) // 16) * 16)
            return nwn, nlr, -(-(B * H * W) // nlr)
        raise native.CtddError(f"weight gradient: no chunk of a {H}x{W} grid fits the kernel's staging slots")

    def _wgrad_entry(self, x_t, gy, N, ldy, Cseg, e, kind, B, H, W, Hin, Win, gb=None):
        """Queue the weight gradient of one K-segment; all of them run as ONE table launch at the end of the backward plan."""
        eng = self.eng
        wk = {SEG_3x3: WG_3x3, SEG_1x1: WG_1x1, SEG_3x3_S2: WG_3x3_S2}[kind]
        nwn, nlr, nchunks = self.wgrad_geometry(wk, B, H, W, N, Cseg)
        for tap in (range(9) if wk == WG_3x3_S2 else (0,)):
            a = _WgradArgs()
            a.x = (x_t.f32 if eng.precise else x_t.hi).data_ptr()
            a.dy = (gy.f32 if eng.precise else gy.hi).data_ptr()
            a.gw = e["gw_ptr"]
            a.B, a.H, a.W, a.Hin, a.Win, a.N, a.ldy, a.C, a.Ktot, a.koff = B, H, W, Hin, Win, N, ldy, Cseg, e["Ktot"], e["koff"]
            a.kind, a.nlr, a.nwn, a.nchunks, a.tap = wk, nlr, nwn, nchunks, tap
            a.gb = gb if tap == 0 else None              # (bias gradient: the column sums of gy, once per convolution)
            self.wgrad_entries.append(a)

    @staticmethod
    def _sum_jobs_table(jobs, dev):
        """(in, B, bstride, jstride, n, out, accumulate) reductions -> device table of 64-output jobs for ctdd_unet_sum_jobs."""
        rows = []
        for inp, Bn, bstride, jstride, n, out, acc in jobs:
            for j0 in range(0, n, 64):
                rows.append(_SumJob(inp, out, bstride, Bn, jstride, n, acc, j0, 0))
        tab = (_SumJob * len(rows))(*rows)
        return torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev), len(rows)

    def _wgrad_table_launch(self, launch):
        """One ctdd_unet_wgrad launch for every queued entry.  M-split per entry: every workgroup gets about the same number
        of (16-pixel step x tap) units, ~`wgrad_wgs_per_cu` workgroups per CU over the whole table -- few enough that the
        float atomics the M-split workgroups meet in stay far below the matrix time."""
        eng, l = self.eng, lib()
        self.wgrad_tabs = []
        for nine in (True, False):                     # nine-tap (stride-1 3x3) entries and one-tap entries: one launch each
            ents = [a for a in self.wgrad_entries if (a.kind == WG_3x3) == nine]
            if ents:
                self._wgrad_launch_table(launch, ents)

    def _wgrad_launch_table(self, launch, ents):
        eng, l = self.eng, lib()
        cost, groups = [], []
        for a in ents:
            nwc = 4 // a.nwn
            KP = -(-(a.nlr * (a.W + 2)) // 16) * 16 if a.kind == WG_3x3 else a.nlr
            # per chunk: (16-pixel steps x taps) matrix instructions per wave + a fixed staging / barrier share (~24 of them since the buffer-load staging; swept 8 / 24 / 48)
            cost.append(a.nchunks * ((KP // 16) * (9 if a.kind == WG_3x3 else 1) + int(getattr(eng.cfg.model, "wgrad_chunk_overhead", 24))))
            groups.append(-(-a.N // (32 * a.nwn)) * -(-a.C // (32 * nwc)))
        total = sum(c * g for c, g in zip(cost, groups))
        target = max(1, total // (256 * int(getattr(eng.cfg.model, "wgrad_wgs_per_cu", 2))))
        flops = 0
        for a, c in zip(ents, cost):
            a.grid_x = max(1, min(a.nchunks, -(-c // target)))
            flops += 2 * a.B * a.H * a.W * a.N * a.C * (9 if a.kind == WG_3x3 else 1)
        tab = (_WgradArgs * len(ents))(*ents)
        dev_tab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.dev)
        self.wgrad_tabs.append((tab, dev_tab))
        nwg = sum(a.grid_x * g for a, g in zip(ents, groups))
        launch(l.ctdd_unet_wgrad, dev_tab.data_ptr(), C.addressof(tab), len(ents), int(eng.precise),
               label=f"wgrad table: {len(ents)} entries ({'3x3' if ents[0].kind == WG_3x3 else '1x1 / stride-2 taps / bias'}), {nwg} workgroups", flops=flops)

    def finish(self, st, eng):
        """Lay out the backward plan (called by UNetEngine._build before its pools are resolved)."""
        l = lib()
        B, dev = self.B, self.dev
        self._grads, self.keep = {}, st.keep
        net = eng.net
        launch, conv, ptr = st.launch, st.conv, st.ptr
        fwd_plan = st.cur_lists["plan"]
        # ---- gradient arena: one flat fp32 buffer, a view per engine-owned parameter (torch layout)
        time_params = {id(p) for p in net.time.parameters()}
        for rb in self.resblocks:
            time_params |= {id(p) for p in rb.time.parameters()}
        self.engine_params = [p for p in net.parameters() if p.requires_grad and id(p) not in time_params]
        self.gflat = torch.zeros(sum(p.numel() for p in self.engine_params), dtype=torch.float32, device=dev)
        self.grad_view, off = {}, 0
        for p in self.engine_params:
            self.grad_view[id(p)] = (off, p.numel(), p.shape)
            off += p.numel()
        gptr = lambda p: self.gflat.data_ptr() + 4 * self.grad_view[id(p)][0]
        # ---- zero arena (fp32, cleared at the start of every backward): packed weight-gradient accumulators, d tproj,
        # GroupNorm sums, padded bias sums.  Sized by a pass over the records, then handed out by a bump allocator.
        Ntot = self.tproj.shape[1]
        need = sum(n + 4 for _, n in self.zero_parts) + B * Ntot + 64
        for kind, r in self.records:
            if kind == "gn":
                need += B * sum(s_.C for s_ in r["srcs"]) * 2 + 4
            elif kind == "conv":
                need += r["N"] * 8 + 4
        self.zbuf = torch.zeros(need, dtype=torch.float32, device=dev)
        zbase, zcur = self.zbuf.data_ptr(), 0

        def zalloc(n):
            nonlocal zcur
            o = zcur
            zcur += (n + 3) // 4 * 4
            assert zcur <= need
            return zbase + 4 * o
        gw_ptr = {id(part): zalloc(part[1]) for part in self.zero_parts}
        for e in self.entries:
            e["gw_ptr"] = gw_ptr[id(e["gw"])]
        dtproj_ptr = zalloc(B * Ntot)
        self.dtproj = self.zbuf[(dtproj_ptr - zbase) // 4:(dtproj_ptr - zbase) // 4 + B * Ntot].view(B, Ntot)

        # ---- backward plan
        bwd, bzero = [], []
        self.wgrad_entries, sum_jobs = [], []
        st.cur_lists["plan"], st.cur_lists["zero"] = bwd, bzero
        # seed: gradient of the network output (logits (B, D, S) fp32 or net_out [B*HW][2C] fp32) -> the mode's operand type
        last = self.records[-1][1]
        assert self.records[-1][0] == "conv" and last["out_f32"] is not None
        Nout, HW0 = last["N"], last["Hout"] * last["Wout"]
        ld0 = -(-Nout // 16) * 16
        self.seed_in = torch.zeros((B * HW0, Nout), dtype=torch.float32, device=dev)       # copy target of autograd's grad_output
        seed = _Grad(eng, B * HW0, Nout, dev, ld=ld0)
        seed.has = True
        launch(l.ctdd_unet_cast_rows, self.seed_in.data_ptr(), B * HW0, Nout, Nout, ld0, ptr(seed.hi), ptr(seed.f32), label="seed cast")
        out_grads = {id(last["out_f32"]): seed}
        self.keep.append(seed)

        def as_seg(g_, Cn):
            t = type("GradSeg", (), {})()
            t.hi, t.f32, t.C, t.stats = g_.hi, g_.f32, Cn, None
            return t

        tb_conv_of = {id(r["out"]): r for kind, r in self.records if kind == "conv" and r["tb"] is not None and r["out"] is not None}
        for kind, r in reversed(self.records):
            if kind == "conv":
                gy = self.g(r["out"]) if r["out"] is not None else out_grads[id(r["out_f32"])]
                N, Ho, Wo, Hi, Wi = r["N"], r["Hout"], r["Wout"], r["Hin"], r["Win"]
                ldy = gy.ld
                if any(e["ldd"] != ldy for e in r["per_seg"]):
                    raise native.CtddError(f"training plan: output gradient rows of {ldy} channels vs weights packed for {r['per_seg'][0]['ldd']} "
                                           "(channel counts must be multiples of 16)")
                gyT = as_seg(gy, ldy)
                # bias gradients (+ the per-sample time-projection gradient): per-(sample, channel) sums of the output gradient
                bps = r["bias_params"]
                bias_scr = None
                if (bps or r["tb"] is not None) and not r.get("sums_done"):
                    out_bn, stride = None, 0
                    if r["tb"] is not None:
                        out_bn, stride = dtproj_ptr + (r["tb"][0] - self.tproj.data_ptr()), r["tb"][1]
                    if out_bn is not None:            # (a time-projection gradient the GroupNorm backward could not give)
                        launch(l.ctdd_unet_colsum, ptr(gy.f32), ptr(gy.hi), B, Ho * Wo, -(-N // 8) * 8, ldy, out_bn, stride, None,
                               label="time-projection gradient")
                    if bps:
                        # sum over all pixels of gy: rides on the staging of the convolution's first weight-gradient entry
                        # (ctdd_wgrad_args.gb), copied to the parameters' gradients by the sums launch
                        bias_scr = zalloc(N)
                        for bp in bps:
                            sum_jobs.append((bias_scr, 1, 0, 1, N, gptr(bp), 0))
                # identity skip / residual
                if r["res"] is not None:
                    gr = self.g(r["res"])
                    launch(l.ctdd_unet_accumulate, ptr(gy.f32), ptr(gy.hi), ptr(gr.f32), ptr(gr.hi), gr.C * B * Ho * Wo, int(gr.has),
                           label="residual gradient")
                    gr.has = True
                for (src, cs, skind), e in zip(r["segs"], r["per_seg"]):
                    one = skind == SEG_1x1
                    self._wgrad_entry(src, gy, N, ldy, cs, e, skind, B, Ho, Wo, Ho if one else Hi, Wo if one else Wi, gb=bias_scr)
                    bias_scr = None
                    # data gradient of the segment: a convolution of gy with the flipped / transposed weights
                    gs = self.g(src)
                    packed = (None, e["dgrad"]) if eng.precise else (e["dgrad"], None)
                    gsT = as_seg(gs, cs)
                    resT = gsT if gs.has else None
                    if skind == SEG_3x3_S2:
                        conv([(gyT, ldy, SEG_3x3_S2T)], None, None, cs, Hi, Wi, Ho, Wo, gsT, res=resT, packed=packed, back=False)
                    else:
                        conv([(gyT, ldy, skind)], None, None, cs, Ho, Wo, Ho, Wo, gsT, res=resT, packed=packed, back=False)
                    gs.has = True
            elif kind == "gn":
                srcs, norm, out = r["srcs"], r["norm"], r["out"]
                ga = self.g(out)
                a = _GnBwdArgs()
                s1 = srcs[0]
                a.s1_f32, a.s1_bf16, a.C1 = ptr(s1.f32), ptr(s1.hi), s1.C
                r["stats_views"].append((a, s1.stats, "st1"))
                g1 = self.g(s1)
                a.d1_f32, a.d1_bf16, a.acc1 = ptr(g1.f32), ptr(g1.hi), int(g1.has)
                g1.has = True
                Ct = s1.C
                if len(srcs) == 2:
                    s2 = srcs[1]
                    a.s2_f32, a.s2_bf16, a.C2 = ptr(s2.f32), ptr(s2.hi), s2.C
                    r["stats_views"].append((a, s2.stats, "st2"))
                    g2 = self.g(s2)
                    a.d2_f32, a.d2_bf16, a.acc2 = ptr(g2.f32), ptr(g2.hi), int(g2.has)
                    g2.has = True
                    Ct += s2.C
                a.gamma, a.beta = norm.weight.data_ptr(), norm.bias.data_ptr()
                a.B, a.HW, a.G, a.eps, a.swish = B, r["HW"], norm.num_groups, r["eps"], int(r["swish"])
                a.da_f32, a.da_bf16 = ptr(ga.f32), ptr(ga.hi)
                a.drop_p, a.rng, a.layer = float(r["drop_p"]), self.rng.data_ptr(), r["layer"]
                sums = zalloc(B * Ct * 2)
                a.sums = sums
                cr = tb_conv_of.get(id(s1)) if len(srcs) == 1 else None
                if cr is not None and len(cr["bias_params"]) == 1:
                    # the GroupNorm input is conv1's output + bias + time projection and has no other consumer: sum_p dX per
                    # (sample, channel) IS the time-projection gradient, its sum over samples the bias gradient (closed form
                    # from the sums: no pass over the gradient tensor)
                    a.dsum_bn, a.dsum_stride = dtproj_ptr + (cr["tb"][0] - self.tproj.data_ptr()), cr["tb"][1]
                    a.dsum_n = gptr(cr["bias_params"][0])
                    cr["sums_done"] = True
                self.keep.append(a)
                launch(l.ctdd_unet_gn_bwd, C.byref(a), label=f"gn bwd C={Ct}")
                sum_jobs.append((sums, B, 2 * Ct, 2, Ct, gptr(norm.bias), 0))
                sum_jobs.append((sums + 4, B, 2 * Ct, 2, Ct, gptr(norm.weight), 0))
            elif kind == "attn":
                gao = self.g(r["ao"])
                T, Cx = r["T"], r["Cx"]
                dq = _Grad(eng, B * T, 3 * Cx, dev)
                dq.has = True
                a = _AttnBwdArgs()
                a.qkv, a.d_out_f32, a.d_out_bf16 = r["qkv"].data_ptr(), ptr(gao.f32), ptr(gao.hi)
                a.B, a.T, a.C, a.heads = B, T, Cx, r["att"].num_heads
                a.d_qkv, a.d_qkv_bf16 = ptr(dq.f32), ptr(dq.hi)
                self.keep.extend([a, dq])
                launch(l.ctdd_unet_attention_bwd, C.byref(a), label="attention bwd")
                out_grads[id(r["qkv"])] = dq
            elif kind == "up":
                cur, up = r["cur"], r["up"]
                gu, gc = self.g(up), self.g(cur)
                launch(l.ctdd_unet_downsum2x, ptr(gu.f32), ptr(gu.hi), B, cur.H, cur.W, cur.C, ptr(gc.f32), ptr(gc.hi), int(gc.has),
                       label="upsample bwd")
                gc.has = True
            elif kind == "first":
                c0, fa, out = r["conv"], r["fa"], r["out"]
                go = self.g(out)
                a = _FirstWgradArgs()
                a.x64, a.x32, a.lo, a.hi = fa.x64, fa.x32, fa.lo, fa.hi
                a.dy_f32, a.dy_bf16 = ptr(go.f32), ptr(go.hi)
                a.B, a.Cin, a.H, a.W, a.Cout = fa.B, fa.Cin, fa.H, fa.W, fa.Cout
                a.gw, a.gbias = gptr(c0.weight), gptr(c0.bias)
                scratch = torch.zeros(int(l.ctdd_unet_first_conv_wgrad_scratch(fa.B, fa.H, fa.Cin, fa.Cout)), dtype=torch.float32, device=dev)
                a.partial = scratch.data_ptr()
                self.keep.extend([a, scratch])
                launch(l.ctdd_unet_first_conv_wgrad, C.byref(a), label="first conv wgrad")

        # ---- split-K partial-sum buffers of the backward convolutions: their own pool, zeroed with the arena
        nz = sum(n for _, n in bzero)
        self.bzpool = torch.zeros(max(nz, 1), dtype=torch.float32, device=dev)
        zo = 0
        for a_, n in bzero:
            a_.acc_buf = self.bzpool.data_ptr() + 4 * zo
            zo += n
        # ---- pack table: one entry per (convolution, K-segment)
        tab = (_PackEntry * len(self.entries))()
        first = 0
        for i, e in enumerate(self.entries):
            t = tab[i]
            t.w, t.fwd, t.dgrad, t.gw, t.grad = e["w"].data_ptr(), e["fwd"].data_ptr(), e["dgrad"].data_ptr(), e["gw_ptr"], gptr(e["w"])
            t.N, t.Cin_tot, t.c_off, t.C, t.ntap, t.Ktot, t.koff, t.flip, t.ldd = (e["N"], e["Cin_tot"], e["c_off"], e["C"], e["ntap"],
                                                                                      e["Ktot"], e["koff"], e["flip"], e["ldd"])
            t.first = first
            first += e["N"] * e["C"] * e["ntap"]
        self.pack_total = first
        self.pack_tab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
        self._wgrad_table_launch(launch)               # every convolution's weight (and bias) gradient: one launch
        self.bwd_sum_tab, nj = self._sum_jobs_table(sum_jobs, dev)
        launch(l.ctdd_unet_sum_jobs, self.bwd_sum_tab.data_ptr(), nj, label=f"small sums: {len(sum_jobs)} reductions")
        launch(l.ctdd_unet_unpack_grads, self.pack_tab.data_ptr(), len(self.entries), self.pack_total, label="unpack gradients")
        # ---- forward prologue: ONE pack launch for every convolution weight (+ dropout step bump), folded biases
        pro = []
        st.cur_lists["plan"] = pro
        launch(l.ctdd_unet_pack_weights, self.pack_tab.data_ptr(), len(self.entries), self.pack_total, int(eng.precise),
               self.rng.data_ptr() if self.dropout else None, label="pack weights")
        if self.bias_jobs:                             # folded biases b = sum of parameters: the "batch" walks the parameters
            jobs = []
            for buf, ps in self.bias_jobs:
                assert len(ps) == 2 and (ps[1].data_ptr() - ps[0].data_ptr()) % 4 == 0
                jobs.append((ps[0].data_ptr(), 2, (ps[1].data_ptr() - ps[0].data_ptr()) // 4, 1, ps[0].numel(), buf.data_ptr(), 0))
            self.fwd_sum_tab, nj = self._sum_jobs_table(jobs, dev)
            launch(l.ctdd_unet_sum_jobs, self.fwd_sum_tab.data_ptr(), nj, label="folded biases")
        fwd_plan[:0] = pro
        st.cur_lists["plan"], st.cur_lists["zero"] = fwd_plan, st.zero_views_fwd
        st.bwd_plan, st.tc = bwd, self


# ---------------------------------------------------------------------- autograd binding
class UNetTrainFn(torch.autograd.Function):
    """logits (or the logistic head's input) = U-Net(x, t) on the training plan; backward = the backward plan.
    Inputs after `tproj` are the engine-owned parameters: their gradients are views of the plan's gradient arena."""

    @staticmethod
    def forward(ctx, eng, st, x, times, tproj, *params):
        out = eng._train_run_forward(st, x, times, tproj)
        ctx.eng, ctx.st, ctx.gen = eng, st, st.gen
        return out

    @staticmethod
    def backward(ctx, dout):
        d_tproj, grads = ctx.eng._train_run_backward(ctx.st, ctx.gen, dout)
        return (None, None, None, None, d_tproj, *grads)
