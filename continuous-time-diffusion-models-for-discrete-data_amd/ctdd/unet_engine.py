"""Hand-written HIP inference engine for the tauLDR U-Net (lib/networks/unet.py; reference
TAUnSDDM/lib/networks/unet.py:303-459 + lib/models/models.py:225-292).

The engine walks the module tree of a built model once, packs every weight into the layouts the
kernels of csrc/unet_kernels.hip stream (implicit-GEMM [N][K] bf16 hi/lo, K = segment -> tap ->
channel), allocates all intermediate NHWC tensors for a given batch size, and records the forward
as a flat list of pre-bound launches.  The list is captured into a HIP graph (torch.cuda.CUDAGraph
is the capture API; every node is one of our kernels) and replayed per call, so a sampler step costs
one graph launch instead of ~150 Python-side launches.

precision = "bf16":  bf16 activations / weights on v_mfma_f32_32x32x16_bf16, fp32 accumulate (the
                     BASELINE config's dtype; throughput path)
precision = "fp32":  fp32 activations / weights on the exact-fp32 v_mfma_f32_32x32x2_f32 -- the path
                     the 1e-4 logit parity test runs.
"""
import ctypes as C
import math

import torch

from . import native

_P, _I, _F = C.c_void_p, C.c_int, C.c_float
SEG_3x3, SEG_1x1, SEG_3x3_S2, SEG_3x3_UP = 0, 1, 2, 3


class _Seg(C.Structure):
    _fields_ = [("hi", _P), ("f32", _P), ("C", _I), ("kind", _I)]


class _ConvArgs(C.Structure):
    _fields_ = [("seg", _Seg * 3), ("nseg", _I), ("w_hi", _P), ("w_f32", _P), ("B", _I), ("H", _I), ("W", _I),
                ("Hin", _I), ("Win", _I), ("N", _I), ("Ktot", _I), ("bias", _P), ("tbias", _P), ("tb_stride", _I),
                ("res_f32", _P), ("res_bf16", _P), ("out_f32", _P), ("out_hi", _P), ("stats", _P), ("logits_C", _I),
                ("ksplit", _I), ("acc_buf", _P), ("act", _I), ("out_lo", _P)]


class _FirstArgs(C.Structure):
    _fields_ = [("x64", _P), ("x32", _P), ("lo", _F), ("hi", _F), ("w", _P), ("bias", _P), ("B", _I), ("Cin", _I),
                ("H", _I), ("W", _I), ("Cout", _I), ("out_f32", _P), ("out_hi", _P), ("stats", _P), ("x0_f32", _P)]


class _GnArgs(C.Structure):
    _fields_ = [("s1_f32", _P), ("s1_bf16", _P), ("st1", _P), ("C1", _I), ("s2_f32", _P), ("s2_bf16", _P), ("st2", _P),
                ("C2", _I), ("gamma", _P), ("beta", _P), ("B", _I), ("HW", _I), ("G", _I), ("eps", _F), ("swish", _I),
                ("out_hi", _P), ("out_f32", _P)]


class _TimeArgs(C.Structure):
    _fields_ = [("t", _P), ("B", _I), ("ch", _I), ("tdim", _I), ("w1", _P), ("b1", _P), ("w2", _P), ("b2", _P),
                ("hid", _P), ("act", _P)]


class _AttnArgs(C.Structure):
    _fields_ = [("qkv", _P), ("B", _I), ("T", _I), ("C", _I), ("heads", _I), ("out_hi", _P), ("out_f32", _P)]


class _LogisticArgs(C.Structure):
    _fields_ = [("net", _P), ("x0", _P), ("B", _I), ("C", _I), ("HW", _I), ("S", _I), ("fix", _I), ("out", _P), ("fast", _I), ("out_bf16", _P)]


_sigs_done = False


def _lib():
    global _sigs_done
    lib = native.load()
    if not _sigs_done:
        for name, argt in (("ctdd_unet_conv", [_P, _I, _I, _I, _P]), ("ctdd_unet_conv_patch", [_P, _I, _I, _I, _P]), ("ctdd_unet_conv_res", [_P, _I, _P]), ("ctdd_unet_conv_ring", [_P, _I, _P]),
                           ("ctdd_unet_upsample2x", [_P, _I, _I, _I, _I, _P, _P]), ("ctdd_unet_first_conv", [_P, _P]),
                           ("ctdd_unet_gn_apply", [_P, _P]), ("ctdd_unet_gn_onepass", [_P, _I, _I, _P]), ("ctdd_unet_channel_stats", [_P, _I, _I, _I, _P, _P]),
                           ("ctdd_unet_time", [_P, _P, _P, _I, _P, _P]), ("ctdd_unet_time_uniform", [_P, _P, _P, _I, _P, _P]),
                           ("ctdd_unet_attention", [_P, _P]),
                           ("ctdd_unet_logistic_head", [_P, _P])):
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = argt, _I
        _sigs_done = True
    return lib


def _unwrap(net):
    """cfg.distributed wraps the network in DistributedDataParallel (models.py:104-107); the engine reads the module behind it."""
    return net.module if hasattr(net, "module") and net.__class__.__name__ == "DistributedDataParallel" else net


def supports(model):
    """The engine covers the reference's image U-Net wrapper without padding."""
    net = _unwrap(getattr(model, "net", None))
    return (net is not None and net.__class__.__name__ == "UNet" and not model.padding
            and model.cfg.model.model_output in ("logits", "logistic_pars"))


def training_supported(model):
    """Whether the hand-written training plan (forward that keeps what backward needs + HIP backward) covers this model:
    the same U-Nets as the inference engine with channel counts in multiples of 16."""
    if getattr(model.cfg.model, "engine_train", "hip") != "hip" or not supports(model):
        return False
    net = _unwrap(model.net)
    return net.channel % 16 == 0 and all((net.channel * int(m_)) % 16 == 0 for m_ in model.cfg.model.ch_mult)


class _GnUncovered(Exception):
    """A GroupNorm shape outside k_gn_onepass (UNetEngine._build falls back to statistics epilogues + k_gn_apply)."""


def _onepass_slab(B, HW, Cn, G, max_threads=1024):
    """Channels per workgroup ctdd_unet_gn_onepass will choose (csrc/unet_kernels.hip), 0 when no slab of whole groups fits."""
    if Cn % 8 or G <= 0 or Cn % G:
        return 0
    cg = Cn // G
    L = cg
    while L % 8:
        L += cg
    best = 0
    for sc in range(L, Cn + 1, L):
        if Cn % sc:
            continue
        noct = sc // 8
        if noct > max_threads:
            continue
        npl = min(max_threads // noct, HW)
        if -(-HW // npl) > 12:
            continue
        wgs = B * (Cn // sc)
        if best == 0 or wgs >= 256:
            best = sc
        if wgs < 256:
            break
    return best


class _Tensor:
    """NHWC activation [B*H*W][C]: bf16 (`hi`) in bf16 mode, fp32 (`f32`) in fp32 mode, plus the offset
    of its per-(b, channel) statistics in the plan's pool."""

    def __init__(self, eng, B, H, W, Cn, stats=True):
        dev, M = eng.dev, B * H * W
        self.B, self.H, self.W, self.C = B, H, W, Cn
        self.f32 = torch.empty((M, Cn), dtype=torch.float32, device=dev) if eng.precise else None
        self.hi = None if eng.precise else torch.empty((M, Cn), dtype=torch.bfloat16, device=dev)
        small = H * W <= getattr(eng, "_plan_no_stats_hw", 0)      # a one-pass GroupNorm level: no statistics from the producers
        self.stats_by_gn = bool(stats and small and getattr(eng, "_plan_gn_writes_stats", False))   # training: k_gn_onepass leaves them for backward
        self.stats = eng.alloc_stats(B * Cn * 2) if (stats and (not small or self.stats_by_gn)) else None
        eng._live.append(self)           # raw pointers are baked into the plan: keep every buffer alive


class UNetEngine:
    def __init__(self, model, precision=None):
        self.model = model
        self.net = _unwrap(model.net)
        self.cfg = model.cfg
        self.dev = next(self.net.parameters()).device
        if self.dev.type != "cuda":
            raise native.CtddError("UNetEngine needs the model on a GPU")
        self.precision = precision or getattr(self.cfg.model, "engine_precision", "bf16")
        if self.precision not in ("bf16", "fp32"):
            raise ValueError(f"unknown engine precision {self.precision}")
        self.precise = self.precision == "fp32"
        self._plans = {}
        self._wver = None
        self._packed = None

    # ------------------------------------------------------------------ weights
    def _weights_version(self):
        return sum(p._version for p in self.net.parameters()) + 7919 * getattr(self.model, "_weights_version", 0)

    def _pack(self, w2d):
        """[N][K] weights in the mode's element type: (bf16 | None, fp32 | None)."""
        w = w2d.detach().float().contiguous()
        return (None, w) if self.precise else (w.to(torch.bfloat16).contiguous(), None)

    @staticmethod
    def _w2d(wsrc, segs):
        """The [N][K] matrix of a convolution, K = segment -> tap -> channel, from per-segment (parameter, channel offset)."""
        parts = []
        for (w, c_off), (_, cs, kind) in zip(wsrc, segs):
            w = w.detach().float()
            N = w.shape[0]
            if kind == SEG_1x1:
                parts.append(w.reshape(N, -1)[:, c_off:c_off + cs])
            else:
                parts.append(w[:, c_off:c_off + cs].permute(0, 2, 3, 1).reshape(N, 9 * cs))
        return torch.cat(parts, dim=1).contiguous()

    @staticmethod
    def _conv_w(weight, splits):
        """torch conv weight [N][Cin][3][3] -> [N][K], K = segment -> tap -> channel."""
        parts, c0 = [], 0
        for cs in splits:
            parts.append(weight[:, c0:c0 + cs].permute(0, 2, 3, 1).reshape(weight.shape[0], 9 * cs))
            c0 += cs
        return torch.cat(parts, dim=1)

    # ------------------------------------------------------------------ plan construction
    def alloc_stats(self, n):
        off = self._stats_off
        self._stats_off += n
        return off

    def _build(self, B, x_dtype, logits_out=None, tc=None, logits_bf16=False, uniform_t=False):
        # bf16 inference plans: GroupNorm as one pass per tensor with the statistics inside (k_gn_onepass), no statistics in
        # the convolution epilogues (cfg.model.gn_onepass, default on); a net with a GroupNorm the kernel does not cover is
        # rebuilt the old way (statistics by the producers, k_gn_apply)
        # (training plans too, cfg.model.gn_onepass_train: the kernel then also writes each source's per-channel sums into the
        #  tensors' statistics buffers, which the GroupNorm backward reads)
        if (not self.precise) and int(getattr(self.cfg.model, "gn_onepass", 1)) and (tc is None or int(getattr(self.cfg.model, "gn_onepass_train", 1))):
            try:
                return self._build_impl(B, x_dtype, logits_out, tc, logits_bf16, uniform_t, onepass_gn=True)
            except _GnUncovered:
                pass
        return self._build_impl(B, x_dtype, logits_out, tc, logits_bf16, uniform_t, onepass_gn=False)

    def _build_impl(self, B, x_dtype, logits_out=None, tc=None, logits_bf16=False, uniform_t=False, onepass_gn=False):
        net, m = self.net, self.cfg.model
        # (levels up to gn_onepass_max_hw pixels per sample: above it the one-workgroup-per-(sample, slab) kernel loses to the many
        #  small workgroups of k_gn_apply, and those tensors keep their statistics epilogues.  MNIST net, batch 256, sampler loop:
        #  off 85.1 k sample-steps/s, 7x7 only 85.8 k, 7x7 + 14x14 87.9 k, all levels 85.1 k)
        self._plan_no_stats_hw = int(getattr(m, "gn_onepass_max_hw", 256)) if onepass_gn else 0
        self._plan_gn_writes_stats = bool(onepass_gn and tc is not None)
        gn_threads = int(getattr(m, "gn_threads", 512))    # (measured in the two-chain sampler loop: 384-512 best, 1024 -1 %)
        lib = _lib()
        dev = self.dev
        Cin, H0, W0 = self.cfg.data.shape
        ch, S = net.channel, net.S
        plan, keep = [], []
        self._stats_off = 0
        self._live = keep
        st = type("Plan", (), {})()
        st.B = B
        st.x_in = torch.zeros((B, Cin, H0, W0), dtype=x_dtype, device=dev)
        st.t_in = torch.zeros((B,), dtype=torch.float32, device=dev)
        stream = lambda: torch.cuda.current_stream().cuda_stream

        def ptr(t):
            return None if t is None else t.data_ptr()

        def launch(fn, *args, label=None, flops=0):
            def run():
                rc = fn(*args, stream())
                if rc != 0:
                    raise native.CtddError(f"{fn.__name__} failed ({rc}): {lib.ctdd_last_error().decode()}")
            run.label = (fn.__name__, label)
            run.flops = flops                      # matrix FLOPs of the launch (bench.py's network roofline)
            cur_lists["plan"].append(run)

        bks = (32, 16) if self.precise else (96, 64, 32, 16)      # fp32 tiles: K = 32 keeps 4 workgroups per CU

        def pick_bk(cs):
            for bk in bks:
                if all(c % bk == 0 for c in cs):
                    return bk
            raise native.CtddError(f"no K tile divides channel counts {cs}")

        def pick_bnt(N, bk):
            if bk == 16:
                return 1
            if N % 96 == 0 and bk in (96, 32):
                return 3
            if N % 128 == 0:
                return 4
            if N % 64 == 0 and bk == 64:
                return 2
            return 1

        stats_views = []          # (tensor, offset) resolved after the pool exists
        zero_views = []           # split-K partial-sum buffers: (conv args, elements)
        cur_lists = {"plan": plan, "zero": zero_views}     # the training context points these at the backward plan

        def conv(segs, wsrc, bias, N, Hout, Wout, Hin, Win, out, tb=None, res=None, logits_C=0, out_f32_tensor=None, bias_params=None,
                 packed=None, back=True):
            """segs: list of (_Tensor, channels, kind); wsrc: per segment (weight parameter [N][Cin_tot][k][k], channel offset) --
            the [N][K] matrix the kernels stream is K = segment -> tap -> channel of those slices.  bias_params: the
            parameters whose sum `bias` is (training: each receives the bias gradient).  packed: (bf16 | None, fp32 | None)
            ready-made weights (the data-gradient convolutions of the training plan)."""
            a = _ConvArgs()
            a.nseg = len(segs)
            for i, (src, cs, kind) in enumerate(segs):
                a.seg[i].hi, a.seg[i].f32, a.seg[i].C, a.seg[i].kind = ptr(src.hi), ptr(src.f32), cs, kind
            Ktot = sum(cs * (1 if kind == SEG_1x1 else 9) for _, cs, kind in segs)
            if packed is not None:
                whi, wf = packed
            elif tc is not None:
                whi, wf = tc.packed_forward(wsrc, segs, N, Ktot)     # persistent buffers, refreshed by ONE pack launch per step
            else:
                w2d = self._w2d(wsrc, segs)
                assert w2d.shape[1] == Ktot
                whi, wf = self._pack(w2d)
            keep.extend([whi, wf, bias])
            a.w_hi, a.w_f32 = ptr(whi), ptr(wf)
            a.B, a.H, a.W, a.Hin, a.Win, a.N, a.Ktot = B, Hout, Wout, Hin, Win, N, Ktot
            a.bias = ptr(bias)
            if tb is not None:
                a.tbias, a.tb_stride = tb
            if res is not None:
                if self.precise:
                    a.res_f32 = ptr(res.f32)
                else:
                    a.res_bf16 = ptr(res.hi)
            if out_f32_tensor is not None:
                if out_f32_tensor.dtype == torch.bfloat16:      # (the bf16 logits of the sampler loops)
                    a.out_hi = ptr(out_f32_tensor)
                else:
                    a.out_f32 = ptr(out_f32_tensor)
            elif out is not None:
                a.out_f32, a.out_hi = ptr(out.f32), ptr(out.hi)
                if out.stats is not None and not out.stats_by_gn:
                    stats_views.append((a, out.stats))
            a.logits_C = logits_C
            keep.append(a)
            cs = [s[1] for s in segs]
            patchable = (not self.precise) and all(s[2] in (SEG_3x3, SEG_1x1) for s in segs) and Wout <= 33
            hw_ = Hout * Wout
            patchable = patchable and N % 8 == 0 and (hw_ >= 32 or hw_ == 16 or B == 1) and (logits_C == 0 or (N // logits_C) % 8 == 0)
            M_ = B * Hout * Wout
            lab = f"{Hout}x{Wout} K={Ktot} N={N} segs={[(c_, k_) for _, c_, k_ in segs]}"
            if tc is not None and back:
                tc.record_conv(conv, segs, wsrc, bias_params, N, Hout, Wout, Hin, Win, out, tb, res, logits_C, out_f32_tensor)
            which = getattr(m, "conv_kernel", "auto")
            only3 = all(s[2] == SEG_3x3 for s in segs)
            if which == "auto":
                # measured at batch 256 (MNIST net): the LDS-DMA ring wins where 512-pixel tiles give >= 160
                # workgroups and every unit has nine taps (28x28, 14x14); the patch kernel (128/256-pixel tiles, two
                # workgroups per CU) elsewhere.  Split-K lost everywhere it was tried (fp32 atomics + finish pass).
                ring_min = int(getattr(m, "ring_min_tiles", 80))
                which = "ring" if (only3 and -(-M_ // 512) * -(-N // 96) >= ring_min and N % 96 == 0) else "patch"
                # (CIFAR net, N = 256 at 16x16: 64-column ring tiles beat the 128-column patch tiles, 49 vs 54 / 69 vs 104 us at batch 128)
                if which == "patch" and only3 and N % 64 == 0 and N % 96 != 0 and Hout * Wout <= 256 and -(-M_ // 512) * (N // 64) >= ring_min:
                    which = "ring"
                # (MNIST net's S = 256 output convolution, 28x28 K = 864: 64-column ring tiles 88 us, the 128-column patch tiles 105 us,
                #  128-column ring tiles 179 us at batch 128; +1.4 % on the sampler loop)
                if which == "patch" and only3 and N % 64 == 0 and N % 96 != 0 and N >= 256 and -(-M_ // 512) * (N // 64) >= ring_min:
                    which = "ring"
            resident = which == "res" and patchable and all(c % 32 == 0 for c in cs) and N % 32 == 0
            ring = which == "ring" and patchable and all(c % 16 == 0 for c in cs) and N % 32 == 0
            if resident or ring:
                # 512-pixel tiles, all nine taps' weights in LDS: register-staged 32-channel units (k_conv_res)
                # or an LDS-DMA ring of 16-channel units (k_conv_ring); csrc/unet_kernels.hip
                bnt = 3 if N % 96 == 0 else 4 if (N % 128 == 0 and resident) else 2
                ntiles = -(-M_ // 512) * -(-N // (32 * bnt))
                units = sum(c // (32 if resident else 16) for c in cs)
                if getattr(m, "conv_ksplit", 1) > 1 and units >= 2 and logits_C == 0:
                    a.ksplit = min(units, int(m.conv_ksplit))
                if a.ksplit > 1:
                    cur_lists["zero"].append((a, M_ * N))
                fn = lib.ctdd_unet_conv_res if resident else lib.ctdd_unet_conv_ring
                # (ring_small_tiles: 256-pixel tiles, four waves, two workgroups per CU -- the variant two concurrent chains can share a CU with)
                sel = bnt + 10 if (ring and bnt in (2, 3) and getattr(m, "ring_small_tiles", 0)) else bnt
                launch(fn, C.byref(a), sel, label=lab + f" {which} bnt={sel} ks={a.ksplit}", flops=2 * M_ * N * Ktot)
            elif patchable:
                # throughput kernel: slab staged once per channel chunk (csrc/unet_kernels.hip: k_conv_patch)
                small = -(-M_ // 128) * -(-N // 96) < 256            # too few 128 x 96 tiles to fill the chip: 32-column tiles
                if small and all(c % 64 == 0 for c in cs) and N % 32 == 0:
                    bk, bnt = 64, 1
                elif small and all(c % 48 == 0 for c in cs) and N % 32 == 0:
                    bk, bnt = 48, 1
                elif all(c % 48 == 0 for c in cs) and (N % 96 == 0 or N % 128 == 0):
                    bk, bnt = 48, (3 if N % 96 == 0 else 4)
                elif all(c % 64 == 0 for c in cs) and N % 64 == 0:
                    bk, bnt = 64, (4 if N % 128 == 0 else 2)
                elif all(c % 32 == 0 for c in cs):
                    bk, bnt = 32, (3 if N % 96 == 0 else 4 if N % 128 == 0 else 1)
                else:
                    bk, bnt = 16, 1
                wm = 64 if (bnt >= 2 and bk in (48, 64) and M_ >= int(getattr(m, "patch_wm64_min_rows", 196 * 128)) and (bk, bnt) != (64, 4)) else 32
                units = sum(c // bk for c in cs)
                nwg = -(-M_ // (4 * wm)) * -(-N // (32 * bnt))
                if getattr(m, "conv_ksplit", 1) > 1 and units >= 2 and logits_C == 0 and out_f32_tensor is None:
                    a.ksplit = min(units, int(m.conv_ksplit))
                elif nwg <= int(getattr(m, "ksplit_max_wgs", 96)) and units >= 4 and logits_C == 0 and out_f32_tensor is None:
                    a.ksplit = max(1, min(units // 2, 256 // nwg))       # tiny grids (4x4 levels): split K to fill the chip
                if a.ksplit > 1:
                    cur_lists["zero"].append((a, M_ * N))
                launch(lib.ctdd_unet_conv_patch, C.byref(a), bk, bnt, wm, label=lab + f" patch bk={bk} bnt={bnt} wm={wm} ks={a.ksplit}",
                       flops=2 * M_ * N * Ktot)
            else:
                bk = pick_bk(cs)
                bnt = pick_bnt(N, bk)
                if (not self.precise) and -(-M_ // 128) * -(-N // (32 * bnt)) < int(getattr(m, "igemm_small_wgs", 256)) and bk in (96, 64, 32):
                    bnt = 1                                        # tiny grids: 32-column tiles, more workgroups
                launch(lib.ctdd_unet_conv, C.byref(a), bk, bnt, int(self.precise), label=lab + f" igemm bk={bk} bnt={bnt}", flops=2 * M_ * N * Ktot)

        def gn_apply(srcs, norm, swish, eps, HW, drop_p=0.0):
            """srcs: one or two _Tensor; returns activated planes tensor (training: dropout applied in place after it)."""
            Ct = sum(s.C for s in srcs)
            out = _Tensor(self, B, srcs[0].H, srcs[0].W, Ct, stats=False)
            a = _GnArgs()
            s1 = srcs[0]
            a.s1_f32, a.s1_bf16, a.C1 = (ptr(s1.f32), None, s1.C) if self.precise else (None, ptr(s1.hi), s1.C)
            if s1.stats is not None:
                stats_views.append((a, s1.stats, "st1"))
            if len(srcs) == 2:
                s2 = srcs[1]
                a.s2_f32, a.s2_bf16, a.C2 = (ptr(s2.f32), None, s2.C) if self.precise else (None, ptr(s2.hi), s2.C)
                if s2.stats is not None:
                    stats_views.append((a, s2.stats, "st2"))
            g, b_ = norm.weight.detach().float().contiguous(), norm.bias.detach().float().contiguous()
            keep.extend([g, b_, a, out])
            a.gamma, a.beta = ptr(g), ptr(b_)
            a.B, a.HW, a.G, a.eps, a.swish = B, HW, norm.num_groups, eps, int(swish)
            a.out_hi, a.out_f32 = ptr(out.hi), ptr(out.f32)
            if onepass_gn and HW > self._plan_no_stats_hw:
                launch(lib.ctdd_unet_gn_apply, C.byref(a), label=f"gn {srcs[0].H}x{srcs[0].W} C={Ct} ({len(srcs)} src)")
            elif onepass_gn and _onepass_slab(B, HW, Ct, norm.num_groups, gn_threads) > 0:
                # inference, bf16: statistics + normalisation in one pass over the tensor (k_gn_onepass); the producers' epilogues
                # then carry no statistics at all (their tensors were created without statistics buffers)
                launch(lib.ctdd_unet_gn_onepass, C.byref(a), 0, gn_threads, label=f"gn1 {srcs[0].H}x{srcs[0].W} C={Ct} ({len(srcs)} src)")
            else:
                if onepass_gn:
                    raise _GnUncovered(f"one-pass GroupNorm does not cover HW={HW} C={Ct} G={norm.num_groups}")
                launch(lib.ctdd_unet_gn_apply, C.byref(a), label=f"gn {srcs[0].H}x{srcs[0].W} C={Ct} ({len(srcs)} src)")
            if tc is not None:
                tc.record_gn(srcs, norm, swish, eps, HW, out, drop_p, launch, stats_views)
            return out

        # ---- time embedding + all ResBlock projections in two launches
        resblocks = [mod.resblocks for mod in list(net.down) + list(net.mid) + list(net.up) if hasattr(mod, "resblocks")]
        tdim = ch * 4
        pw = torch.cat([rb.time[1].weight.detach().float() for rb in resblocks], 0).t().contiguous()   # [tdim][Ntot]
        pb = torch.cat([rb.time[1].bias.detach().float() for rb in resblocks], 0).contiguous()
        Ntot = pw.shape[1]
        st.tact = torch.empty((B, tdim), dtype=torch.float32, device=dev)
        time_row = uniform_t == "row" and tc is None
        uniform_t = bool(uniform_t) and tc is None
        # uniform_t: every sample at the same time (the samplers): ONE projection row from one fused launch, read by the
        # convolutions with a zero batch stride (csrc/unet_kernels.hip: k_time_uniform).  "row": that row comes from the caller
        # (a sampler's grid of times is known when it starts: time_table() computes every step's row at once) -- no time
        # launch in the plan at all
        st.time_row = time_row
        st.tproj = torch.empty((1 if uniform_t else B, Ntot), dtype=torch.float32, device=dev)
        tb_stride = 0 if uniform_t else Ntot
        ta = _TimeArgs()
        tw = [net.time[1].weight.t(), net.time[1].bias, net.time[3].weight.t(), net.time[3].bias]      # weights as [in][out]
        tw = [w.detach().float().contiguous() for w in tw]
        st.thid = torch.empty((B, tdim), dtype=torch.float32, device=dev)
        ta.t, ta.B, ta.ch, ta.tdim = ptr(st.t_in), B, ch, tdim
        ta.w1, ta.b1, ta.w2, ta.b2, ta.hid, ta.act = ptr(tw[0]), ptr(tw[1]), ptr(tw[2]), ptr(tw[3]), ptr(st.thid), ptr(st.tact)
        keep.extend(tw + [pw, pb, ta])
        if tc is None and not time_row:
            launch(lib.ctdd_unet_time_uniform if uniform_t else lib.ctdd_unet_time, C.byref(ta), ptr(pw), ptr(pb), Ntot, ptr(st.tproj))
        elif tc is None:
            st.tproj.zero_()
        else:
            tc.tproj, tc.resblocks = st.tproj, resblocks          # filled by the caller before the plan runs
        toff = {}
        o = 0
        for rb in resblocks:
            toff[id(rb)] = o
            o += rb.time[1].weight.shape[0]

        # ---- first conv
        c0 = net.down[0]
        cur = _Tensor(self, B, H0, W0, ch)
        fa = _FirstArgs()
        if x_dtype == torch.int64:
            fa.x64 = ptr(st.x_in)
        else:
            fa.x32 = ptr(st.x_in)
        fa.lo, fa.hi = float(net.x_min_max[0]), float(net.x_min_max[1])
        w0, b0 = c0.weight.detach().float().contiguous(), c0.bias.detach().float().contiguous()
        fa.w, fa.bias, fa.B, fa.Cin, fa.H, fa.W, fa.Cout = ptr(w0), ptr(b0), B, Cin, H0, W0, ch
        fa.out_f32, fa.out_hi = ptr(cur.f32), ptr(cur.hi)
        logistic = m.model_output == "logistic_pars"
        if logistic:
            st.x0 = torch.empty((B, Cin, H0, W0), dtype=torch.float32, device=dev)
            fa.x0_f32 = ptr(st.x0)
        if cur.stats is not None:
            stats_views.append((fa, cur.stats))
        keep.extend([w0, b0, fa])
        launch(lib.ctdd_unet_first_conv, C.byref(fa))
        if tc is not None:
            tc.record_first(c0, fa, cur)

        def resblock(rb, srcs):
            """srcs: list of 1-2 tensors forming the (virtual) channel concatenation."""
            Hc, Wc = srcs[0].H, srcs[0].W
            cs = [s.C for s in srcs]
            cout = rb.conv1.weight.shape[0]
            a1 = gn_apply(srcs, rb.norm1, True, rb.norm1.eps, Hc * Wc)
            h = _Tensor(self, B, Hc, Wc, cout)
            b1 = rb.conv1.bias.detach().float().contiguous()
            conv([(a1, a1.C, SEG_3x3)], [(rb.conv1.weight, 0)], b1, cout, Hc, Wc,
                 Hc, Wc, h, tb=(st.tproj.data_ptr() + 4 * toff[id(rb)], tb_stride), bias_params=[rb.conv1.bias])
            drop = float(rb.dropout.p) if (tc is not None and tc.dropout) else 0.0
            a2 = gn_apply([h], rb.norm2, True, rb.norm2.eps, Hc * Wc, drop_p=drop)
            y = _Tensor(self, B, Hc, Wc, cout)
            wsrc = [(rb.conv2.weight, 0)]
            bias2, bias_params = rb.conv2.bias.detach().float(), [rb.conv2.bias]
            segs = [(a2, cout, SEG_3x3)]
            res = None
            if rb.skip is not None:
                c_ = 0
                for s_ in srcs:                       # linear skip folded in as 1x1 K-segments on the raw input
                    segs.append((s_, s_.C, SEG_1x1))
                    wsrc.append((rb.skip.weight, c_))
                    c_ += s_.C
                bias_params.append(rb.skip.bias)
                bias2 = tc.summed_bias(bias_params) if tc is not None else bias2 + rb.skip.bias.detach().float()
            else:
                res = srcs[0]
            conv(segs, wsrc, bias2.contiguous(), cout, Hc, Wc, Hc, Wc, y, res=res, bias_params=bias_params)
            return y

        def attention(att, x):
            T = x.H * x.W
            an = gn_apply([x], att.norm, False, att.norm.eps, T)
            Cx = x.C
            qkv = torch.empty((B * T, 3 * Cx), dtype=torch.float32, device=dev)
            keep.append(qkv)
            if tc is not None:
                tc.begin_attention(att, x, qkv)
            conv([(an, Cx, SEG_1x1)], [(att.qkv.weight, 0)],
                 att.qkv.bias.detach().float().contiguous(), 3 * Cx, x.H, x.W, x.H, x.W, None, out_f32_tensor=qkv,
                 bias_params=[att.qkv.bias])
            ao = _Tensor(self, B, x.H, x.W, Cx, stats=False)
            aa = _AttnArgs()
            aa.qkv, aa.B, aa.T, aa.C, aa.heads, aa.out_hi, aa.out_f32 = ptr(qkv), B, T, Cx, att.num_heads, ptr(ao.hi), ptr(ao.f32)
            keep.extend([aa, ao])
            launch(lib.ctdd_unet_attention, C.byref(aa))
            if tc is not None:
                tc.record_attention(att, qkv, ao, B, T, Cx)
            y = _Tensor(self, B, x.H, x.W, Cx)
            conv([(ao, Cx, SEG_1x1)], [(att.proj_out.weight, 0)],
                 att.proj_out.bias.detach().float().contiguous(), Cx, x.H, x.W, x.H, x.W, y, res=x, bias_params=[att.proj_out.bias])
            return y

        feats = [cur]
        for layer in list(net.down)[1:]:
            if hasattr(layer, "resblocks"):
                cur = resblock(layer.resblocks, [cur])
                if layer.attention is not None:
                    cur = attention(layer.attention, cur)
            else:                                      # Downsample: stride-2 conv, pad right/bottom by one
                cv = layer.downsample[0]
                Ho, Wo = (cur.H + 1 - 3) // 2 + 1, (cur.W + 1 - 3) // 2 + 1
                y = _Tensor(self, B, Ho, Wo, cur.C)
                conv([(cur, cur.C, SEG_3x3_S2)], [(cv.weight, 0)],
                     cv.bias.detach().float().contiguous(), cur.C, Ho, Wo, cur.H, cur.W, y, bias_params=[cv.bias])
                cur = y
            feats.append(cur)
        for layer in net.mid:
            cur = resblock(layer.resblocks, [cur])
            if layer.attention is not None:
                cur = attention(layer.attention, cur)
        for layer in net.up:
            if hasattr(layer, "resblocks"):
                cur = resblock(layer.resblocks, [cur, feats.pop()])
                if layer.attention is not None:
                    cur = attention(layer.attention, cur)
            else:                                      # Upsample: nearest x2 folded into the conv's addressing
                cv = layer[1]
                y = _Tensor(self, B, cur.H * 2, cur.W * 2, cur.C)
                if self.precise and tc is None:
                    conv([(cur, cur.C, SEG_3x3_UP)], [(cv.weight, 0)],
                         cv.bias.detach().float().contiguous(), cur.C, cur.H * 2, cur.W * 2, cur.H, cur.W, y)
                else:                                  # materialise the 2x grid (cheap), then a stride-1 convolution (training: both modes)
                    up = _Tensor(self, B, cur.H * 2, cur.W * 2, cur.C, stats=False)
                    if self.precise:
                        launch(lib.ctdd_unet_upsample2x_f32, ptr(cur.f32), B, cur.H, cur.W, cur.C, ptr(up.f32))
                    else:
                        launch(lib.ctdd_unet_upsample2x, ptr(cur.hi), B, cur.H, cur.W, cur.C, ptr(up.hi))
                    if tc is not None:
                        tc.record_upsample(cur, up)
                    conv([(up, cur.C, SEG_3x3)], [(cv.weight, 0)],
                         cv.bias.detach().float().contiguous(), cur.C, up.H, up.W, up.H, up.W, y, bias_params=[cv.bias])
                cur = y
        ao = gn_apply([cur], net.out[0], True, net.out[0].eps, cur.H * cur.W)
        oc = net.out[2]
        n_out = oc.weight.shape[0]
        D = Cin * H0 * W0
        if logistic:
            st.net_out = torch.empty((B * H0 * W0, n_out), dtype=torch.float32, device=dev)
            conv([(ao, ao.C, SEG_3x3)], [(oc.weight, 0)],
                 oc.bias.detach().float().contiguous(), n_out, H0, W0, H0, W0, None, out_f32_tensor=st.net_out, bias_params=[oc.bias])
            ldt = torch.bfloat16 if (logits_bf16 and tc is None and not self.precise and S % 4 == 0) else torch.float32
            st.logits = logits_out if logits_out is not None else torch.empty((B, D, S), dtype=ldt, device=dev)
            assert st.logits.dtype == ldt
            la = _LogisticArgs()
            la.net, la.x0, la.B, la.C, la.HW, la.S, la.fix = ptr(st.net_out), ptr(st.x0), B, Cin, H0 * W0, S, int(bool(m.fix_logistic))
            if ldt == torch.bfloat16:                  # (the sampler loops of the bf16 engine: the head writes what the bf16 step kernel reads)
                la.out_bf16 = ptr(st.logits)
            else:
                la.out = ptr(st.logits)
            la.fast = 0 if self.precise else 1
            keep.append(la)
            if tc is None:                             # (training: the head runs as differentiable device ops on net_out)
                launch(lib.ctdd_unet_logistic_head, C.byref(la))
        else:
            ldt = torch.bfloat16 if logits_bf16 else torch.float32
            st.logits = logits_out if logits_out is not None else torch.empty((B, D, S), dtype=ldt, device=dev)
            assert st.logits.dtype == ldt
            conv([(ao, ao.C, SEG_3x3)], [(oc.weight, 0)],
                 oc.bias.detach().float().contiguous(), n_out, H0, W0, H0, W0, None, out_f32_tensor=st.logits,
                 logits_C=Cin, bias_params=[oc.bias])

        if tc is not None:                             # backward plan: built before the pools are laid out
            st.launch, st.conv, st.stats_views, st.ptr, st.cur_lists, st.keep = launch, conv, stats_views, ptr, cur_lists, keep
            st.zero_views_fwd = zero_views
            tc.finish(st, self)
        # ---- the per-(b, channel) statistics pool: one buffer, zeroed once per forward
        st.stats = torch.zeros((max(self._stats_off, 1),), dtype=torch.float64, device=dev)
        base = st.stats.data_ptr()
        for item in stats_views:
            if len(item) == 2:
                item[0].stats = base + 8 * item[1]
            else:
                setattr(item[0], item[2], base + 8 * item[1])
        nz = sum(n for _, n in zero_views)
        st.zpool = torch.zeros((max(nz, 1),), dtype=torch.float32, device=dev)
        zo = 0
        for a_, n in zero_views:
            a_.acc_buf = st.zpool.data_ptr() + 4 * zo
            zo += n
        st.plan, st.keep, st.graph = plan, keep, None
        return st

    # ------------------------------------------------------------------ execution
    def _run_plan(self, st):
        st.stats.zero_()
        st.zpool.zero_()
        for step in st.plan:
            step()

    def _prepare(self, B, x_dtype, x, times, logits_out=None, logits_bf16=False, uniform_t=False, time_row=None):
        """Build, warm up and capture the plan for (B, dtype)."""
        st = self._build(B, x_dtype, logits_out, logits_bf16=logits_bf16, uniform_t=uniform_t)
        st.x_in.copy_(x.reshape(st.x_in.shape))
        self._set_time(st, times, time_row)
        self._run_plan(st)                    # eager warm-up (also sets the LDS attributes)
        torch.cuda.synchronize()
        if getattr(self.cfg.model, "engine_graph", True):
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._run_plan(st)
                st.graph = g
            except native.CtddError:          # a kernel refused its arguments: never hide that
                raise
            except RuntimeError as e:         # stream capture refused (a HIP error, not a kernel-argument error): eager launches
                import warnings
                warnings.warn(f"[ctdd] UNetEngine: HIP-graph capture failed ({e}); the plan runs as eager launches", RuntimeWarning)
                st.graph = None
                torch.cuda.synchronize()
        return st

    @staticmethod
    def _set_time(st, times, time_row):
        if st.time_row:
            if time_row is None or time_row.numel() != st.tproj.numel():
                raise native.CtddError("UNetEngine: this plan takes a precomputed time-projection row (time_table()[i])")
            st.tproj.copy_(time_row.reshape(st.tproj.shape))
        else:
            st.t_in.copy_(times.float())

    @staticmethod
    def _replay(st, x, times, time_row=None):
        st.x_in.copy_(x.reshape(st.x_in.shape))
        UNetEngine._set_time(st, times, time_row)
        if st.graph is not None:
            st.graph.replay()
        else:
            for step in st.plan:              # (stats / split-K pools are zeroed inside _run_plan for the eager path)
                pass
            raise native.CtddError("eager replay goes through _run_plan")

    def time_table(self, times):
        """(T,) times -> (T, Ntot) fp32: the time embedding MLP and every ResBlock's time projection for T time values at once
        (`ctdd_unet_time`, the per-sample path with the T values as its rows).  A sampler computes it for its whole grid when it
        starts and hands row i to step i (`time_row=`): the plans then run without the two time launches at their head."""
        lib = _lib()
        net, dev = self.net, self.dev
        ver = self._weights_version()
        tw_ = self.__dict__.get("_time_w")
        if tw_ is None or tw_[0] != ver:
            ch = self.cfg.model.ch
            resblocks = [mod.resblocks for mod in list(net.down) + list(net.mid) + list(net.up) if hasattr(mod, "resblocks")]
            pw = torch.cat([rb.time[1].weight.detach().float() for rb in resblocks], 0).t().contiguous()   # [tdim][Ntot]
            pb = torch.cat([rb.time[1].bias.detach().float() for rb in resblocks], 0).contiguous()
            tw = [net.time[1].weight.t(), net.time[1].bias, net.time[3].weight.t(), net.time[3].bias]
            tw = [w.detach().float().contiguous() for w in tw]
            tw_ = self._time_w = (ver, ch, pw, pb, tw)
        _, ch, pw, pb, tw = tw_
        t = times.to(dev).float().contiguous().reshape(-1)
        T, tdim, Ntot = t.numel(), ch * 4, pw.shape[1]
        hid = torch.empty((T, tdim), dtype=torch.float32, device=dev)
        act = torch.empty((T, tdim), dtype=torch.float32, device=dev)
        out = torch.empty((T, Ntot), dtype=torch.float32, device=dev)
        ta = _TimeArgs()
        ta.t, ta.B, ta.ch, ta.tdim = t.data_ptr(), T, ch, tdim
        ta.w1, ta.b1, ta.w2, ta.b2, ta.hid, ta.act = tw[0].data_ptr(), tw[1].data_ptr(), tw[2].data_ptr(), tw[3].data_ptr(), hid.data_ptr(), act.data_ptr()
        rc = lib.ctdd_unet_time(C.byref(ta), pw.data_ptr(), pb.data_ptr(), Ntot, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise native.CtddError(f"ctdd_unet_time failed ({rc}): {lib.ctdd_last_error().decode()}")
        return out

    def __call__(self, x, times, logits_bf16=False, uniform_time=False, slot=None, time_row=None):
        """logits_bf16: write the (B, D, S) logits in bf16 (bf16 engine: the output convolution's epilogue, or the logistic head).
        uniform_time: the caller guarantees that every entry of `times` is the same value (the sampler loops): the time path
        runs once, as one launch, for times[0].
        slot: the caller drives several INDEPENDENT sub-batches itself, each on its own stream (TauL's pipelined loop): plan
        `slot` has buffers of its own and replays on the caller's current stream, with no sub-batch split in here."""
        B = x.shape[0]
        lb = bool(logits_bf16) and not self.precise and (self.cfg.model.model_output == "logits" or self.net.S % 4 == 0)
        ut = "row" if time_row is not None else bool(uniform_time)          # (time_row: the row of time_table() for this call's time)
        ver = self._weights_version()
        if ver != self._wver:                     # weights changed (optimizer step, EMA swap): re-pack
            self._plans.clear()
            self._wver = ver
        if x.dtype not in (torch.int64, torch.int32):
            raise native.CtddError(f"UNetEngine expects integer states, got {x.dtype}")
        # Sub-batches on parallel streams: samples are independent, and two half-size forwards in flight fill the
        # CUs that one forward leaves idle (half-empty last rounds of the 392-tile grids, 98-workgroup 7x7 levels).
        nsub = int(getattr(self.cfg.model, "engine_streams", 2)) if slot is None else 1
        if nsub > 1 and B % nsub == 0 and B // nsub >= 32 and getattr(self.cfg.model, "engine_graph", True):
            key = (B, x.dtype, nsub, ut, lb)
            grp = self._plans.get(key)
            Bs = B // nsub
            xs = x.reshape(B, -1)
            if grp is None:
                C_, H_, W_ = self.cfg.data.shape
                logits = torch.empty((B, C_ * H_ * W_, self.net.S), dtype=torch.bfloat16 if lb else torch.float32, device=self.dev)
                subs = [self._prepare(Bs, x.dtype, xs[i * Bs:(i + 1) * Bs], times[i * Bs:(i + 1) * Bs], logits[i * Bs:(i + 1) * Bs],
                                      logits_bf16=lb, uniform_t=ut, time_row=time_row) for i in range(nsub)]
                grp = self._plans[key] = (logits, subs, [torch.cuda.Stream(device=self.dev) for _ in range(nsub - 1)])
            logits, subs, streams = grp
            if all(s.graph is not None for s in subs):
                main = torch.cuda.current_stream()
                ready = torch.cuda.Event()
                ready.record(main)
                done = []
                for i in range(1, nsub):
                    with torch.cuda.stream(streams[i - 1]):
                        streams[i - 1].wait_event(ready)
                        self._replay(subs[i], xs[i * Bs:(i + 1) * Bs], times[i * Bs:(i + 1) * Bs], time_row)
                        ev = torch.cuda.Event()
                        ev.record(streams[i - 1])
                        done.append(ev)
                self._replay(subs[0], xs[:Bs], times[:Bs], time_row)
                for ev in done:
                    main.wait_event(ev)
                return logits
        key = (B, x.dtype, ut, lb) if slot is None else (B, x.dtype, "slot", int(slot), ut, lb)
        st = self._plans.get(key)
        if st is None:
            st = self._plans[key] = self._prepare(B, x.dtype, x, times, logits_bf16=lb, uniform_t=ut, time_row=time_row)
        st.x_in.copy_(x.reshape(st.x_in.shape))
        self._set_time(st, times, time_row)
        if st.graph is not None:
            st.graph.replay()
        else:
            self._run_plan(st)
        return st.logits

    # ------------------------------------------------------------------ training (ctdd/unet_train.py lays the plans out)
    def _time_projections(self, times, resblocks):
        """temb -> every ResBlock's time projection, (B, Ntot): the B x 4ch time MLP stays differentiable device ops."""
        net = self.net
        act = torch.nn.functional.silu(net.time(times.float()))
        w = torch.cat([rb.time[1].weight for rb in resblocks], 0)
        b = torch.cat([rb.time[1].bias for rb in resblocks], 0)
        return torch.nn.functional.linear(act, w, b)

    def train_forward(self, x, times):
        """model(x, t) with gradients: logits (B, D, S) attached to autograd through ONE Function whose backward is the
        hand-written backward plan."""
        from . import unet_train
        unet_train.lib()                                   # (binds the training entry points' signatures)
        B = x.shape[0]
        if x.dtype not in (torch.int64, torch.int32):
            raise native.CtddError(f"UNetEngine expects integer states, got {x.dtype}")
        dropout = bool(self.model.training) and any(float(getattr(mod, "p", 0.0)) > 0 for mod in self.net.modules()
                                                    if mod.__class__.__name__ == "Dropout")
        key = (B, x.dtype, dropout)
        pool = self.__dict__.setdefault("_train_plans", {}).setdefault(key, [])
        st = next((p for p in pool if not p.busy), None)
        if st is None:
            if len(pool) >= int(getattr(self.cfg.model, "engine_train_plans", 3)):
                st = min(pool, key=lambda p: p.gen)          # oldest forward never got its backward: reuse, its ctx is invalidated
            else:
                tc = unet_train.TrainCtx(self, B, dropout)
                st = self._build(B, x.dtype, tc=tc)
                st.gen, st.busy, st.fgraph, st.bgraph = 0, False, None, None
                tc.rng[0] = native.dropout_seed()
                self._train_warm_and_capture(st, x, times)
                pool.append(st)
        tc = st.tc
        Cin, H0, W0 = self.cfg.data.shape
        tproj = self._time_projections(times, tc.resblocks)
        out = unet_train.UNetTrainFn.apply(self, st, x, times, tproj, *tc.engine_params)
        if self.cfg.model.model_output == "logistic_pars":
            from lib.models.models import logistic_logits
            no = out.view(B, H0 * W0, 2 * Cin)
            loc, log_scale = no[..., :Cin].permute(0, 2, 1), no[..., Cin:].permute(0, 2, 1)          # (B, C, HW)
            mu = torch.tanh(loc + st.x0.view(B, Cin, H0 * W0))
            logits = logistic_logits(mu.unsqueeze(-1), log_scale.unsqueeze(-1), self.net.S, bool(self.cfg.model.fix_logistic))
            return logits.reshape(B, Cin * H0 * W0, self.net.S)
        return out

    def _run_bwd_plan(self, st):
        tc = st.tc
        tc.zbuf.zero_()
        tc.bzpool.zero_()
        tc.gflat.zero_()
        for step in st.bwd_plan:
            step()

    def _train_warm_and_capture(self, st, x, times):
        """Run both plans once eagerly (first launches set kernel attributes), then capture each as ONE HIP graph."""
        with torch.no_grad():
            st.x_in.copy_(x.reshape(st.x_in.shape))
            st.t_in.copy_(times.float())
            st.tc.tproj.copy_(self._time_projections(times, st.tc.resblocks))
            self._run_plan(st)
            self._run_bwd_plan(st)                  # (zero seed: only exercises the launches)
            torch.cuda.synchronize()
            if getattr(self.cfg.model, "engine_graph", True):
                try:
                    gf = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gf):
                        self._run_plan(st)
                    gb = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gb):
                        self._run_bwd_plan(st)
                    st.fgraph, st.bgraph = gf, gb
                except native.CtddError:
                    raise
                except RuntimeError as e:
                    import warnings
                    warnings.warn(f"[ctdd] UNetEngine: HIP-graph capture of the training plans failed ({e}); eager launches", RuntimeWarning)
                    st.fgraph = st.bgraph = None
                    torch.cuda.synchronize()

    def _train_run_forward(self, st, x, times, tproj):
        st.gen = self.__dict__["_train_gen"] = self.__dict__.get("_train_gen", 0) + 1
        st.busy = True
        st.x_in.copy_(x.reshape(st.x_in.shape))
        st.t_in.copy_(times.float())
        st.tc.tproj.copy_(tproj.detach())
        if st.fgraph is not None:
            st.fgraph.replay()
        else:
            self._run_plan(st)
        out = st.net_out if self.cfg.model.model_output == "logistic_pars" else st.logits
        return out.clone()

    def _train_run_backward(self, st, gen, dout):
        if gen != st.gen:
            raise native.CtddError("the training plan of this forward was reused by a later forward before its backward ran "
                                   "(more concurrent training forwards than cfg.model.engine_train_plans)")
        tc = st.tc
        # gradients handed out by the previous backward are views of the arena: a caller that kept them (no zero_grad) gets copies
        lo, hi = tc.gflat.data_ptr(), tc.gflat.data_ptr() + 4 * tc.gflat.numel()
        for p in tc.engine_params:
            if p.grad is not None and lo <= p.grad.data_ptr() < hi:
                p.grad = p.grad.clone()
        Cin, H0, W0 = self.cfg.data.shape
        d = dout.detach().float()
        if self.cfg.model.model_output != "logistic_pars" and Cin > 1:        # (B, C*HW, S) -> NHWC rows [b*HW + p][c*S + s]
            S = self.net.S
            d = d.view(-1, Cin, H0 * W0, S).permute(0, 2, 1, 3)
        tc.seed_in.copy_(d.reshape(tc.seed_in.shape))
        if st.bgraph is not None:
            st.bgraph.replay()
        else:
            self._run_bwd_plan(st)
        st.busy = False
        grads = [tc.gflat[o:o + n].view(shape) for o, n, shape in (tc.grad_view[id(p)] for p in tc.engine_params)]
        return tc.dtproj.clone(), grads
