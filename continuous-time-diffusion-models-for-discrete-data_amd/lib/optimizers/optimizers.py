"""Optimizer registry entries (reference lib/optimizers/optimizers.py:4-6: plain Adam(params, lr)).

`Adam` is torch.optim.Adam (same constructor defaults, same `state_dict` layout, so reference
checkpoints load) plus `fused_step`: gradient clipping, the Adam update and the EMA shadow update of
ALL parameter tensors in two libctdd launches (csrc/optim.hip, K28).  `Standard.step` takes that path
whenever the parameters live on a GPU; it fails loudly if libctdd is missing."""
import ctypes as C

import torch

import lib.optimizers.optimizers_utils as optimizers_utils


class _OptTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("shadow", C.c_void_p),
                ("n", C.c_int64)]


class _OptChunk(C.Structure):
    _fields_ = [("tensor", C.c_int), ("pad", C.c_int), ("start", C.c_int64)]


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr):
        super().__init__(params, lr)
        self._tables = None          # (key, tensors_dev, chunks_dev, nchunks)
        self._scratch = None

    def _ensure_state(self, p):
        st = self.state[p]
        if len(st) == 0:             # torch.optim.Adam's lazy state initialisation
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def fused_step(self, max_norm=0.0, shadow_params=None, ema_decay=-1.0, all_params=None):
        """clip_grad_norm_(max_norm) -> Adam step -> shadow lerp for every parameter that has a gradient.
        shadow_params is aligned with `all_params` (the model's trainable parameters in order).
        torch.optim.Adam keeps the step count per parameter and clip_grad_norm_ takes ONE norm over all parameters:
        tensors are batched by (param group, step count) -- one batch in the usual case -- the norm is summed over all
        batches first, then each batch is updated with its own bias corrections."""
        from ctdd import native
        lib = native.load()
        shadow_of = {}
        if shadow_params is not None:
            shadow_of = {id(p): s for p, s in zip(all_params, shadow_params)}
        batches = {}                                   # (group index, step) -> [parameters]
        for gi, group in enumerate(self.param_groups):
            if group["weight_decay"] != 0 or group["amsgrad"] or group["maximize"]:
                raise native.CtddError("fused_step implements plain Adam (no weight decay / amsgrad / maximize)")
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self._ensure_state(p)
                step = st["step"]
                batches.setdefault((gi, int(step.item() if torch.is_tensor(step) else step)), []).append(p)
        if not batches:
            return
        if self._tables is None:
            self._tables = {}
        stream = torch.cuda.current_stream().cuda_stream
        prepared = []
        for (gi, step), ps in batches.items():
            rows = []
            for p in ps:
                st = self.state[p]
                g, m, v, s = p.grad, st["exp_avg"], st["exp_avg_sq"], shadow_of.get(id(p))
                for t, nm in ((p, "parameter"), (g, "gradient"), (m, "exp_avg"), (v, "exp_avg_sq")) + (((s, "shadow"),) if s is not None else ()):
                    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                        raise native.CtddError(f"fused_step: {nm} must be a contiguous fp32 GPU tensor (got {t.dtype} on {t.device})")
                rows.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), s.data_ptr() if s is not None else 0, p.numel()))
            key = tuple(rows)
            dev = ps[0].device
            tab = self._tables.get((gi, key))          # cached per (param group, tensor set)
            if tab is None:
                if len(self._tables) >= 8:
                    self._tables.clear()
                ce = lib.ctdd_opt_chunk_elems()
                tt = (_OptTensor * len(rows))(*[_OptTensor(*r) for r in rows])
                chunks = [(i, 0, s0) for i, r in enumerate(rows) for s0 in range(0, r[5], ce)]
                cc = (_OptChunk * len(chunks))(*[_OptChunk(*c) for c in chunks])
                tdev = torch.frombuffer(bytearray(bytes(tt)), dtype=torch.uint8).to(dev)
                cdev = torch.frombuffer(bytearray(bytes(cc)), dtype=torch.uint8).to(dev)
                tab = self._tables[(gi, key)] = (key, tdev, cdev, len(chunks))
            if self._scratch is None or self._scratch.device != dev:
                self._scratch = torch.zeros((1,), dtype=torch.float64, device=dev)
            prepared.append((gi, step, ps, tab))
        if max_norm > 0.0:
            for n_, (_, _, _, tab) in enumerate(prepared):
                rc = lib.ctdd_grad_sumsq(tab[1].data_ptr(), tab[2].data_ptr(), tab[3], self._scratch.data_ptr(), int(n_ == 0), stream)
                if rc != 0:
                    raise native.CtddError(f"ctdd_grad_sumsq failed ({rc}): {lib.ctdd_last_error().decode()}")
        for gi, step, ps, tab in prepared:
            group = self.param_groups[gi]
            b1, b2 = group["betas"]
            rc = lib.ctdd_adam_ema_apply(tab[1].data_ptr(), tab[2].data_ptr(), tab[3], float(group["lr"]), float(b1), float(b2),
                                         float(group["eps"]), step + 1, float(max_norm), float(ema_decay if shadow_params is not None else -1.0),
                                         self._scratch.data_ptr(), stream)
            if rc != 0:
                raise native.CtddError(f"ctdd_adam_ema_apply failed ({rc}): {lib.ctdd_last_error().decode()}")
            for p in ps:
                self.state[p]["step"] += 1


@optimizers_utils.register_optimizer
def Adam(params, cfg):
    return FusedAdam(params, cfg.optimizer.lr)
