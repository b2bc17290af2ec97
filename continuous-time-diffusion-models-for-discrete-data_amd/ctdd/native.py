"""ctypes binding of the C ABI declared in include/ctdd.h.

torch is used for device memory and streams only: every wrapper checks dtype / device /
contiguity, passes raw device pointers + the current HIP stream, and raises CtddError with the
library's message on a non-zero status.  There is NO fallback: if libctdd.so is missing or an
operand is not on a GPU the call fails loudly (the CPU restatement lives in /oracle and is test
infrastructure only).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

BRANCH_CTELBO, BRANCH_CRM = 0, 1
LOGIT_TYPES = {"direct": 0, "reverse_prob": 1, "reverse_logscale": 2}
STEP_ORDINAL, STEP_CORRECTOR, STEP_COUNT_RAW, STEP_CRM, STEP_COUNT_JUMPS, STEP_BF16, STEP_LOGITS_BF16 = 1, 2, 4, 8, 16, 32, 64


class CtddError(RuntimeError):
    pass


def lib_path():
    """libctdd.so next to the package; CTDD_LIBRARY names another build of it (A/B measurements of one kernel change)."""
    return os.environ.get("CTDD_LIBRARY") or os.path.join(_HERE, "libctdd.so")


_P, _I, _F, _U64, _U32, _I64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_int64
_SIGS = {
    "ctdd_abi_version": ([], _I),
    "ctdd_last_error": ([], C.c_char_p),
    "ctdd_rate_table": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P], _I),
    "ctdd_noise_categorical": ([_P, _P, _P, _P, _U64, _U64, _I, _I, _I, _P, _P], _I),
    "ctdd_xtilde_sample": ([_P, _P, _P, _P, _P, _U64, _U64, _I, _I, _I, _P, _P, _P, _P], _I),
    "ctdd_logprob": ([_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P], _I),
    "ctdd_reverse_rates": ([_I, _I, _P, _P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P], _I),
    "ctdd_tauleap_apply": ([_P, _P, _P, _I, _I, _I, _I, _P, _P, _P], _I),
    "ctdd_tauleap_draw": ([_P, _P, _P, _F, _U32, _U64, _U64, _I, _I, _I, _P, _P, _P], _I),
    "ctdd_tauleap_step": ([_I, _I, _P, _P, _P, _P, _P, _F, _F, _F, _U32, _U64, _U64, _I, _I, _I, _P, _P, _P], _I),
    "ctdd_lbjf_step": ([_I, _I, _P, _P, _P, _P, _F, _F, _F, _U32, _P, _U64, _U64, _I, _I, _I, _P, _P, _P, _P], _I),
    "ctdd_logprob_rp_mfma": ([_P, _P, _P, _I, _I, _I, _P, _P, _P, _P], _I),
    "ctdd_logprob_rp_bwd_mfma": ([_P, _P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _P, _P], _I),
    "ctdd_exact_step": ([_P, _P, _P, _P, _P, _U64, _U64, _I, _I, _I, _P, _P, _P, _P], _I),
    "ctdd_midpoint_predict": ([_I, _I, _P, _P, _P, _P, _F, _F, _F, _I, _I, _I, _P, _P], _I),
    "ctdd_lbjf_from_rates": ([_P, _P, _F, _P, _U64, _U64, _I, _I, _I, _P, _P, _P, _P], _I),
    "ctdd_midpoint_from_rates": ([_P, _P, _F, _I, _I, _I, _P, _P], _I),
    "ctdd_argmax": ([_P, _I, _I, _I, _P, _P], _I),
    "ctdd_initial_samples": ([_P, _U64, _U64, _I, _I, _I, _P, _P], _I),
    "ctdd_philox_uniform": ([_U64, _U64, _I64, _I, _P, _P], _I),
    "ctdd_s256_step_table_bytes": ([], _I64),
    "ctdd_s256_prepare": ([_P, _P, _F, _I, _P, _P, _P, _P], _I),
    "ctdd_s256_prepare_crm": ([_P, _P, _I, _P, _P, _P, _P], _I),
    "ctdd_tauleap_step_s256": ([_P, _P, _P, _P, _P, _P, _F, _F, _U32, _U64, _U64, _I, _I, _P, _P, _P, _P], _I),
    "ctdd_crm_loss": ([_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P], _I),
    "ctdd_ctelbo_scratch_bytes": ([_I, _I, _I], _I64),
    "ctdd_ctelbo_loss": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P], _I),
    "ctdd_ctelbo_loss_terms": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _F, _P, _P, _P, _P], _I),
    "ctdd_score_elbo_loss": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _P, _P], _I),
    "ctdd_crm_loss_ll": ([_P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P], _I),
    "ctdd_score_elbo_loss_ll": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _P, _P], _I),
    "ctdd_logprob_bwd": ([_I, _P, _P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _P], _I),
    "ctdd_opt_chunk_elems": ([], _I),
    "ctdd_adam_ema_step": ([_P, _P, _I, _F, _F, _F, _F, _I64, _F, _F, _P, _P], _I),
    "ctdd_grad_sumsq": ([_P, _P, _I, _P, _I, _P], _I),
    "ctdd_adam_ema_apply": ([_P, _P, _I, _F, _F, _F, _F, _I64, _F, _F, _P, _P], _I),
}
UNET_EXPORTS = ("ctdd_unet_conv", "ctdd_unet_conv_patch", "ctdd_unet_conv_res", "ctdd_unet_conv_ring", "ctdd_unet_upsample2x", "ctdd_unet_first_conv", "ctdd_unet_gn_apply", "ctdd_unet_gn_onepass", "ctdd_unet_channel_stats",
                "ctdd_unet_time", "ctdd_unet_time_uniform", "ctdd_unet_attention", "ctdd_unet_logistic_head")    # bound in ctdd/unet_engine.py
HOLLOW_EXPORTS = ("ctdd_gemm_bf16", "ctdd_hollow_small_linear", "ctdd_hollow_embed", "ctdd_hollow_layernorm", "ctdd_hollow_add", "ctdd_hollow_put_rows", "ctdd_hollow_attention", "ctdd_hollow_attention_bf16")
UNET_TRAIN_EXPORTS = ("ctdd_unet_wgrad", "ctdd_unet_gn_bwd", "ctdd_unet_dropout", "ctdd_unet_colsum", "ctdd_unet_sum_batch", "ctdd_unet_sum_jobs", "ctdd_unet_accumulate",
                      "ctdd_unet_downsum2x", "ctdd_unet_upsample2x_f32", "ctdd_unet_cast_rows", "ctdd_unet_attention_bwd",
                      "ctdd_unet_first_conv_wgrad", "ctdd_unet_first_conv_wgrad_scratch", "ctdd_unet_pack_weights",
                      "ctdd_unet_unpack_grads")    # bound in ctdd/unet_train.py
HOLLOW_TRAIN_EXPORTS = ("ctdd_hollow_layernorm_bwd", "ctdd_hollow_attention_train", "ctdd_hollow_attention_bwd", "ctdd_hollow_attention_train_bf16", "ctdd_hollow_attention_bwd_bf16", "ctdd_hollow_act", "ctdd_hollow_relu_bf16", "ctdd_hollow_colsum", "ctdd_hollow_dropout",
                        "ctdd_hollow_embed_bwd")                                       # bound in ctdd/hollow_train.py
EXPORTS = tuple(_SIGS) + UNET_EXPORTS + HOLLOW_EXPORTS + UNET_TRAIN_EXPORTS + HOLLOW_TRAIN_EXPORTS


def load():
    """dlopen libctdd.so once; raise (never fall back) when it is absent."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise CtddError(f"{path} not found: build it with `python __graft_entry__.py` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        lib = C.CDLL(path)
        for name, (argt, rest) in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = argt, rest
        _LIB = lib
    return _LIB


def _ptr(t, dtype=None, name="tensor"):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise CtddError(f"{name}: expected a torch tensor, got {type(t)}")
    if not t.is_cuda:
        raise CtddError(f"{name}: must live on a GPU (got {t.device}); libctdd has no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise CtddError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise CtddError(f"{name}: must be contiguous")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check(rc, what):
    if rc != 0:
        msg = load().ctdd_last_error().decode(errors="replace")
        raise CtddError(f"{what} failed with status {rc}: {msg}")


f32, i32 = torch.float32, torch.int32


def rate_table(eigvecs, right, eigvals, base_rate, integral, beta, S, normalise, clamp_below=1e-8,
               want_qt0=True, want_qt0T=False, want_rate=False, want_noise_probs=False):
    nT = integral.numel()
    dev = integral.device
    mk = lambda want: torch.empty((nT, S, S), dtype=f32, device=dev) if want else None
    q, qT, r, pn = mk(want_qt0), mk(want_qt0T), mk(want_rate), mk(want_noise_probs)
    rc = load().ctdd_rate_table(_ptr(eigvecs, f32, "eigvecs"), _ptr(right, f32, "right"), _ptr(eigvals, f32, "eigvals"),
                                _ptr(base_rate, f32, "base_rate"), _ptr(integral, f32, "integral"),
                                _ptr(beta, f32, "beta"), nT, S, int(bool(normalise)), float(clamp_below),
                                _ptr(q), _ptr(qT), _ptr(r), _ptr(pn), _stream())
    _check(rc, "ctdd_rate_table")
    return q, qT, r, pn


def noise_categorical(probs, x0, tidx=None, E=None, seed=0, offset=0):
    B, D = x0.shape
    S = probs.shape[-1]
    out = torch.empty((B, D), dtype=i32, device=x0.device)
    rc = load().ctdd_noise_categorical(_ptr(probs, f32, "probs"), _ptr(tidx, i32, "tidx"), _ptr(x0, i32, "x0"),
                                       _ptr(E, f32, "E"), seed, offset, B, D, S, _ptr(out), _stream())
    _check(rc, "ctdd_noise_categorical")
    return out


def xtilde_sample(rate, x_t, tidx=None, E_dim=None, E_val=None, seed=0, offset=0):
    B, D = x_t.shape
    S = rate.shape[-1]
    dev = x_t.device
    dims = torch.empty((B,), dtype=i32, device=dev)
    newval = torch.empty((B,), dtype=i32, device=dev)
    xt = torch.empty((B, D), dtype=i32, device=dev)
    rc = load().ctdd_xtilde_sample(_ptr(rate, f32, "rate"), _ptr(tidx, i32, "tidx"), _ptr(x_t, i32, "x_t"),
                                   _ptr(E_dim, f32, "E_dim"), _ptr(E_val, f32, "E_val"), seed, offset, B, D, S,
                                   _ptr(dims), _ptr(newval), _ptr(xt), _stream())
    _check(rc, "ctdd_xtilde_sample")
    return dims, newval, xt


def _rp_mfma_ok(logit_type, logits, qt0):
    N, D, S = logits.shape
    return logit_type == "reverse_prob" and S % 32 == 0 and 32 <= S <= 256 and qt0 is not None and qt0.dim() == 3 and qt0.shape[0] == N


def logprob(logits, x, qt0, logit_type, tidx=None, qt0T=None):
    """get_logprob_with_logits.  qt0T given (per-sample transposed tables), reverse_prob, S % 32 == 0: the matrix-core path."""
    N, D, S = logits.shape
    ll_all = torch.empty_like(logits)
    ll_xt = torch.empty((N, D), dtype=f32, device=logits.device)
    if qt0T is not None and _rp_mfma_ok(logit_type, logits, qt0):
        scratch = torch.empty_like(logits)
        rc = load().ctdd_logprob_rp_mfma(_ptr(logits, f32, "logits"), _ptr(x, i32, "x"), _ptr(qt0T, f32, "qt0T"), N, D, S, _ptr(scratch),
                                         _ptr(ll_all), _ptr(ll_xt), _stream())
        _check(rc, "ctdd_logprob_rp_mfma")
        _count("ctdd_logprob")
        return ll_all, ll_xt
    rc = load().ctdd_logprob(_ptr(logits, f32, "logits"), _ptr(x, i32, "x"), _ptr(qt0, f32, "qt0"),
                             _ptr(tidx, i32, "tidx"), LOGIT_TYPES[logit_type], N, D, S, _ptr(ll_all), _ptr(ll_xt),
                             _stream())
    _check(rc, "ctdd_logprob")
    return ll_all, ll_xt


def reverse_rates(branch, logit_type, logits, x, qt0, rate, eps, tidx=None, want_ratio=True):
    N, D, S = logits.shape
    rr = torch.empty_like(logits)
    ratio = torch.empty_like(logits) if want_ratio else None
    rc = load().ctdd_reverse_rates(branch, LOGIT_TYPES[logit_type], _ptr(logits, f32, "logits"), _ptr(x, i32, "x"),
                                   _ptr(qt0, f32, "qt0"), _ptr(rate, f32, "rate"), _ptr(tidx, i32, "tidx"),
                                   float(eps), N, D, S, _ptr(rr), _ptr(ratio), _stream())
    _check(rc, "ctdd_reverse_rates")
    return rr, ratio


def tauleap_apply(x, jump_nums, is_ordinal, x_base=None, changed=None):
    N, D, S = jump_nums.shape
    out = torch.empty((N, D), dtype=i32, device=x.device)
    rc = load().ctdd_tauleap_apply(_ptr(x, i32, "x"), _ptr(x_base, i32, "x_base"), _ptr(jump_nums, f32, "jump_nums"),
                                   int(bool(is_ordinal)), N, D, S, _ptr(out), _ptr(changed, i32, "changed"), _stream())
    _check(rc, "ctdd_tauleap_apply")
    return out


def tauleap_draw(rates, x, h, is_ordinal, seed, offset, x_base=None, changed=None):
    N, D, S = rates.shape
    out = torch.empty((N, D), dtype=i32, device=x.device)
    rc = load().ctdd_tauleap_draw(_ptr(rates, f32, "rates"), _ptr(x, i32, "x"), _ptr(x_base, i32, "x_base"), float(h),
                                  STEP_ORDINAL if is_ordinal else 0, seed, offset, N, D, S, _ptr(out),
                                  _ptr(changed, i32, "changed"), _stream())
    _check(rc, "ctdd_tauleap_draw")
    return out


def tauleap_step(branch, logit_type, logits, x, qt0, base_rate, beta, eps, h, flags, seed, offset,
                 x_base=None, out=None, changed=None):
    N, D, S = logits.shape
    if out is None:
        out = torch.empty((N, D), dtype=i32, device=x.device)
    rc = load().ctdd_tauleap_step(branch, LOGIT_TYPES[logit_type], _ptr(logits, f32, "logits"), _ptr(x, i32, "x"),
                                  _ptr(x_base, i32, "x_base"), _ptr(qt0, f32, "qt0"), _ptr(base_rate, f32, "base_rate"),
                                  float(beta), float(eps), float(h), int(flags), seed, offset, N, D, S,
                                  _ptr(out, i32, "out"), _ptr(changed, i32, "changed"), _stream())
    _check(rc, "ctdd_tauleap_step")
    return out


def lbjf_step(branch, logit_type, logits, x, qt0, base_rate, beta, eps, h, flags=0, E=None, seed=0, offset=0,
              want_probs=False, changed=None):
    N, D, S = logits.shape
    out = torch.empty((N, D), dtype=i32, device=x.device)
    probs = torch.empty_like(logits) if want_probs else None
    rc = load().ctdd_lbjf_step(branch, LOGIT_TYPES[logit_type], _ptr(logits, f32, "logits"), _ptr(x, i32, "x"),
                               _ptr(qt0, f32, "qt0"), _ptr(base_rate, f32, "base_rate"), float(beta), float(eps),
                               float(h), int(flags), _ptr(E, f32, "E"), seed, offset, N, D, S, _ptr(out), _ptr(probs),
                               _ptr(changed, i32, "changed"), _stream())
    _check(rc, "ctdd_lbjf_step")
    return (out, probs) if want_probs else out


def exact_step(logits, x, q_lo, q_step, E=None, seed=0, offset=0, want_probs=False, changed=None):
    """ExactSampling step: x_new ~ Categorical((softmax(logits) @ q_lo) * q_step[:, x]) per dimension (K5-style contraction +
    exponential race in one launch)."""
    N, D, S = logits.shape
    out = torch.empty((N, D), dtype=i32, device=x.device)
    probs = torch.empty_like(logits) if want_probs else None
    rc = load().ctdd_exact_step(_ptr(logits, f32, "logits"), _ptr(x, i32, "x"), _ptr(q_lo, f32, "q_lo"), _ptr(q_step, f32, "q_step"),
                                _ptr(E, f32, "E"), seed, offset, N, D, S, _ptr(out), _ptr(probs), _ptr(changed, i32, "changed"), _stream())
    _check(rc, "ctdd_exact_step")
    return (out, probs) if want_probs else out


def midpoint_predict(branch, logit_type, logits, x, qt0, base_rate, beta, eps, h):
    N, D, S = logits.shape
    out = torch.empty((N, D), dtype=i32, device=x.device)
    rc = load().ctdd_midpoint_predict(branch, LOGIT_TYPES[logit_type], _ptr(logits, f32, "logits"), _ptr(x, i32, "x"),
                                      _ptr(qt0, f32, "qt0"), _ptr(base_rate, f32, "base_rate"), float(beta),
                                      float(eps), float(h), N, D, S, _ptr(out), _stream())
    _check(rc, "ctdd_midpoint_predict")
    return out


def lbjf_from_rates(rates, x, h, E=None, seed=0, offset=0, want_probs=False, changed=None):
    """LBJF posterior + draw on masked reverse rates (N, D, S) that are already computed (tauleap_step_s256(want_rates=True))."""
    N, D, S = rates.shape
    out = torch.empty((N, D), dtype=i32, device=x.device)
    probs = torch.empty_like(rates) if want_probs else None
    rc = load().ctdd_lbjf_from_rates(_ptr(rates, f32, "rates"), _ptr(x, i32, "x"), float(h), _ptr(E, f32, "E"), seed, offset,
                                     N, D, S, _ptr(out), _ptr(probs), _ptr(changed, i32, "changed"), _stream())
    _check(rc, "ctdd_lbjf_from_rates")
    return (out, probs) if want_probs else out


def midpoint_from_rates(rates, x, h):
    N, D, S = rates.shape
    out = torch.empty((N, D), dtype=i32, device=x.device)
    rc = load().ctdd_midpoint_from_rates(_ptr(rates, f32, "rates"), _ptr(x, i32, "x"), float(h), N, D, S, _ptr(out), _stream())
    _check(rc, "ctdd_midpoint_from_rates")
    return out


def argmax(logits):
    N, D, S = logits.shape
    out = torch.empty((N, D), dtype=i32, device=logits.device)
    _check(load().ctdd_argmax(_ptr(logits, f32, "logits"), N, D, S, _ptr(out), _stream()), "ctdd_argmax")
    return out


def initial_samples(N, D, S, device, seed, offset, cdf=None):
    out = torch.empty((N, D), dtype=i32, device=device)
    if not out.is_cuda:
        raise CtddError("initial_samples: device must be a GPU")
    _check(load().ctdd_initial_samples(_ptr(cdf, f32, "cdf"), seed, offset, N, D, S, _ptr(out), _stream()),
           "ctdd_initial_samples")
    return out


def philox_uniform(seed, offset, nrows, nblk, device):
    out = torch.empty((nrows, nblk * 4), dtype=f32, device=device)
    _check(load().ctdd_philox_uniform(seed, offset, nrows, nblk, _ptr(out), _stream()), "ctdd_philox_uniform")
    return out


def crm_loss(logits, xt, x0, qt0, loss_type, scale, nll_scale):
    """K12: (loss scalar tensor, d loss / d logits) for the CRM objectives with direct logits."""
    B, D, S = logits.shape
    grad = torch.empty_like(logits)
    rows = torch.empty((B * D,), dtype=torch.float64, device=logits.device)
    out = torch.empty((1,), dtype=torch.float32, device=logits.device)
    lt = {"rm": 0, "mle": 1, "elbo": 2}[loss_type]
    _check(load().ctdd_crm_loss(_ptr(logits, torch.float32, "logits"), _ptr(xt, torch.int32, "xt"),
                                _ptr(x0, torch.int32, "x0") if x0 is not None else None,
                                _ptr(qt0, torch.float32, "qt0") if qt0 is not None else None, B, D, S, lt, float(scale),
                                float(nll_scale), _ptr(grad), _ptr(rows), _ptr(out), _stream()), "ctdd_crm_loss")
    return out[0], grad


def dropout_seed():
    """Seed of a network's dropout stream: drawn from torch's CPU generator (reproducible under torch.manual_seed) and mixed with
    the data-parallel rank, so that ranks seeded alike still drop different units."""
    seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            seed = (seed + dist.get_rank() * 0x9E3779B97F4A7C15) % (2**62)
    except Exception:
        pass
    return seed


LAUNCH_COUNTS = {}      # entry point -> calls (tests assert that an objective really ran through the HIP path)


def _count(name):
    LAUNCH_COUNTS[name] = LAUNCH_COUNTS.get(name, 0) + 1


def crm_loss_ll(ll_all, xt, qt0, loss_type, scale):
    """K12 on ll_all (reverse logit types): (loss scalar tensor, d loss / d ll_all)."""
    B, D, S = ll_all.shape
    grad = torch.empty_like(ll_all)
    rows = torch.empty((B * D,), dtype=torch.float64, device=ll_all.device)
    out = torch.empty((1,), dtype=torch.float32, device=ll_all.device)
    lt = {"rm": 0, "mle": 1, "elbo": 2}[loss_type]
    _check(load().ctdd_crm_loss_ll(_ptr(ll_all, torch.float32, "ll_all"), _ptr(xt, torch.int32, "xt"),
                                   _ptr(qt0, torch.float32, "qt0") if qt0 is not None else None, B, D, S, lt, float(scale),
                                   _ptr(grad), _ptr(rows), _ptr(out), _stream()), "ctdd_crm_loss_ll")
    _count("ctdd_crm_loss_ll")
    return out[0], grad


def score_elbo_loss_ll(ll_all, x0, x_tilde, reg_x, qt0, rate, eps, nll_scale):
    """ScoreElbo on ll_all (reverse logit types): (loss scalar tensor, d loss / d ll_all)."""
    B, D, S = ll_all.shape
    lib = load()
    scratch = torch.empty((int(lib.ctdd_ctelbo_scratch_bytes(B, D, S)),), dtype=torch.uint8, device=ll_all.device)
    grad = torch.empty_like(ll_all)
    out = torch.empty((1,), dtype=torch.float32, device=ll_all.device)
    _check(lib.ctdd_score_elbo_loss_ll(_ptr(ll_all, torch.float32, "ll_all"), _ptr(x0, torch.int32, "x0"), _ptr(x_tilde, torch.int32, "x_tilde"),
                                       _ptr(reg_x, torch.int32, "reg_x"), _ptr(qt0, torch.float32, "qt0"), _ptr(rate, torch.float32, "rate"),
                                       B, D, S, float(eps), float(nll_scale), _ptr(scratch), _ptr(grad), _ptr(out), _stream()),
           "ctdd_score_elbo_loss_ll")
    _count("ctdd_score_elbo_loss_ll")
    return out[0], grad


def logprob_bwd(logit_type, logits, qt0, qt0T, dll, x0=None, nll_scale=0.0, ll_all=None):
    """d/dlogits of ll_all = get_logprob_with_logits(logits) for the reverse logit types given dll = d loss / d ll_all;
    x0: also the cross-entropy term nll_scale * sum -log_softmax(logits)[x0] (gradient added, value returned).
    ll_all (the forward's output) given, reverse_prob, S % 32 == 0: the matrix-core path."""
    B, D, S = logits.shape
    grad = torch.empty_like(logits)
    ce_rows = torch.zeros((B * D,), dtype=torch.float64, device=logits.device) if x0 is not None else None
    out_ce = torch.zeros((1,), dtype=torch.float32, device=logits.device)
    if ll_all is not None and _rp_mfma_ok(logit_type, logits, qt0):
        scratch = torch.empty((2,) + tuple(logits.shape), dtype=torch.float32, device=logits.device)
        _check(load().ctdd_logprob_rp_bwd_mfma(_ptr(logits, torch.float32, "logits"), _ptr(qt0, torch.float32, "qt0"), _ptr(ll_all, torch.float32, "ll_all"),
                                               _ptr(dll, torch.float32, "dll"), _ptr(x0, torch.int32, "x0") if x0 is not None else None,
                                               float(nll_scale), B, D, S, _ptr(scratch), _ptr(grad), _ptr(ce_rows) if ce_rows is not None else None,
                                               _ptr(out_ce), _stream()), "ctdd_logprob_rp_bwd_mfma")
        _count("ctdd_logprob_bwd")
        return grad, out_ce[0]
    _check(load().ctdd_logprob_bwd(LOGIT_TYPES[logit_type], _ptr(logits, torch.float32, "logits"), _ptr(qt0, torch.float32, "qt0"),
                                   _ptr(qt0T, torch.float32, "qt0T"), _ptr(dll, torch.float32, "dll"),
                                   _ptr(x0, torch.int32, "x0") if x0 is not None else None, float(nll_scale), B, D, S, _ptr(grad),
                                   _ptr(ce_rows) if ce_rows is not None else None, _ptr(out_ce), _stream()), "ctdd_logprob_bwd")
    _count("ctdd_logprob_bwd")
    return grad, out_ce[0]


def score_elbo_loss(logits, x0, x_tilde, reg_x, qt0, rate, eps, nll_scale):
    """(loss scalar tensor, d loss / d logits) of ScoreElbo with direct logits."""
    B, D, S = logits.shape
    lib = load()
    scratch = torch.empty((int(lib.ctdd_ctelbo_scratch_bytes(B, D, S)),), dtype=torch.uint8, device=logits.device)
    grad = torch.empty_like(logits)
    out = torch.empty((1,), dtype=torch.float32, device=logits.device)
    _check(lib.ctdd_score_elbo_loss(_ptr(logits, torch.float32, "logits"), _ptr(x0, torch.int32, "x0"), _ptr(x_tilde, torch.int32, "x_tilde"),
                                    _ptr(reg_x, torch.int32, "reg_x"), _ptr(qt0, torch.float32, "qt0"), _ptr(rate, torch.float32, "rate"),
                                    B, D, S, float(eps), float(nll_scale), _ptr(scratch), _ptr(grad), _ptr(out), _stream()),
           "ctdd_score_elbo_loss")
    return out[0], grad


def ctelbo_loss(logits, x0, x_tilde, qt0, qt0T, rate, eps, elbo_scale, nll_scale, reg_scale=None):
    """K11: (loss scalar tensor, d loss / d logits) of the tauLDR CT-ELBO.  `reg_scale` None: the one-forward-pass objective
    (both ELBO terms weighted `elbo_scale`); otherwise the signal term is weighted `elbo_scale` and the regulariser
    `reg_scale`, every term evaluated at the state passed as `x_tilde` (one half of the two-forward-pass objective)."""
    B, D, S = logits.shape
    lib = load()
    scratch = torch.empty((int(lib.ctdd_ctelbo_scratch_bytes(B, D, S)),), dtype=torch.uint8, device=logits.device)
    grad = torch.empty_like(logits)
    out = torch.empty((1,), dtype=torch.float32, device=logits.device)
    head = (_ptr(logits, torch.float32, "logits"), _ptr(x0, torch.int32, "x0"), _ptr(x_tilde, torch.int32, "x_tilde"),
            _ptr(qt0, torch.float32, "qt0"), _ptr(qt0T, torch.float32, "qt0T"), _ptr(rate, torch.float32, "rate"), B, D, S, float(eps))
    tail = (float(nll_scale), _ptr(scratch), _ptr(grad), _ptr(out), _stream())
    _count("ctdd_ctelbo_loss" if reg_scale is None else "ctdd_ctelbo_loss_terms")
    if reg_scale is None:
        _check(lib.ctdd_ctelbo_loss(*head, float(elbo_scale), *tail), "ctdd_ctelbo_loss")
    else:
        _check(lib.ctdd_ctelbo_loss_terms(*head, float(elbo_scale), float(reg_scale), *tail), "ctdd_ctelbo_loss_terms")
    return out[0], grad


# ---------------------------------------------------------------- S = 256 fast path
class S256Tables:
    """Derived tables for the MFMA tau-leaping kernel: per-step blocks (resident for the whole
    time grid) + the two per-model base-rate views."""

    def __init__(self, qt0, base_rate, eps, crm=False, bf16=False):
        """crm: tables of the CRM branch with logit_type reverse_prob (unit left scaling; step calls carry STEP_CRM).
        bf16: step calls carry STEP_BF16 -- one bf16 product for the S x S contraction (csrc/steps_s256_b16.hip; the mode of
        the bf16 score network, relative rate error <= 3 * 2^-8) instead of the three split-bf16 products of the parity mode."""
        self.crm = bool(crm)
        self.bf16 = bool(bf16)
        nT, S, _ = qt0.shape
        if S != 256:
            raise CtddError("S256Tables needs S == 256")
        self.block = int(load().ctdd_s256_step_table_bytes())
        dev = qt0.device
        self.steps = torch.empty((nT, self.block), dtype=torch.uint8, device=dev)
        self.RT0 = torch.empty((S, S), dtype=f32, device=dev)
        self.R0 = torch.empty((S, S), dtype=f32, device=dev)
        if self.crm:
            rc = load().ctdd_s256_prepare_crm(_ptr(qt0, f32, "qt0"), _ptr(base_rate, f32, "base_rate"), nT,
                                              _ptr(self.steps), _ptr(self.RT0), _ptr(self.R0), _stream())
        else:
            rc = load().ctdd_s256_prepare(_ptr(qt0, f32, "qt0"), _ptr(base_rate, f32, "base_rate"), float(eps), nT,
                                          _ptr(self.steps), _ptr(self.RT0), _ptr(self.R0), _stream())
        _check(rc, "ctdd_s256_prepare")

    def step_ptr(self, i):
        return self.steps.data_ptr() + i * self.block


def tauleap_step_s256(logits, x, tables, i, beta, h, flags, seed, offset, x_base=None, out=None, changed=None,
                      want_rates=False, want_x=True):
    N, D, S = logits.shape
    if S != 256:
        raise CtddError("tauleap_step_s256 needs S == 256")
    rates = torch.empty(logits.shape, dtype=f32, device=logits.device) if want_rates else None
    if want_x and out is None:
        out = torch.empty((N, D), dtype=i32, device=x.device)
    lflag = 0
    if logits.dtype == torch.bfloat16:                       # bf16 logits (the bf16 U-Net engine's output): bf16 step only
        if not tables.bf16:
            raise CtddError("tauleap_step_s256: bf16 logits need S256Tables(..., bf16=True)")
        lflag = STEP_LOGITS_BF16
    rc = load().ctdd_tauleap_step_s256(_ptr(logits, torch.bfloat16 if lflag else f32, "logits"), _ptr(x, i32, "x"), _ptr(x_base, i32, "x_base"),
                                       tables.step_ptr(i), _ptr(tables.RT0), _ptr(tables.R0), float(beta), float(h),
                                       int(flags) | (STEP_CRM if tables.crm else 0) | (STEP_BF16 if tables.bf16 else 0) | lflag, seed, offset, N, D, _ptr(rates), _ptr(out, i32, "out") if want_x else None,
                                       _ptr(changed, i32, "changed"), _stream())
    _check(rc, "ctdd_tauleap_step_s256")
    return (out, rates) if want_rates else out
