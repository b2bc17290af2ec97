"""Golden-vector generator  --  runs ONLY in the build container (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [group ...]

Imports the reference (read-only, with the build-owned stub packages in oracle/stubs standing in
for third-party modules that are absent from this image), runs each hot-path function on small
seeded inputs and freezes inputs + outputs as `tests/golden/*.npz`.  The fixtures are data; no
reference source is copied.  Pins P1..P8 of SURVEY.md section 8(c).
"""
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "stubs"), "/root/reference/TAUnSDDM", ROOT]
warnings.filterwarnings("ignore")

import math  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ml_collections import ConfigDict  # noqa: E402  (stub)

import lib.models.forward_model as ref_fm  # noqa: E402
import lib.models.model_utils as ref_mu  # noqa: E402
import lib.sampling.sampling as ref_sampling  # noqa: E402
import lib.losses.losses as ref_losses  # noqa: E402

from oracle.toy_model import toy_logits  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(1)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def base_cfg(S, D, model_name="GaussianTargetRateImageX0PredEMAPaul"):
    c = ConfigDict()
    c.device = "cpu"
    c.distributed = False
    c.data = ConfigDict(dict(S=S, name="SyntheticData"))
    c.model = ConfigDict(dict(name=model_name, concat_dim=D, rate_sigma=6.0, Q_sigma=512.0,
                              time_exp=100.0, time_base=3.0, rate_const=1.7, t_func="sqrt_cos",
                              sigma_min=1.0, sigma_max=100.0, log_prob="cat"))
    c.loss = ConfigDict(dict(name="CTElbo", eps_ratio=1e-9, nll_weight=0.0, min_time=0.01,
                             one_forward_pass=True, logit_type="direct", loss_type="rm", ce_coeff=0.0))
    c.training = ConfigDict(dict(max_t=1.0, n_iters=1000))
    c.sampler = ConfigDict(dict(name="TauL", num_steps=6, min_t=0.01, eps_ratio=1e-9,
                                initial_dist="gaussian", num_corrector_steps=0,
                                corrector_step_size_multiplier=1.5, corrector_entry_time=0.0,
                                is_ordinal=True))
    return c


def make_ref_model(kind, cfg, scale=1.0):
    """A reference forward-process object + the shared toy score function."""
    base = {"gaussian": ref_fm.GaussianTargetRate, "uniform": ref_fm.UniformRate,
            "univar": ref_fm.UniformVariantRate, "birthdeath": ref_fm.BirthDeathForwardBase}[kind]

    class RefToy(base):
        def __init__(self):
            base.__init__(self, cfg, "cpu")
            self.device = "cpu"
            self.calls = []

        def __call__(self, x, t, *a):
            self.calls.append((x.clone(), t.clone()))
            return toy_logits(x, t, cfg.data.S, scale)

    return RefToy()


# ------------------------------------------------------------------------------------ P1
def gen_forward_process():
    ts = torch.tensor([0.01, 0.25, 0.5, 0.75, 1.0])
    arrs = {"ts": ts}
    for S in (8, 256):
        m = ref_fm.GaussianTargetRate(base_cfg(S, 4), "cpu")
        q, r = m.transition(ts), m.rate(ts)
        rows = np.array([0, 1, S // 4, S // 2 - 1, S // 2, S // 2 + 1, S - 2, S - 1])
        if S == 8:
            arrs.update(g8_base_rate=m.base_rate, g8_qt0=q, g8_rate=r)
        else:
            arrs.update(g256_rows=rows, g256_base_rate_rows=m.base_rate[rows], g256_base_rate_rowsum_abs=m.base_rate.abs().sum(1),
                        g256_qt0_rows=q[:, rows], g256_qt0_colsum=q.sum(1), g256_qt0_nnz=(q > 0).sum(-1),
                        g256_rate_rows=r[:, rows])
    c = base_cfg(3, 4)
    m = ref_fm.UniformRate(c, "cpu")
    arrs.update(u3_rate_matrix=m.rate_matrix, u3_qt0=m.transition(ts), u3_rate=m.rate(ts),
                u3_between=m.transit_between(torch.tensor([0.1, 0.2]), torch.tensor([0.4, 0.9])))
    tsv = torch.tensor([0.01, 0.25, 0.5, 0.75, 0.99999])
    arrs["ts_univar"] = tsv
    y = torch.tensor([[0, 1, 1], [1, 0, 1], [0, 0, 1], [1, 1, 1], [1, 0, 0]])
    arrs["univar_y"] = y
    for S, tf in ((2, "log_sqr"), (3, "sqrt_cos"), (3, "log"), (2, "sqrt_cos")):
        c = base_cfg(S, 4)
        c.model.t_func = tf
        m = ref_fm.UniformVariantRate(c, "cpu")
        k = f"v{S}_{tf}"
        arrs.update({k + "_qt0": m.transition(tsv), k + "_rate": m.rate(tsv),
                     k + "_rate_mat": m.rate_mat(y % S, tsv),
                     k + "_between": m.transit_between(tsv * 0.5, tsv)})
    m = ref_fm.BirthDeathForwardBase(base_cfg(8, 4), "cpu")
    arrs.update(bd8_base_rate=m.base_rate, bd8_qt0=m.transition(ts), bd8_rate=m.rate(ts))
    m = ref_fm.GaussianTargetRate(base_cfg(8, 4), "cpu")
    arrs.update(g8_rate_mat=m.rate_mat(torch.tensor([[0, 3, 7], [1, 1, 2], [5, 6, 0], [4, 4, 4], [7, 0, 2]]), ts),
                g8_between=m.transit_between(ts * 0.5, ts))
    save("forward_process", **arrs)


# ------------------------------------------------------------------------------------ P2
def gen_noising():
    """Run the reference CTElbo.calc_loss with one_forward_pass=False so that both x_t and
    x_tilde reach the (toy) model, and capture them.  E noise is regenerated from the same seed
    in the RNG order of SURVEY App. C."""
    arrs = {}
    for tag, kind, S, B, D in (("g16", "gaussian", 16, 4, 16), ("g256", "gaussian", 256, 3, 20),
                               ("v3", "univar", 3, 4, 16), ("v2", "univar", 2, 5, 8)):
        cfg = base_cfg(S, D)
        cfg.loss.one_forward_pass = False
        if tag == "v2":
            cfg.model.t_func = "log_sqr"
        model = make_ref_model(kind, cfg)
        loss = ref_losses.CTElbo(cfg)
        seed = 1234 + S
        g = torch.Generator().manual_seed(99)
        x0 = torch.randint(0, S, (B, D), generator=g)
        torch.manual_seed(seed)
        val = loss.calc_loss({"model": model, "n_iter": 0}, x0.clone())
        (x_t, ts), (x_tilde, _) = model.calls[0], model.calls[1]
        torch.manual_seed(seed)
        u = torch.rand((B,))
        E_xt = torch.empty(B * D, S).exponential_(1)
        E_dim = torch.empty(B, D).exponential_(1)
        E_val = torch.empty(B, S).exponential_(1)
        ts2 = u * (cfg.training.max_t - cfg.loss.min_time) + cfg.loss.min_time
        assert torch.equal(ts2, ts)
        arrs.update({f"{tag}_x0": x0, f"{tag}_ts": ts, f"{tag}_E_xt": E_xt, f"{tag}_E_dim": E_dim,
                     f"{tag}_E_val": E_val, f"{tag}_x_t": x_t, f"{tag}_x_tilde": x_tilde,
                     f"{tag}_qt0": model.transition(ts) if S <= 16 else model.transition(ts)[:, :4],
                     f"{tag}_loss": val})
    save("noising", **arrs)


# ------------------------------------------------------------------------------------ P3 / P4
def gen_rates():
    arrs = {}
    g = torch.Generator().manual_seed(7)
    for tag, kind, S, N, D in (("g256", "gaussian", 256, 2, 8), ("g16", "gaussian", 16, 3, 6),
                               ("v3", "univar", 3, 4, 9), ("u3", "uniform", 3, 2, 5)):
        cfg = base_cfg(S, D)
        model = make_ref_model(kind, cfg)
        logits = torch.randn(N, D, S, generator=g) * 2.0
        x = torch.randint(0, S, (N, D), generator=g)
        t = torch.tensor([0.07, 0.5, 0.93, 0.3][:N])
        arrs.update({f"{tag}_logits": logits, f"{tag}_x": x, f"{tag}_t": t})
        for lt in ("direct", "reverse_prob", "reverse_logscale"):
            cfg.loss.logit_type = lt
            ll_all, ll_xt = ref_mu.get_logprob_with_logits(cfg, model, x, t, logits)
            arrs.update({f"{tag}_{lt}_ll_all": ll_all, f"{tag}_{lt}_ll_xt": ll_xt})
            cfg.loss.name = "CatRM"
            rr, ratio = ref_sampling.get_reverse_rates(model, logits, x, t, cfg, N, D, S)
            arrs.update({f"{tag}_{lt}_crm_rates": rr, f"{tag}_{lt}_crm_ratio": ratio})
        cfg.loss.name = "CTElbo"
        # in sampling all rows share t; also pin the per-row-t form used by losses
        for nm, tt in (("shared", torch.full((N,), 0.37)), ("perrow", t)):
            rr, ratio = ref_sampling.get_reverse_rates(model, logits, x, tt, cfg, N, D, S)
            arrs.update({f"{tag}_ctelbo_{nm}_rates": rr, f"{tag}_ctelbo_{nm}_ratio": ratio})
    save("rates", **arrs)


# ------------------------------------------------------------------------------------ P5 / P6
def gen_samplers():
    arrs = {}
    runs = [
        # tag, sampler, kind, S, D, N, overrides
        ("taul_g16_ord", "TauL", "gaussian", 16, 12, 8, dict(is_ordinal=True)),
        ("taul_g16_nonord", "TauL", "gaussian", 16, 12, 8, dict(is_ordinal=False, scale=4.0)),
        ("taul_g16_corr", "TauL", "gaussian", 16, 12, 6, dict(corrector_entry_time=0.6, num_corrector_steps=2)),
        ("taul_g256", "TauL", "gaussian", 256, 10, 4, dict(num_steps=5)),
        ("taul_v3_crm", "TauL", "univar", 3, 15, 8, dict(loss="CatRM", logit_type="reverse_prob", is_ordinal=False, initial_dist="uniform", max_t=0.99999)),
        ("taul_v2_crm_direct", "TauL", "univar", 2, 32, 8, dict(loss="CatRMNLL", logit_type="direct", is_ordinal=False, initial_dist="uniform", max_t=0.99999, t_func="log_sqr")),
        ("taul_g16_lambda", "TauL", "gaussian", 16, 12, 5, dict(loss="CTElboLambda")),
        ("lbjf_g16", "LBJF", "gaussian", 16, 12, 8, dict()),
        ("lbjf_g16_corr", "LBJF", "gaussian", 16, 12, 6, dict(corrector_entry_time=0.6, num_corrector_steps=2)),
        ("lbjf_v2_crm", "LBJF", "univar", 2, 32, 8, dict(loss="CatRM", logit_type="reverse_logscale", initial_dist="uniform", max_t=0.99999, t_func="log_sqr")),
        ("midpoint_v3", "MidPointTauL", "univar", 3, 15, 8, dict(loss="CatRM", logit_type="reverse_prob", is_ordinal=False, initial_dist="uniform", max_t=0.99999, data_name="Maze3S", scale=3.0)),
        ("midpoint_v3_ord", "MidPointTauL", "univar", 3, 15, 8, dict(loss="CTElbo", is_ordinal=True, initial_dist="uniform", max_t=0.99999, data_name="Maze3S", scale=3.0)),
        ("midpoint_v2", "MidPointTauL", "univar", 2, 32, 8, dict(loss="CatRMNLL", logit_type="direct", is_ordinal=False, initial_dist="uniform", max_t=0.99999, t_func="log_sqr", data_name="SyntheticData")),
        ("pctaul_g16", "PCTauL", "gaussian", 16, 12, 6, dict(corrector_entry_time=0.7, num_corrector_steps=2, num_steps=8)),
        ("exact_v3", "ExactSampling", "univar", 3, 15, 8, dict(loss="CatRM", initial_dist="uniform", max_t=0.99999)),
        ("exact_u3", "ExactSampling", "uniform", 3, 9, 6, dict(loss="CatRM", initial_dist="uniform", max_t=0.99999)),
    ]
    for tag, sname, kind, S, D, N, ov in runs:
        cfg = base_cfg(S, D)
        cfg.sampler.name = sname
        cfg.loss.name = ov.get("loss", "CTElbo")
        cfg.loss.logit_type = ov.get("logit_type", "direct")
        cfg.data.name = ov.get("data_name", "SyntheticData")
        cfg.model.t_func = ov.get("t_func", "sqrt_cos")
        cfg.training.max_t = ov.get("max_t", 1.0)
        for k in ("is_ordinal", "corrector_entry_time", "num_corrector_steps", "num_steps", "initial_dist"):
            if k in ov:
                cfg.sampler[k] = ov[k]
        scale = ov.get("scale", 1.0)
        model = make_ref_model(kind, cfg, scale)
        sampler = ref_sampling.sampling_utils.get_sampler(cfg)
        seed = 4321
        torch.manual_seed(seed)
        out = sampler.sample(model, N)
        x_init = model.calls[0][0]
        meta = dict(sampler=sname, kind=kind, S=S, D=D, N=N, seed=seed, scale=scale,
                    loss=cfg.loss.name, logit_type=cfg.loss.logit_type, t_func=cfg.model.t_func,
                    max_t=cfg.training.max_t, **{k: cfg.sampler[k] for k in cfg.sampler})
        arrs[f"{tag}__meta"] = np.array(repr(meta))
        arrs[f"{tag}__x_init"] = x_init
        if sname == "PCTauL":
            arrs[f"{tag}__samples"] = out
        else:
            arrs[f"{tag}__samples"] = out[0]
            for i, extra in enumerate(out[1:]):
                arrs[f"{tag}__aux{i}"] = np.asarray(extra, dtype=np.float64)
    save("samplers", **arrs)


# ------------------------------------------------------------------------------------ P8 (U-Net)
def _tiny_unet_cfg(model_output, S, ch=16, image=8, channels=1):
    c = base_cfg(S, image * image * channels)
    c.data.image_size, c.data.shape = image, [channels, image, image]
    c.model.update(dict(name="GaussianTargetRateImageX0PredEMAPaul", padding=False, ema_decay=0.999, ch=ch,
                        num_res_blocks=1, ch_mult=[1, 2], input_channels=channels, data_min_max=[0, S - 1],
                        dropout=0.1, fix_logistic=False, model_output=model_output, num_heads=2,
                        attn_resolutions=[int(ch / 2)]))
    return c


def gen_unet():
    """Tiny reference U-Net models (logits head and logistic head).  The reference initialises the
    second conv of every ResBlock and the output conv at scale 1e-10 (logits ~ 0), so every
    parameter is re-drawn N(0, 0.15/sqrt(fan_in)+) here to make all paths matter."""
    import lib.models.models  # noqa: F401  registers
    arrs = {}
    for tag, mo, S, chn in (("logits", "logits", 16, 1), ("logistic", "logistic_pars", 16, 3)):
        cfg = _tiny_unet_cfg(mo, S, channels=chn)
        torch.manual_seed(11)
        model = ref_mu.create_model(cfg, torch.device("cpu"))
        g = torch.Generator().manual_seed(12)
        with torch.no_grad():
            for name, p in model.named_parameters():
                if p.dim() > 1:
                    fan_in = p[0].numel()
                    p.copy_(torch.randn(p.shape, generator=g) * (1.0 / math.sqrt(fan_in)))
                elif "norm" in name and name.endswith("weight") or name.endswith("0.weight"):
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
        model.init_ema()          # eval() swaps the EMA shadow in: make it the re-drawn weights
        model.eval()
        B, D = 3, cfg.model.concat_dim
        x = torch.randint(0, S, (B, D), generator=g)
        t = torch.tensor([0.05, 0.5, 0.97])
        with torch.no_grad():
            logits = model(x, t)
        sd = {k: v for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)}
        arrs.update({f"{tag}__x": x, f"{tag}__t": t, f"{tag}__out": logits})
        arrs.update({f"{tag}__sd__{k}": v for k, v in sd.items()})
        arrs[f"{tag}__cfg"] = np.array(repr(dict(ch=cfg.model.ch, ch_mult=list(cfg.model.ch_mult), n_res_blocks=1,
                                                 num_heads=2, x_min_max=[0, S - 1], model_output=mo, S=S,
                                                 data_shape=list(cfg.data.shape))))
    save("unet", **arrs)


# ------------------------------------------------------------------------------------ P7 (losses)
def gen_losses():
    """Reference losses on the toy score function under a fixed seed.  The noised states the
    reference drew are captured from the model calls; E noise is regenerated in its RNG order so
    that the oracle can rebuild x_t / x~ itself.  d(loss)/d(scale) pins the gradient path."""
    arrs = {}
    cases = [
        # tag, loss, kind, S, B, D, overrides
        ("ctelbo_g16", "CTElbo", "gaussian", 16, 4, 12, dict(nll_weight=0.3)),
        ("ctelbo2_g16", "CTElbo", "gaussian", 16, 4, 12, dict(one_forward_pass=False, nll_weight=0.0)),
        ("nll_g16", "NLL", "gaussian", 16, 4, 12, dict()),
        ("lambda_g16", "CTElboLambda", "gaussian", 16, 4, 12, dict(n_iter=250)),
        ("ctelbo_g256", "CTElbo", "gaussian", 256, 2, 9, dict(nll_weight=0.001)),
        ("catrm_v3_rm", "CatRM", "univar", 3, 5, 15, dict(logit_type="reverse_prob", loss_type="rm")),
        ("catrm_v3_mle", "CatRM", "univar", 3, 5, 15, dict(logit_type="direct", loss_type="mle")),
        ("catrm_v3_elbo", "CatRM", "univar", 3, 5, 15, dict(logit_type="reverse_logscale", loss_type="elbo", ce_coeff=0.25)),
        ("catrmnll_v2", "CatRMNLL", "univar", 2, 6, 32, dict(logit_type="reverse_prob", loss_type="rm", nll_weight=0.01, t_func="log_sqr", max_t=0.99999)),
        ("catrmnll_g16_elbo", "CatRMNLL", "gaussian", 16, 3, 10, dict(logit_type="direct", loss_type="elbo", nll_weight=0.1)),
        ("nllorig_v3", "NLLOriginal", "univar", 3, 5, 15, dict()),
        ("score_v3", "ScoreElbo", "univar", 3, 5, 15, dict(logit_type="reverse_prob", nll_weight=0.01)),
        ("score_g16", "ScoreElbo", "gaussian", 16, 4, 12, dict(logit_type="direct", nll_weight=0.5, one_forward_pass=False)),
    ]
    for tag, lname, kind, S, B, D, ov in cases:
        cfg = base_cfg(S, D)
        cfg.loss.name = lname
        for k in ("nll_weight", "one_forward_pass", "logit_type", "loss_type", "ce_coeff"):
            if k in ov:
                cfg.loss[k] = ov[k]
        cfg.model.t_func = ov.get("t_func", "sqrt_cos")
        cfg.training.max_t = ov.get("max_t", 1.0)
        model = make_ref_model(kind, cfg)
        theta = torch.tensor(1.5, requires_grad=True)
        model.calls = []
        base_call = model.__class__.__call__

        def call(self, x, t, *a, _th=theta, _S=S):
            self.calls.append((x.clone(), t.clone()))
            return toy_logits(x, t, _S, 1.0) * _th
        model.__class__.__call__ = call
        loss = getattr(ref_losses, lname)(cfg)
        g = torch.Generator().manual_seed(17)
        x0 = torch.randint(0, S, (B, D), generator=g)
        state = {"model": model, "n_iter": ov.get("n_iter", 0)}
        seed = 777
        torch.manual_seed(seed)
        old_order = lname in ("CatRMNLL", "ScoreElbo")
        val = loss.calc_loss(x0.clone(), state) if old_order else loss.calc_loss(state, x0.clone())
        grad, = torch.autograd.grad(val, theta)
        ts = model.calls[0][1]
        # regenerate the noise in the reference's RNG order
        torch.manual_seed(seed)
        u = torch.rand((B,))
        E_xt = torch.empty(B * D, S).exponential_(1)
        extra = {}
        if lname in ("CTElbo", "NLL", "CTElboLambda", "ScoreElbo"):
            extra = {"E_dim": torch.empty(B, D).exponential_(1), "E_val": torch.empty(B, S).exponential_(1)}
        meta = dict(loss=lname, kind=kind, S=S, B=B, D=D, t_func=cfg.model.t_func, max_t=cfg.training.max_t,
                    n_iter=state["n_iter"], n_iters=cfg.training.n_iters, theta=1.5,
                    **{k: cfg.loss[k] for k in ("eps_ratio", "nll_weight", "min_time", "one_forward_pass", "logit_type", "loss_type", "ce_coeff")})
        arrs[f"{tag}__meta"] = np.array(repr(meta))
        arrs.update({f"{tag}__x0": x0, f"{tag}__u": u, f"{tag}__ts": ts, f"{tag}__E_xt": E_xt, f"{tag}__loss": val.detach(),
                     f"{tag}__grad": grad, **{f"{tag}__{k}": v for k, v in extra.items()}})
        arrs[f"{tag}__x_first"] = model.calls[0][0]
        model.__class__.__call__ = base_call
    save("losses", **arrs)


# ------------------------------------------------------------------------------------ P8 (hollow transformer)
def gen_hollow():
    """Tiny reference hollow transformers (S=3 maze-like and S=2 synthetic-like), re-drawn weights."""
    import lib.models.models  # noqa: F401
    arrs = {}
    for tag, S, D, E, layers, mlp, mname, tf in (("s3", 3, 12, 32, 2, 48, "UniVarHollowEMA", "sqrt_cos"),
                                                  ("s2", 2, 16, 16, 1, 32, "UniVarHollowEMA", "log_sqr")):
        cfg = base_cfg(S, D, mname)
        cfg.model.update(dict(t_func=tf, net_arch="bidir_transformer", nets="bidir_transformer2", use_cat=False,
                              embed_dim=E, bidir_readout="attention", use_one_hot_input=False, dropout_rate=0.1,
                              num_layers=layers, num_heads=4, attention_dropout_rate=0.1, transformer_norm_type="prenorm",
                              mlp_dim=mlp, out_dim=None, readout_dim=S, num_output_ffresiduals=2, qkv_dim=E,
                              ema_decay=0.999, time_scale_factor=1000, is_ebm=False))
        torch.manual_seed(21)
        model = ref_mu.create_model(cfg, torch.device("cpu"))
        g = torch.Generator().manual_seed(22)
        with torch.no_grad():
            for name, p in model.named_parameters():
                if p.dim() > 1:
                    p.copy_(torch.randn(p.shape, generator=g) * (1.0 / math.sqrt(p.shape[-1])))
                elif "norm" in name and name.endswith("weight") or ".ln" in name and name.endswith("weight") or "resid_layers.1.weight" in name or "resid_layers.3.weight" in name:
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
        model.init_ema()
        model.eval()
        x = torch.randint(0, S, (3, D), generator=g)
        t = torch.tensor([0.05, 0.5, 0.97])
        with torch.no_grad():
            out = model(x, t)
        sd = {k: v for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)}
        arrs.update({f"{tag}__x": x, f"{tag}__t": t, f"{tag}__out": out})
        arrs.update({f"{tag}__sd__{k}": v for k, v in sd.items()})
        arrs[f"{tag}__cfg"] = np.array(repr(dict(S=S, D=D, embed_dim=E, num_layers=layers, num_heads=4, mlp_dim=mlp,
                                                 time_scale_factor=1000, t_func=tf)))
    save("hollow", **arrs)


GROUPS = {"forward_process": gen_forward_process, "noising": gen_noising, "rates": gen_rates,
          "samplers": gen_samplers, "unet": gen_unet, "losses": gen_losses, "hollow": gen_hollow}

if __name__ == "__main__":
    names = sys.argv[1:] or list(GROUPS)
    for n in names:
        GROUPS[n]()
