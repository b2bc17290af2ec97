"""Oracle (test infrastructure): score networks as plain functional fp32 torch-CPU code driven by
a reference-format state_dict (a torch fp32 reference for floating-point kernels).

U-Net: restates TAUnSDDM/lib/networks/unet.py:303-459 (+ blocks 79-241) and the wrapper
lib/models/models.py:225-292.  No module objects: the weights are looked up by their reference
parameter names, so a reference checkpoint or the golden fixture drives it directly.
"""
import math

import torch
import torch.nn.functional as F


def _swish(x):
    return x * torch.sigmoid(x)


def _gn(x, sd, key, eps):
    w = sd[key + ".weight"]
    return F.group_norm(x, min(w.numel() // 4, 32), w, sd[key + ".bias"], eps)


def _conv(x, sd, key, stride=1, padding=1):
    return F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], stride=stride, padding=padding)


def _lin(x, sd, key):
    return F.linear(x, sd[key + ".weight"], sd[key + ".bias"])


def _resblock(x, temb, sd, pre):
    h = _conv(_swish(_gn(x, sd, pre + ".norm1", 1e-6)), sd, pre + ".conv1")
    h = h + _lin(_swish(temb), sd, pre + ".time.1")[:, :, None, None]
    h = _conv(_swish(_gn(h, sd, pre + ".norm2", 1e-6)), sd, pre + ".conv2")      # dropout = identity (eval)
    if pre + ".skip.weight" in sd:
        x = _lin(x.permute(0, 2, 3, 1), sd, pre + ".skip").permute(0, 3, 1, 2)
    return h + x


def _attention(x, sd, pre, heads):
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(xf, sd, pre + ".norm", 1e-5), sd[pre + ".qkv.weight"], sd[pre + ".qkv.bias"])
    qkv = qkv.reshape(b * heads, -1, xf.shape[-1])
    ch = qkv.shape[1] // 3
    q, k, v = torch.split(qkv, ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, xf.shape[-1])
    a = F.conv1d(a, sd[pre + ".proj_out.weight"], sd[pre + ".proj_out.bias"])
    return (xf + a).reshape(b, c, hh, ww)


def unet_forward(sd, x, t, *, ch, ch_mult, n_res_blocks, num_heads, x_min_max, model_output, S, prefix="net."):
    """x (B,C,H,W) float, t (B,).  Returns (B,C,H,W,S) logits or (mu, log_scale)."""
    sd = {k[len(prefix):]: v for k, v in sd.items() if isinstance(v, torch.Tensor) and k.startswith(prefix)}
    half = ch // 2
    inv_freq = torch.exp(torch.arange(half, dtype=torch.float32) * (-math.log(10000) / (half - 1)))
    arg = torch.outer(t.float(), inv_freq)
    temb = torch.cat([arg.sin(), arg.cos()], dim=-1)
    temb = _lin(_swish(_lin(temb, sd, "time.1")), sd, "time.3")
    lo, hi = x_min_max
    h = x0 = 2 * ((x.float() - lo) / (hi - lo)) - 1
    B, C, H, W = x.shape
    feats = []
    i = 0
    h = _conv(h, sd, "down.0")
    feats.append(h)
    i = 1
    levels = len(ch_mult)
    for lv in range(levels):
        for _ in range(n_res_blocks):
            h = _resblock(h, temb, sd, f"down.{i}.resblocks")
            if f"down.{i}.attention.qkv.weight" in sd:
                h = _attention(h, sd, f"down.{i}.attention", num_heads)
            feats.append(h)
            i += 1
        if lv != levels - 1:
            h = _conv(F.pad(h, [0, 1, 0, 1]), sd, f"down.{i}.downsample.0", stride=2, padding=0)
            feats.append(h)
            i += 1
    h = _resblock(h, temb, sd, "mid.0.resblocks")
    h = _attention(h, sd, "mid.0.attention", num_heads)
    h = _resblock(h, temb, sd, "mid.1.resblocks")
    i = 0
    for lv in reversed(range(levels)):
        for _ in range(n_res_blocks + 1):
            h = _resblock(torch.cat((h, feats.pop()), 1), temb, sd, f"up.{i}.resblocks")
            if f"up.{i}.attention.qkv.weight" in sd:
                h = _attention(h, sd, f"up.{i}.attention", num_heads)
            i += 1
        if lv != 0:
            h = _conv(F.interpolate(h, scale_factor=2, mode="nearest"), sd, f"up.{i}.1")
            i += 1
    out = _conv(_swish(_gn(h, sd, "out.0", 1e-6)), sd, "out.2")
    if model_output == "logistic_pars":
        loc, log_scale = torch.chunk(out, 2, dim=1)
        return torch.tanh(loc + x0), log_scale
    return out.reshape(B, C, S, H, W).permute(0, 1, 3, 4, 2).contiguous()


def logistic_logits(mu, log_scale, S, fix_logistic):
    """models.py:249-283: mu/log_scale (B,C,H,W) -> (B,C,H,W,S)."""
    mu, log_scale = mu.unsqueeze(-1), log_scale.unsqueeze(-1)
    inv_scale = torch.exp(-(log_scale - 2))
    bw = 2.0 / S
    centres = torch.linspace(-1.0 + bw / 2, 1.0 - bw / 2, S).view(1, 1, 1, 1, S)
    left = (centres - bw / 2 - mu) * inv_scale
    right = (centres + bw / 2 - mu) * inv_scale
    cl, cr = F.logsigmoid(left), F.logsigmoid(right)
    lme = lambda a, b: a + torch.log1p(-torch.exp(b - a) + 1e-6)
    l1 = lme(cr, cl)
    if fix_logistic:
        return torch.min(l1, lme(-left + cl, -right + cr))
    return l1


def image_model_forward(sd, x, t, *, data_shape, S, model_output, fix_logistic=False, **unet_kw):
    """ImageX0PredBasePaul.forward (models.py:225-292), no padding: x (B,D) -> (B,D,S)."""
    B = x.shape[0]
    C, H, W = data_shape
    out = unet_forward(sd, x.view(B, C, H, W), t, S=S, model_output=model_output, **unet_kw)
    if model_output == "logits":
        return out.reshape(B, C * H * W, S)
    return logistic_logits(out[0], out[1], S, fix_logistic).reshape(B, C * H * W, S)


# =========================================================================== hollow transformer
def _ln(x, sd, key):
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], 1e-5)


def _mha(x, sd, pre, heads, mask):
    """nn.MultiheadAttention(batch_first) self-attention with an additive (L,L) mask, eval mode."""
    B, L, E = x.shape
    qkv = F.linear(x, sd[pre + ".in_proj_weight"], sd[pre + ".in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    hd = E // heads
    sh = lambda z: z.view(B, L, heads, hd).transpose(1, 2)
    w = torch.softmax(sh(q) @ sh(k).transpose(-1, -2) / math.sqrt(hd) + mask, dim=-1)
    o = (w @ sh(v)).transpose(1, 2).reshape(B, L, E)
    return F.linear(o, sd[pre + ".out_proj.weight"], sd[pre + ".out_proj.bias"])


def _positional(L, E):
    pos = torch.arange(L).unsqueeze(1)
    div = torch.exp(torch.arange(0, E, 2) * (-math.log(10000.0) / E))
    pe = torch.zeros(L, E)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def _unidir(x_embed, temb, sd, pre, direction, layers, heads):
    """UniDirectionalTransformer.forward (hollow_networks.py:520-568), prenorm, eval mode."""
    B, D, E = x_embed.shape
    t = temb.unsqueeze(1)
    if direction == "l2r":
        x = torch.cat([t, x_embed[:, :-1]], dim=1)
        blocked = torch.triu(torch.ones(D, D, dtype=torch.bool), diagonal=1)
    else:
        x = torch.cat([x_embed[:, 1:], t], dim=1)
        blocked = torch.tril(torch.ones(D, D, dtype=torch.bool), diagonal=-1)
    mask = torch.zeros(D, D).masked_fill(blocked, float("-inf"))
    x = x + _positional(D, E)
    for i in range(layers):
        b = f"{pre}.trans_block_layers.{i}"
        x = x + _mha(_ln(x, sd, b + ".self_attention_block.norm"), sd, b + ".self_attention_block.self_attention", heads, mask)
        z = _ln(x, sd, b + ".feed_forward_block.norm")
        z = F.linear(F.relu(F.linear(z, sd[b + ".feed_forward_block.mlp.fc1.weight"], sd[b + ".feed_forward_block.mlp.fc1.bias"])),
                     sd[b + ".feed_forward_block.mlp.fc2.weight"])
        x = x + z
    return x


def hollow_forward(sd, x, t, *, S, embed_dim, num_layers, num_heads, time_scale_factor, num_output_ffresiduals=2,
                   prefix="net."):
    """BidirectionalTransformer2.forward (hollow_networks.py:726-755) with use_cat=False, prenorm,
    attention readout (CrossAttention 204-280 + ResidualReadout 90-132); dropout off."""
    sd = {k[len(prefix):]: v for k, v in sd.items() if isinstance(v, torch.Tensor) and k.startswith(prefix)}
    E, B, D = embed_dim, x.shape[0], x.shape[1]
    half = E // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    arg = (t.float() * time_scale_factor)[:, None] * freq[None, :]
    temb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    xn = (x.float() / (S - 1)) * 2 - 1
    x_embed = F.linear(xn.view(B, D, 1), sd["input_embedding.weight"], sd["input_embedding.bias"])
    l2r = _unidir(x_embed, temb, sd, "module_l2r", "l2r", num_layers, num_heads)
    r2l = _unidir(x_embed, temb, sd, "module_r2l", "r2l", num_layers, num_heads)
    # attention readout
    ro = "readout_module"
    inputs = l2r + r2l
    a, b = _ln(l2r, sd, ro + ".ln1"), _ln(r2l, sd, ro + ".ln2")
    H = num_heads
    hd = E // H
    allk = torch.cat([temb.unsqueeze(1), a, b], dim=1)
    q = F.linear(a + b, sd[ro + ".cross_attention.dense_query.weight"]).view(B, D, H, hd) / math.sqrt(hd)
    k = F.linear(allk, sd[ro + ".cross_attention.dense_key.weight"], sd[ro + ".cross_attention.dense_key.bias"]).view(B, 2 * D + 1, H, hd)
    v = F.linear(allk, sd[ro + ".cross_attention.dense_val.weight"], sd[ro + ".cross_attention.dense_val.bias"]).view(B, 2 * D + 1, H, hd)
    logits = torch.einsum("bqhd,bkhd->bhqk", q, k)
    ones = torch.ones(D, D, dtype=torch.bool)
    allow = torch.cat([torch.ones(D, 1, dtype=torch.bool), torch.tril(ones), torch.triu(ones)], dim=-1)
    w = torch.softmax(torch.where(allow.view(1, 1, D, -1), logits, torch.tensor(torch.finfo(torch.float32).min)), dim=-1)
    att = torch.einsum("bhqk,bkhd->bqhd", w, v).reshape(B, D, H * hd)
    z = F.linear(att, sd[ro + ".cross_attention.out_linear.weight"], sd[ro + ".cross_attention.out_linear.bias"]) + inputs
    # residual readout with FiLM
    m = ro + ".model"
    lin = lambda u, key: F.linear(u, sd[key + ".weight"], sd[key + ".bias"])
    tt = lin(F.gelu(lin(temb, m + ".mlp.layers.0")), m + ".mlp.layers.2")
    z = lin(z, m + ".input_layer")
    for i in range(num_output_ffresiduals):
        r = lin(F.gelu(lin(z, f"{m}.resid_layers.{2 * i}.layers.0")), f"{m}.resid_layers.{2 * i}.layers.2")
        z = _ln(z + r, sd, f"{m}.resid_layers.{2 * i + 1}")
        fa, fb = torch.chunk(lin(tt, f"{m}.film_layer.{i}").unsqueeze(1), 2, dim=-1)
        z = fa * z + fb
    return lin(z, m + ".logits_layer")
