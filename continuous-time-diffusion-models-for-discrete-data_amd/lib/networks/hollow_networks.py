"""SDDM hollow (bidirectional-causal) transformer (reference lib/networks/hollow_networks.py:
BidirectionalTransformer2 668-755, UniDirectionalTransformer 497-568, TransformerBlock 423-447,
SelfAttentionBlock 311-340, FeedForwardBlock / TransformerMlpBlock 343-420, AttentionReadout
283-308, CrossAttention 204-280, ResidualReadout 90-132, PositionalEncoding 1136-1156).

Token d of the output sees x_0..x_{d-1} through the left-to-right stack and x_{d+1}..x_{D-1} through
the right-to-left stack, never x_d itself ("hollow"), so logits[d] parameterises
p(x^d | x^{\\d}).  The module tree keeps the reference's parameter names (including the two
sub-modules its forward never uses, `embedding` and `temb_net`) so checkpoints load unchanged.
Device ops (autograd-capable); the HIP inference plan is ctdd/hollow_engine.py, the HIP training path ctdd/hollow_train.py.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def transformer_timestep_embedding(timesteps, embedding_dim, device="cpu", max_positions=10000):
    assert embedding_dim % 2 == 0 and timesteps.dim() == 1
    half = embedding_dim // 2
    freq = torch.exp(torch.arange(half, device=timesteps.device, dtype=torch.float32) * -(math.log(max_positions) / (half - 1)))
    arg = timesteps[:, None] * freq[None, :]
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)


def normalize_input(x, S):
    return (x / (S - 1)) * 2 - 1


class MLP(nn.Module):
    """Linear stack with an activation between layers (names: layers.0, layers.2, ...)."""

    def __init__(self, features, activation=nn.ReLU):
        super().__init__()
        self.features, self.activation = features, activation
        mods = []
        for i in range(len(features) - 1):
            mods.append(nn.Linear(features[i], features[i + 1]))
            if i != len(features) - 2:
                mods.append(self.activation)
        self.layers = nn.Sequential(*mods)

    def forward(self, x):
        return self.layers(x)


class PositionalEncoding(nn.Module):
    def __init__(self, device, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pos = torch.arange(max_len, device=device).unsqueeze(1)
        div = torch.exp(torch.arange(0, d_model, 2, device=device) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model, device=device)
        pe[0, :, 0::2] = torch.sin(pos * div)
        pe[0, :, 1::2] = torch.cos(pos * div)
        self.pe = pe                                   # plain attribute, as in the reference (not a buffer)

    def forward(self, x):
        return self.dropout(x + self.pe[:, : x.size(1), :].to(x.device))


class SelfAttentionBlock(nn.Module):
    def __init__(self, config):
        super().__init__()
        m = config.model
        self.prenorm = m.transformer_norm_type == "prenorm"
        self.self_attention = nn.MultiheadAttention(embed_dim=m.embed_dim, num_heads=m.num_heads,
                                                    dropout=m.attention_dropout_rate, batch_first=True)
        self.dropout = nn.Dropout(m.dropout_rate)
        self.norm = nn.LayerNorm(m.embed_dim)

    def forward(self, inputs, masks):
        if self.prenorm:
            x = self.norm(inputs)
            x, _ = self.self_attention(x, x, x, attn_mask=masks, need_weights=False)
            return self.dropout(x) + inputs
        x, _ = self.self_attention(inputs, inputs, inputs, attn_mask=masks, need_weights=False)
        return self.norm(self.dropout(x) + inputs)


class TransformerMlpBlock(nn.Module):
    """Linear(E->mlp) ReLU Dropout Linear(mlp->out, no bias) Dropout."""

    def __init__(self, mlp_dim, embed_dim, out_dim=None, dropout_rate=0.0):
        super().__init__()
        self.fc1 = nn.Linear(embed_dim, mlp_dim)
        self.activation = nn.ReLU()
        self.dropout1 = nn.Dropout(p=dropout_rate)
        self.fc2 = nn.Linear(mlp_dim, out_dim if out_dim is not None else embed_dim, bias=False)
        self.dropout2 = nn.Dropout(p=dropout_rate)
        nn.init.xavier_uniform_(self.fc1.weight)
        nn.init.xavier_uniform_(self.fc2.weight)

    def forward(self, x):
        return self.dropout2(self.fc2(self.dropout1(self.activation(self.fc1(x)))))


class FeedForwardBlock(nn.Module):
    def __init__(self, config):
        super().__init__()
        m = config.model
        self.prenorm = m.transformer_norm_type == "prenorm"
        self.mlp = TransformerMlpBlock(mlp_dim=m.mlp_dim, dropout_rate=m.dropout_rate, embed_dim=m.embed_dim)
        self.norm = nn.LayerNorm(m.embed_dim)

    def forward(self, x):
        return x + self.mlp(self.norm(x)) if self.prenorm else self.norm(x + self.mlp(x))


class TransformerBlock(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self_attention_block = SelfAttentionBlock(config)
        self.feed_forward_block = FeedForwardBlock(config)

    def forward(self, inputs, masks):
        return self.feed_forward_block(self.self_attention_block(inputs, masks))


class UniDirectionalTransformer(nn.Module):
    """One causal direction over [temb, x_0..x_{D-2}] (l2r) or [x_1..x_{D-1}, temb] (r2l)."""

    def __init__(self, config, direction):
        super().__init__()
        m = config.model
        self.direction = direction
        self.dropout = nn.Dropout(m.dropout_rate)
        self.trans_block_layers = nn.ModuleList([TransformerBlock(config) for _ in range(m.num_layers)])
        self.pos_embed = PositionalEncoding(config.device, m.embed_dim, m.dropout_rate, m.concat_dim)

    def forward(self, x, temb, conditioner=None):
        temb = temb.unsqueeze(1)
        cond = temb if conditioner is None else torch.cat([conditioner, temb], dim=1)
        L = x.size(1) + cond.size(1) - 1
        blocked = torch.ones((L, L), device=x.device, dtype=torch.bool)
        if self.direction == "l2r":
            x = torch.cat([cond, x[:, :-1]], dim=1)
            blocked = torch.triu(blocked, diagonal=1)
        else:
            x = torch.cat([x[:, 1:], cond], dim=1)
            blocked = torch.tril(blocked, diagonal=-1)
        mask = torch.zeros((L, L), device=x.device).masked_fill(blocked, float("-inf"))
        x = self.dropout(self.pos_embed(x))
        for blk in self.trans_block_layers:
            x = blk(x, masks=mask)
        return x


class CrossAttention(nn.Module):
    """Readout attention: query l2r+r2l over keys [temb | l2r (j<=i) | r2l (j>=i)]."""

    def __init__(self, config):
        super().__init__()
        m = config.model
        self.num_heads = m.num_heads
        self.head_dim = m.qkv_dim // m.num_heads
        self.dense_query = nn.Linear(m.qkv_dim, self.num_heads * self.head_dim, bias=False)
        self.dense_key = nn.Linear(m.qkv_dim, self.num_heads * self.head_dim)
        self.dense_val = nn.Linear(m.qkv_dim, self.num_heads * self.head_dim)
        self.out_linear = nn.Linear(m.qkv_dim, m.embed_dim)

    def forward(self, l2r, r2l, temb):
        B, D, _ = l2r.shape
        H, hd = self.num_heads, self.head_dim
        allk = torch.cat([temb.unsqueeze(1), l2r, r2l], dim=1)                       # (B, 2D+1, E)
        q = self.dense_query(l2r + r2l).view(B, D, H, hd) / math.sqrt(hd)
        k = self.dense_key(allk).view(B, 2 * D + 1, H, hd)
        v = self.dense_val(allk).view(B, 2 * D + 1, H, hd)
        logits = torch.einsum("bqhd,bkhd->bhqk", q, k)
        ones = torch.ones((D, D), device=l2r.device, dtype=torch.bool)
        allow = torch.cat([torch.ones((D, 1), device=l2r.device, dtype=torch.bool), torch.tril(ones), torch.triu(ones)], dim=-1)
        w = F.softmax(logits.masked_fill(~allow.view(1, 1, D, 2 * D + 1), torch.finfo(logits.dtype).min), dim=-1)
        x = torch.einsum("bhqk,bkhd->bqhd", w, v).reshape(B, D, H * hd)
        return self.out_linear(x)


def apply_film(film_params, x):
    a, b = torch.chunk(film_params.unsqueeze(1), 2, dim=-1)
    return a * x + b


class ResidualReadout(nn.Module):
    def __init__(self, config, readout_dim=0):
        super().__init__()
        m = config.model
        self.n_res = m.num_output_ffresiduals
        E = m.embed_dim
        self.out_dim = readout_dim if readout_dim != 0 else config.data.S
        self.input_layer = nn.Linear(E, 2 * E)
        self.mlp = MLP([E, m.mlp_dim, 4 * E], activation=nn.GELU())
        resid, film = [], []
        for _ in range(self.n_res):
            resid.append(MLP([2 * E, m.mlp_dim, 2 * E], activation=nn.GELU()))
            resid.append(nn.LayerNorm(2 * E))
            film.append(nn.Linear(4 * E, 4 * E))
        self.resid_layers = nn.ModuleList(resid)
        self.film_layer = nn.ModuleList(film)
        self.logits_layer = nn.Linear(2 * E, self.out_dim)

    def forward(self, x, temb):
        temb = self.mlp(temb)
        x = self.input_layer(x)
        for i in range(self.n_res):
            x = self.resid_layers[2 * i + 1](x + self.resid_layers[2 * i](x))
            x = apply_film(self.film_layer[i](temb), x)
        return self.logits_layer(x)


class AttentionReadout(nn.Module):
    def __init__(self, config, readout_dim=0):
        super().__init__()
        self.prenorm = config.model.transformer_norm_type == "prenorm"
        if config.model.transformer_norm_type not in ("prenorm", "postnorm"):
            raise ValueError("unknown norm type %s" % config.model.transformer_norm_type)
        self.cross_attention = CrossAttention(config)
        self.model = ResidualReadout(config, readout_dim)
        self.ln1 = nn.LayerNorm(config.model.embed_dim)
        self.ln2 = nn.LayerNorm(config.model.embed_dim)

    def forward(self, l2r, r2l, temb):
        inputs = l2r + r2l
        if self.prenorm:
            x = self.cross_attention(self.ln1(l2r), self.ln2(r2l), temb) + inputs
        else:
            x = self.ln1(self.cross_attention(l2r, r2l, temb) + inputs)
        return self.model(x, temb)


class BidirectionalTransformer2(nn.Module):
    def __init__(self, config, readout_dim=None):
        super().__init__()
        m = config.model
        self.config = config
        self.S, self.embed_dim, self.mlp_dim = config.data.S, m.embed_dim, m.mlp_dim
        self.embedding = nn.Embedding(self.S, m.embed_dim)            # unused by forward (kept for checkpoints)
        self.temb_scale = m.time_scale_factor
        self.use_cat, self.use_one_hot_input = m.use_cat, m.use_one_hot_input
        if m.net_arch != "bidir_transformer":
            raise ValueError("Unknown net_arch: %s" % m.net_arch)
        self.module_l2r = UniDirectionalTransformer(config, "l2r")
        self.module_r2l = UniDirectionalTransformer(config, "r2l")
        self.readout_dim = self.S if readout_dim is None else readout_dim
        if m.bidir_readout != "attention":
            raise ValueError(f"only bidir_readout='attention' is built (got {m.bidir_readout})")
        self.readout_module = AttentionReadout(config, readout_dim=self.readout_dim)
        if self.use_cat:
            self.input_embedding = nn.Linear(self.S, self.embed_dim) if self.use_one_hot_input else nn.Embedding(self.S, self.embed_dim)
        else:
            self.input_embedding = nn.Linear(1, self.embed_dim)
        self.temb_net = nn.Sequential(nn.Linear(int(self.embed_dim / 2), self.mlp_dim), nn.ReLU(),
                                      nn.Linear(self.mlp_dim, self.embed_dim))    # unused by forward

    def forward(self, x, t):
        hook = getattr(self, "_engine_hook", None)       # the HIP training path (ctdd/hollow_train.py), entered through
        if hook is not None:                             # this forward so a DistributedDataParallel wrapper sees the call
            return hook(x, t)
        temb = transformer_timestep_embedding(t * self.temb_scale, self.embed_dim)
        B, D = x.shape
        if self.use_cat:
            x_embed = self.input_embedding(F.one_hot(x.long(), self.S).float()) if self.use_one_hot_input else self.input_embedding(x.long())
        else:
            x_embed = self.input_embedding(normalize_input(x.float(), self.S).view(B, D, 1))
        l2r = self.module_l2r(x_embed, temb)
        r2l = self.module_r2l(x_embed, temb)
        return self.readout_module(l2r, r2l, temb).view(B, D, self.readout_dim)
