"""D3PM-style 2-D U-Net score network (reference lib/networks/unet.py:303-459).

The module tree reproduces the reference's parameter names (`time.1`, `down.N.resblocks.conv1`,
`down.N.downsample.0`, `mid.0.attention.qkv`, `up.N.1`, `out.2`, ...) so its checkpoints load
unchanged.  Two executors read this tree:
  * `forward()` -- differentiable device ops, used for training (autograd);
  * `ctdd.unet_engine.UNetEngine` -- the hand-written HIP inference engine (csrc/unet_kernels.hip)
    that the samplers run under `torch.no_grad()` in eval mode.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

import lib.networks.network_utils as network_utils


def _vs_uniform_(w, scale=1.0):
    """Variance-scaling init, fan_avg, uniform (unet.py:17-38)."""
    fan_in, fan_out = nn.init._calculate_fan_in_and_fan_out(w)
    bound = math.sqrt(3 * scale / ((fan_in + fan_out) / 2))
    with torch.no_grad():
        return w.uniform_(-bound, bound)


def conv3x3(cin, cout, stride=1, padding=1, scale=1.0):
    m = nn.Conv2d(cin, cout, 3, stride=stride, padding=padding)
    _vs_uniform_(m.weight, scale)
    nn.init.zeros_(m.bias)
    return m


def dense(cin, cout, scale=1.0):
    m = nn.Linear(cin, cout)
    _vs_uniform_(m.weight, scale)
    nn.init.zeros_(m.bias)
    return m


def group_norm(ch, eps=1e-6):
    return nn.GroupNorm(num_groups=min(ch // 4, 32), num_channels=ch, eps=eps)


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


class TimeEmbedding(nn.Module):
    """sin/cos features of the RAW time t (no 1000x scaling), unet.py:223-241."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        half = dim // 2
        self.inv_freq = torch.exp(torch.arange(half, dtype=torch.float32) * (-math.log(10000) / (half - 1)))

    def forward(self, t):
        arg = torch.outer(t.reshape(-1).float(), self.inv_freq.to(t.device))
        return torch.cat([arg.sin(), arg.cos()], dim=-1).view(*t.shape, self.dim)


class ResBlock(nn.Module):
    """GN-Swish-conv3x3 (+time proj) GN-Swish-dropout-conv3x3(scale 1e-10) + (linear) skip."""

    def __init__(self, cin, cout, time_dim, dropout):
        super().__init__()
        self.norm1 = group_norm(cin)
        self.activation1 = Swish()
        self.conv1 = conv3x3(cin, cout)
        self.time = nn.Sequential(Swish(), dense(time_dim, cout))
        self.norm2 = group_norm(cout)
        self.activation2 = Swish()
        self.dropout = nn.Dropout(dropout)
        self.conv2 = conv3x3(cout, cout, scale=1e-10)
        self.skip = dense(cin, cout) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(self.activation1(self.norm1(x)))
        h = h + self.time(temb)[:, :, None, None]
        h = self.conv2(self.dropout(self.activation2(self.norm2(h))))
        if self.skip is not None:
            x = self.skip(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return h + x


class SelfAttention(nn.Module):
    """Spatial self-attention with a zero-initialised output projection (unet.py:152-200)."""

    def __init__(self, channels, n_head=1):
        super().__init__()
        self.channels, self.num_heads = channels, n_head
        self.norm = nn.GroupNorm(num_groups=min(channels // 4, 32), num_channels=channels)   # eps 1e-5
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = nn.Conv1d(channels, channels, 1)
        for p in self.proj_out.parameters():
            p.detach().zero_()

    def forward(self, x):
        b, c, *spatial = x.shape
        x = x.reshape(b, c, -1)
        qkv = self.qkv(self.norm(x)).reshape(b * self.num_heads, -1, x.shape[-1])
        ch = qkv.shape[1] // 3
        q, k, v = torch.split(qkv, ch, dim=1)
        s = 1 / math.sqrt(math.sqrt(ch))
        w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s).float(), dim=-1).type(q.dtype)
        a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, x.shape[-1])
        return (x + self.proj_out(a)).reshape(b, c, *spatial)


class ResBlockWithAttention(nn.Module):
    def __init__(self, cin, cout, time_dim, dropout, attention_head=1, use_attention=False):
        super().__init__()
        self.resblocks = ResBlock(cin, cout, time_dim, dropout)
        self.attention = SelfAttention(cout, n_head=attention_head) if use_attention else None

    def forward(self, x, temb):
        h = self.resblocks(x, temb)
        return h if self.attention is None else self.attention(h)


class Downsample(nn.Module):
    """Stride-2 conv on an input padded by one row/column at the bottom/right."""

    def __init__(self, ch):
        super().__init__()
        self.downsample = nn.Sequential(conv3x3(ch, ch, stride=2, padding=0))

    def forward(self, x):
        return self.downsample(F.pad(x, [0, 1, 0, 1]))


class Upsample(nn.Sequential):
    def __init__(self, ch):
        super().__init__(nn.Upsample(scale_factor=2, mode="nearest"), conv3x3(ch, ch))


class UNet(nn.Module):
    def __init__(self, in_channel, out_channel, channel, channel_multiplier, n_res_blocks, attn_resolutions,
                 x_min_max, num_heads, dropout, model_output, num_classes, img_size):
        super().__init__()
        self.model_output, self.S, self.out_channel = model_output, num_classes, out_channel
        self.x_min_max = x_min_max
        self.in_channel, self.channel, self.img_size, self.num_heads = in_channel, channel, img_size, num_heads
        time_dim = channel * 4
        strides = [img_size // int(r) for r in attn_resolutions]
        levels = len(channel_multiplier)
        self.time = nn.Sequential(TimeEmbedding(channel), dense(channel, time_dim), Swish(), dense(time_dim, time_dim))

        down, skips, ch = [conv3x3(in_channel, channel)], [channel], channel
        for lv, mult in enumerate(channel_multiplier):
            for _ in range(n_res_blocks):
                down.append(ResBlockWithAttention(ch, channel * mult, time_dim, dropout, num_heads, 2**lv in strides))
                ch = channel * mult
                skips.append(ch)
            if lv != levels - 1:
                down.append(Downsample(ch))
                skips.append(ch)
        self.down = nn.ModuleList(down)
        self.mid = nn.ModuleList([ResBlockWithAttention(ch, ch, time_dim, dropout, num_heads, True),
                                  ResBlockWithAttention(ch, ch, time_dim, dropout=dropout)])
        up = []
        for lv in reversed(range(levels)):
            for _ in range(n_res_blocks + 1):
                cout = channel * channel_multiplier[lv]
                up.append(ResBlockWithAttention(ch + skips.pop(), cout, time_dim, dropout, num_heads, 2**lv in strides))
                ch = cout
            if lv != 0:
                up.append(Upsample(ch))
        self.up = nn.ModuleList(up)
        n_out = out_channel * 2 if model_output == "logistic_pars" else out_channel * self.S
        self.out = nn.Sequential(group_norm(ch), Swish(), conv3x3(ch, n_out, scale=1e-10))
        self.D = img_size * img_size

    def forward(self, x, t):
        hook = getattr(self, "_engine_hook", None)
        if hook is not None:
            # cfg.distributed: the model wrapper calls this module through DistributedDataParallel.forward (so that the
            # reducer expects this iteration's gradients) and the hand-written training plan takes over from here
            return hook(x, t)
        temb = self.time(t)
        B, C, H, W = x.shape
        h = x0 = network_utils.center_data(x, self.x_min_max)
        stack = []
        for layer in self.down:
            h = layer(h, temb) if isinstance(layer, ResBlockWithAttention) else layer(h)
            stack.append(h)
        for layer in self.mid:
            h = layer(h, temb)
        for layer in self.up:
            if isinstance(layer, ResBlockWithAttention):
                h = layer(torch.cat((h, stack.pop()), 1), temb)
            else:
                h = layer(h)
        out = self.out(h)
        if self.model_output == "logistic_pars":
            loc, log_scale = torch.chunk(out, 2, dim=1)
            return torch.tanh(loc + x0), log_scale
        # (B, C*S, H, W) -> (B, C, H, W, S)
        return out.reshape(B, self.out_channel, self.S, H, W).permute(0, 1, 3, 4, 2).contiguous()
