"""Build-owned test stub (timm absent): placeholders so the reference's unused DiT module imports."""
import torch.nn as nn


class Attention(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


class Mlp(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


class PatchEmbed(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
