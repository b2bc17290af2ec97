"""Frechet distance between two image sets in a feature space (reference lib/datasets/mnist_fid.py:21-192).

    FID = |mu1 - mu2|^2 + Tr(S1 + S2 - 2 (S1 S2)^(1/2))

over the activation statistics (mean, covariance) of a feature network.  The reference hard-wires the FID InceptionV3
(pool_3, 2048-d) whose weights it downloads at first use (lib/datasets/mnist_is.py:186-210) and whose blocks subclass
torchvision's; neither exists offline, so the feature extractor is an argument here: any module mapping (B, 3, H, W)
floats in [0, 1] to (B, dims) or (B, dims, h, w) activations (or a list whose first entry is that, as the reference's
InceptionV3 returns).  Everything downstream of the activations follows the reference: fp64 statistics, np.cov,
scipy sqrtm with the eps * I retry for singular products and the 1e-3 bound on imaginary parts."""
import warnings

import numpy as np
import torch
from scipy import linalg


def get_activations(images, model, batch_size=50, dims=2048, device="cuda"):
    """(N, dims) float64 activations of `model` over `images` (array or tensor (N, C, H, W)), spatial maps averaged
    (mnist_fid.py:21-71).  The last, smaller batch is kept."""
    model.eval()
    n = len(images)
    batch_size = min(batch_size, n)
    out = np.empty((n, dims))
    images = torch.as_tensor(np.asarray(images)) if not torch.is_tensor(images) else images
    with torch.no_grad():
        for i0 in range(0, n, batch_size):
            pred = model(images[i0:i0 + batch_size].to(device).float())
            if isinstance(pred, (list, tuple)):
                pred = pred[0]
            if pred.dim() == 4:
                pred = pred.mean(dim=(2, 3))
            out[i0:i0 + pred.shape[0]] = pred.double().cpu().numpy()
    return out


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """Frechet distance between N(mu1, sigma1) and N(mu2, sigma2) (mnist_fid.py:74-128)."""
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    if mu1.shape != mu2.shape or sigma1.shape != sigma2.shape:
        raise ValueError("mean / covariance shapes of the two sets differ")
    diff = mu1 - mu2
    root, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(root).all():
        warnings.warn(f"fid: singular covariance product; adding {eps} to the diagonals")
        off = np.eye(sigma1.shape[0]) * eps
        root = linalg.sqrtm((sigma1 + off).dot(sigma2 + off))
    if np.iscomplexobj(root):
        if not np.allclose(np.diagonal(root).imag, 0, atol=1e-3):
            raise ValueError(f"Imaginary component {np.max(np.abs(root.imag))}")
        root = root.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2.0 * np.trace(root))


def calculate_activation_statistics(images, model, batch_size=50, dims=2048, device="cpu"):
    act = get_activations(images, model, batch_size, dims, device)
    return np.mean(act, axis=0), np.cov(act, rowvar=False)


def normalize_input(x, S):
    return x / (S - 1)


def evaluate_fid_score(images1, images2, batch_size=50, model=None, dims=2048, device=None, S=256):
    """FID between two sets of (N, 1, H, W) images with states 0..S-1: scaled to [0, 1], grey tiled to 3 channels, then
    the Frechet distance of the activation statistics (mnist_fid.py:156-192)."""
    if model is None:
        raise RuntimeError("evaluate_fid_score needs a feature network: the reference's FID InceptionV3 weights are a "
                           "download (mnist_is.py:186-210) and are not available offline; pass model=<extractor>, dims=<width>")
    device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
    sets = []
    for im in (images1, images2):
        im = normalize_input(np.asarray(im, dtype=np.float64), S)
        im = np.tile(im, (1, 3, 1, 1))
        if im.shape[-1] == 1:
            im = np.concatenate([im, im, im], axis=-1)
        sets.append(im)
    if any(s.max() > 1 or s.min() < 0 for s in sets):
        warnings.warn("FID score: the values of images should be in range [0,1].")
    model = model.to(device)
    m1, s1 = calculate_activation_statistics(sets[0], model, batch_size, dims, device)
    m2, s2 = calculate_activation_statistics(sets[1], model, batch_size, dims, device)
    return calculate_frechet_distance(m1, s1, m2, s2)
