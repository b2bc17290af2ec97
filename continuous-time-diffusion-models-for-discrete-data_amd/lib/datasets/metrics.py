"""Sample-quality metric for the binary toy data (reference lib/datasets/metrics.py:6-56, 168-222): the
unbiased MMD^2 estimate with the exponential Hamming kernel k(x,y) = exp(-bd * |x - y|_1), evaluated on the
device in row blocks (the reference materialises (N, M, D) differences)."""
import torch


def exp_hamming_gram_sum(x, y, bandwidth, skip_diagonal=False, block=1024):
    """sum_{i,j} exp(-bd * sum_d |x_id - y_jd|), optionally without the i == j terms (x is y)."""
    x, y = x.float(), y.float()
    total = torch.zeros((), dtype=torch.float64, device=x.device)
    for i in range(0, x.shape[0], block):
        d = torch.cdist(x[i:i + block], y, p=1)
        k = torch.exp(-bandwidth * d).double()
        if skip_diagonal:
            idx = torch.arange(i, min(i + block, x.shape[0]), device=x.device)
            k[idx - i, idx] = 0.0
        total += k.sum()
    return total


def binary_exp_hamming_mmd(x, y, cfg=None, bandwidth=0.1):
    n, m = x.shape[0], y.shape[0]
    kxx = exp_hamming_gram_sum(x, x, bandwidth, True) / n / (n - 1)
    kyy = exp_hamming_gram_sum(y, y, bandwidth, True) / m / (m - 1)
    kxy = exp_hamming_gram_sum(x, y, bandwidth) / n / m
    return (kxx + kyy - 2 * kxy).float()


def eval_mmd(config, model, sampler, dataloader, n_rounds=10, n_samples=1024):
    """Mean MMD^2 between `n_samples` data rows and as many samples drawn with `sampler.sample(model, n)`."""
    avg = 0.0
    with torch.no_grad():
        for _ in range(n_rounds):
            rows, have = [], 0
            while have < n_samples:
                for batch in dataloader:
                    b = batch[0] if isinstance(batch, (list, tuple)) else batch
                    rows.append(b.reshape(b.shape[0], -1))
                    have += b.shape[0]
                    if have >= n_samples:
                        break
            gt = torch.cat(rows, 0)[:n_samples].to(config.device)
            out = sampler.sample(model, n_samples)
            x0 = torch.as_tensor(out[0] if isinstance(out, tuple) else out, device=config.device)
            avg = avg + binary_exp_hamming_mmd(gt, x0, config)
    return avg / n_rounds
