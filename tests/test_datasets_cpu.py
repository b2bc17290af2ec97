"""CPU: dataset adapters and the MMD metric.  The reference needs torchvision / absl /
downloads for these modules and was not imported for them: parity unpinned -- format readers are checked
on files written here, the codec by round trips and Gray-code properties, the MMD against the reference's
formula written out with dense (N, M, D) differences."""
import gzip
import struct

import numpy as np
import torch


def _cfg(train=True):
    from ctdd.config_dict import ConfigDict
    c = ConfigDict()
    c.data = ConfigDict()
    c.data.train, c.data.download, c.data.use_augm, c.data.image_size = train, False, False, 28
    return c


def test_discrete_mnist_reads_idx(tmp_path):
    import lib.datasets.mnist as dm
    import lib.datasets.dataset_utils as du
    raw = tmp_path / "MNIST" / "raw"
    raw.mkdir(parents=True)
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, (7, 28, 28), dtype=np.uint8)
    labs = rng.integers(0, 10, (7,), dtype=np.uint8)
    with gzip.open(raw / "train-images-idx3-ubyte.gz", "wb") as f:
        f.write(struct.pack(">HBBIII", 0, 8, 3, 7, 28, 28) + imgs.tobytes())
    with open(raw / "train-labels-idx1-ubyte", "wb") as f:
        f.write(struct.pack(">HBBI", 0, 8, 1, 7) + labs.tobytes())
    cfg = _cfg()
    cfg.data.name = "DiscreteMNIST"
    ds = du.get_dataset(cfg, torch.device("cpu"), str(tmp_path))
    assert len(ds) == 7
    img, t = ds[3]
    assert img.dtype == torch.uint8 and img.shape == (1, 28, 28) and int(t) == int(labs[3])
    np.testing.assert_array_equal(img[0].numpy(), imgs[3])
    cfg.data.use_augm = True
    assert dm.DiscreteMNIST(cfg, torch.device("cpu"), str(tmp_path))[0][0].shape == (1, 28, 28)


def test_discrete_cifar10_reads_binary_batches(tmp_path):
    import lib.datasets.mnist as dm
    base = tmp_path / "cifar-10-batches-bin"
    base.mkdir()
    rng = np.random.default_rng(1)
    rec = rng.integers(0, 256, (4, 3073), dtype=np.uint8)
    rec[:, 0] %= 10
    (base / "test_batch.bin").write_bytes(rec.tobytes())
    ds = dm.DiscreteCIFAR10(_cfg(train=False), torch.device("cpu"), str(tmp_path))
    img, t = ds[2]
    assert img.shape == (3, 32, 32) and int(t) == int(rec[2, 0])
    np.testing.assert_array_equal(img.numpy().reshape(-1), rec[2, 1:])


def test_gray_codec_round_trip_and_adjacency(tmp_path):
    import lib.datasets.synthetic as sy
    D, scale = 32, 1000.0
    rng = np.random.default_rng(2)
    pts = rng.uniform(-4, 4, (500, 2))
    for mode in ("gray", "normal"):
        bits = sy.float2bin(pts, D, scale, mode)
        assert bits.shape == (500, D) and set(np.unique(bits)) <= {0, 1}
        back = sy.bin2float(bits, D, scale, mode)
        np.testing.assert_allclose(back, np.trunc(pts * scale) / scale, atol=1e-12)
    # Gray property: magnitudes m and m+1 differ in exactly one bit
    m = np.arange(0, 2000)
    a = sy.float2bin(np.stack([m, m], 1) / scale + 1e-9, D, scale, "gray")
    b = sy.float2bin(np.stack([m + 1, m + 1], 1) / scale + 1e-9, D, scale, "gray")
    assert ((a != b).sum(1) == 2).all()                      # one bit per coordinate
    np.save(tmp_path / "toy.npy", sy.float2bin(pts, D, scale))
    ds = sy.SyntheticData(None, torch.device("cpu"), str(tmp_path / "toy.npy"))
    assert len(ds) == 500 and ds[5].shape == (D,)


def test_exp_hamming_mmd_matches_dense_formula():
    from lib.datasets.metrics import binary_exp_hamming_mmd
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 2, (70, 32), generator=g)
    y = (torch.rand((90, 32), generator=g) < 0.3).long()

    def dense(a, b):
        return torch.exp(-0.1 * (a.float().unsqueeze(1) - b.float().unsqueeze(0)).abs().sum(-1))
    kxx = (dense(x, x) * (1 - torch.eye(70))).sum() / 70 / 69
    kyy = (dense(y, y) * (1 - torch.eye(90))).sum() / 90 / 89
    want = kxx + kyy - 2 * dense(x, y).sum() / 70 / 90
    got = binary_exp_hamming_mmd(x, y)
    np.testing.assert_allclose(got.item(), want.item(), rtol=1e-5)
    assert binary_exp_hamming_mmd(x, x.clone()).abs().item() < 0.05 and got.item() > 0.01


def test_maze_generation_and_accuracy():
    """Maze3S data (reference lib/datasets/maze.py:758-966): perfect 7x7-cell mazes on a 15x15 grid, solved by BFS;
    maze_acc accepts exactly the samples that equal the solved form of their own walls."""
    import random
    import lib.datasets.maze as mz
    rng = random.Random(3)
    for random_entry in (False, True):
        g = mz.grow_tree_maze(rng, random_entry=random_entry)
        assert g.shape == (15, 15)
        # a spanning tree over 49 cells: 49 cells + 48 passages + 2 openings are floor, the rest wall
        assert (g == mz.FLOOR).sum() == 49 + 48 + 2 and (g == mz.WALL).sum() == 225 - 99
        assert (g[1::2, 1::2] == mz.FLOOR).all() and (g[0::2, 0::2] == mz.WALL).all()
        assert len(mz.find_entries(g)) == 2
        if not random_entry:
            assert g[0, 1] == mz.FLOOR and g[14, 13] == mz.FLOOR
        s = mz.find_path(g.copy(), random_entry)
        assert s is not None and (s == mz.PATH).sum() >= 15 and s[(s != mz.PATH)].tolist() == g[(s != mz.PATH)].tolist()
    data = mz.maze_gen(16, device="cpu", seed=11)
    again = mz.maze_gen(16, device="cpu", seed=11)
    assert data.shape == (16, 1, 15, 15) and data.dtype == torch.int64 and torch.equal(data, again)
    assert set(np.unique(data.numpy()).tolist()) == {0, 1, 2}
    valid = mz.maze_acc(data.numpy().reshape(16, -1), verbose=False)
    assert valid.shape == (16, 15, 15) and mz.maze_acc.last["accuracy"] == 1.0
    # corrupt half of them: a path cell turned into free floor is no longer the solved form
    bad = data.numpy().reshape(16, 15, 15).copy()
    for i in range(8):
        ys, xs = np.nonzero(bad[i] == mz.PATH)
        bad[i, ys[3], xs[3]] = mz.FLOOR
    valid = mz.maze_acc(bad, verbose=False)
    assert valid.shape[0] == 8 and abs(mz.maze_acc.last["accuracy"] - 0.5) < 1e-12
    assert mz.maze_acc(np.zeros((2, 225), dtype=np.int64), verbose=False).shape == (0, 15, 15)
    # BFS tie-break: a 2-wide room has two shortest paths; the first direction of the reference's order wins
    room = np.zeros((15, 15), dtype=np.int64)
    room[0, 1] = room[14, 13] = 2
    room[1:14, 1:14] = 2
    s = mz.find_path(room.copy(), True)
    assert (s == 1).sum() == 27 and s[1, 13] == 1 and s[13, 1] == 2       # goes right along the top first, then down


def test_maze_dataset_registry():
    import lib.datasets.maze  # noqa: F401
    import lib.datasets.dataset_utils as du
    from config.maze_config.config_hollow_maze import get_config
    cfg = get_config()
    ds = du.get_dataset(cfg, "cpu")
    assert len(ds) == cfg.data.batch_size
    item = ds[0]
    assert item.shape == (1, 15, 15) and int(item.max()) <= 2
    cfg.data.name, cfg.data.limit = "Maze3SComplete", 5
    full = du.get_dataset(cfg, "cpu")
    assert len(full) == 5 and full[4].shape == (1, 15, 15)


def test_frechet_distance_plumbing():
    """FID plumbing (reference lib/datasets/mnist_fid.py:74-192) with a stand-in feature network: closed forms of the
    Frechet distance, and the image path (scaling by S-1, grey -> 3 channels, statistics in fp64)."""
    import pytest
    import lib.datasets.mnist_fid as fid
    rng = np.random.default_rng(0)
    d = 6
    a, b = rng.uniform(0.5, 2.0, d), rng.uniform(0.5, 2.0, d)
    m1, m2 = rng.normal(size=d), rng.normal(size=d)
    # commuting (diagonal) covariances: |dm|^2 + sum (sqrt a - sqrt b)^2
    want = ((m1 - m2) ** 2).sum() + ((np.sqrt(a) - np.sqrt(b)) ** 2).sum()
    assert abs(fid.calculate_frechet_distance(m1, np.diag(a), m2, np.diag(b)) - want) < 1e-9
    q = np.linalg.qr(rng.normal(size=(d, d)))[0]
    s1 = q @ np.diag(a) @ q.T
    assert abs(fid.calculate_frechet_distance(m1, s1, m1, s1)) < 1e-8
    with pytest.raises(ValueError):
        fid.calculate_frechet_distance(m1, s1, m2[:3], s1[:3, :3])

    class Feat(torch.nn.Module):                       # (B, 3, H, W) -> (B, 4, 2, 2) maps, returned in a list as InceptionV3 does
        def forward(self, x):
            return [torch.nn.functional.adaptive_avg_pool2d(x[:, :1] * torch.tensor([1.0, 2.0, -1.0, 0.5]).view(1, 4, 1, 1), 2)]

    x1 = rng.integers(0, 256, (40, 1, 8, 8))
    x2 = rng.integers(0, 128, (33, 1, 8, 8))
    got = fid.evaluate_fid_score(x1, x2, batch_size=16, model=Feat(), dims=4, device="cpu")
    f1 = (x1 / 255.0).mean(axis=(1, 2, 3))[:, None] * np.array([1.0, 2.0, -1.0, 0.5])
    f2 = (x2 / 255.0).mean(axis=(1, 2, 3))[:, None] * np.array([1.0, 2.0, -1.0, 0.5])
    want = fid.calculate_frechet_distance(f1.mean(0), np.cov(f1, rowvar=False), f2.mean(0), np.cov(f2, rowvar=False))
    assert abs(got - want) < 1e-5 * max(1.0, abs(want))     # (features pass through fp32 in the network)
    assert abs(fid.evaluate_fid_score(x1, x1, model=Feat(), dims=4, device="cpu")) < 1e-5
    with pytest.raises(RuntimeError):
        fid.evaluate_fid_score(x1, x2)
