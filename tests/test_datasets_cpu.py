"""CPU: dataset adapters and the MMD metric.  The maze solver / `maze_acc`, the Gray codec and the MMD
are pinned by outputs of the reference itself (tests/golden/aux_*.npz, written by oracle/gen_golden_aux.py; tests at the
end of this file).  The image loaders (`DiscreteMNIST` / `DiscreteCIFAR10`) stay parity unpinned: the reference's versions
subclass torchvision and download; here the format readers are checked on files written by the test."""
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


# ---------------------------------------------------------------- pinned by outputs of the reference itself
# tests/golden/aux_*.npz are written by oracle/gen_golden_aux.py from the imported reference (lib/datasets/maze.py,
# synthetic.py, metrics.py); see that file's header for what was stubbed (command-line / logging imports only).
def test_maze_solver_and_accuracy_match_the_reference(golden):
    """`find_path` (maze.py:780-818: entry search, BFS tie-break order) and `maze_acc` (866-898) on mazes the reference's
    own `maze_gen` drew: the mirror re-solves their bare walls to exactly the reference's solved grids, and keeps exactly the
    samples the reference's `maze_acc` keeps (clean, corrupted and unsolvable ones)."""
    import lib.datasets.maze as mz
    g = golden("aux_maze")
    for tag, rt in (("fixed", False), ("random", True)):
        walls, want = g[f"{tag}__walls"], g[f"{tag}__resolved"]
        for i in range(walls.shape[0]):
            got = mz.find_path(walls[i].copy(), rt)
            assert got is not None
            np.testing.assert_array_equal(got, want[i])
        np.testing.assert_array_equal(want, g[f"{tag}__solved"])            # (the reference's mazes are their own solved form)
    kept = mz.maze_acc(g["acc__samples"].copy(), verbose=False)
    np.testing.assert_array_equal(kept, g["acc__kept"])
    assert abs(mz.maze_acc.last["accuracy"] - g["acc__kept"].shape[0] / g["acc__samples"].shape[0]) < 1e-12


def test_maze_generator_matches_the_reference_in_distribution(golden):
    """The mirror draws from its own `random.Random` stream, so single mazes differ from the reference's; the LAW is pinned:
    every maze has the same 126 wall cells (a perfect maze on 7x7 cells + two openings), and the mean solution length, its
    spread and the share of rotated mazes agree with 400 reference mazes within sampling error."""
    import lib.datasets.maze as mz
    g = golden("aux_maze")
    ref = g["stats__counts"].astype(np.float64)                             # (400, 3): wall / path / floor counts
    mine = mz.maze_gen(400, random_transform=True, device="cpu", seed=99).numpy().reshape(-1, 15, 15)
    cnt = np.stack([(mine == s).sum(axis=(1, 2)) for s in (0, 1, 2)], 1).astype(np.float64)
    assert (cnt[:, 0] == 126).all() and (ref[:, 0] == 126).all()
    se = np.sqrt(ref[:, 1].var() / 400 + cnt[:, 1].var() / 400)
    assert abs(cnt[:, 1].mean() - ref[:, 1].mean()) < 4 * se, (cnt[:, 1].mean(), ref[:, 1].mean())
    assert 0.7 < cnt[:, 1].std() / ref[:, 1].std() < 1.4
    rot = float((mine[:, 0, :] == 0).all(axis=1).mean())
    assert abs(rot - float(g["stats__rot_share"][0])) < 4 * np.sqrt(0.25 / 400 * 2)
    kept = mz.maze_acc(mine, verbose=False)                                 # every generated maze is its own solved form
    assert kept.shape[0] == 400


def test_gray_codec_matches_the_reference(golden):
    """`float2bin` / `bin2float` (synthetic.py:164-224) in both bin maps: same bits for the same points, same points back."""
    import lib.datasets.synthetic as sy
    g = golden("aux_synthetic")
    for D in (32, 16):
        for mode in ("gray", "normal"):
            pts, bits, back = g[f"D{D}__{mode}__points"], g[f"D{D}__{mode}__bits"], g[f"D{D}__{mode}__back"]
            scale = float(g[f"D{D}__{mode}__scale"][0])
            got = sy.float2bin(pts, D, scale, binmode=mode)
            np.testing.assert_array_equal(np.asarray(got), bits)
            np.testing.assert_allclose(np.asarray(sy.bin2float(bits, D, scale, binmode=mode)), back, rtol=0, atol=1e-12)


def test_mmd_matches_the_reference(golden):
    """`binary_exp_hamming_mmd` (metrics.py:24-56) on seeded bit arrays: the blocked on-device evaluation against the
    reference's dense (N, M, D) one, two bandwidths, square and ragged shapes."""
    import lib.datasets.metrics as mt
    g = golden("aux_metrics")
    for tag in ("a", "b", "c"):
        x, y = torch.from_numpy(g[f"{tag}__x"]), torch.from_numpy(g[f"{tag}__y"])
        for bw in (0.1, 0.5):
            want = float(g[f"{tag}__mmd_bw{bw}"][0])
            got = float(mt.binary_exp_hamming_mmd(x, y, None, bandwidth=bw))
            assert abs(got - want) < 1e-6 + 1e-5 * abs(want), (tag, bw, got, want)
        sim = g[f"{tag}__sim_bw0.1"]
        tot = float(mt.exp_hamming_gram_sum(x.float(), y.float(), 0.1))
        assert abs(tot - float(sim.astype(np.float64).sum())) < 1e-4 * float(sim.sum())
