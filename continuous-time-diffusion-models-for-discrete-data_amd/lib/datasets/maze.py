"""Solved 15x15 mazes as 3-state grids (reference lib/datasets/maze.py:758-966): 0 = wall, 2 = free floor, 1 = the
shortest path between the two border openings.  `maze_gen` draws perfect mazes on 7x7 cells with the growing-tree rule
(newest cell with probability weightLow %, a random open cell between the two weights, the oldest one above weightHigh),
opens an entry in the top row and an exit in the bottom row (fixed at the corners, or random columns), solves the maze by
breadth-first search and, under `random_transform`, rotates half of them by 90 degrees.  `maze_acc` is the evaluation
the reference runs on samples: a sample counts iff re-solving its own walls reproduces it exactly.

The grid is held as one int array (cell (cx, cy) at [2 cy + 1, 2 cx + 1], passages between neighbouring cells); the
reference builds tile objects and renders them through PIL.  Mazes come from this module's own `random.Random`
(seedable through `seed=`), so the stream differs from the reference's global `random`: the DISTRIBUTION is the same,
individual mazes are not -- data is random either way."""
import random
from collections import deque

import numpy as np
import torch
from torch.utils.data import Dataset

import lib.datasets.dataset_utils as dataset_utils

WALL, PATH, FLOOR = 0, 1, 2
_NEIGH = ((0, -1), (1, 0), (0, 1), (-1, 0))          # N, E, S, W in cell coordinates (dx, dy)
_BFS_DIRS = ((0, 1), (1, 0), (0, -1), (-1, 0))       # order of maze.py:782: decides ties between equally short paths


def grow_tree_maze(rng, dim_x=7, dim_y=7, weight_high=99, weight_low=97, random_entry=False):
    """One perfect maze as a (2 dim_y + 1, 2 dim_x + 1) int array of WALL / FLOOR (maze.py:419-522, 314-329, 584-722)."""
    grid = np.zeros((2 * dim_y + 1, 2 * dim_x + 1), dtype=np.int64)
    seen = np.zeros((dim_y, dim_x), dtype=bool)
    cx, cy = rng.randrange(dim_x), rng.randrange(dim_y)
    seen[cy, cx] = True
    grid[2 * cy + 1, 2 * cx + 1] = FLOOR
    live = [(cx, cy)]
    while live:
        c = rng.random() * 100
        if c <= weight_low:
            idx = len(live) - 1
        elif c < weight_high:
            idx = rng.randrange(len(live))
        else:
            idx = 0
        cx, cy = live[idx]
        free = [(cx + dx, cy + dy) for dx, dy in _NEIGH
                if 0 <= cx + dx < dim_x and 0 <= cy + dy < dim_y and not seen[cy + dy, cx + dx]]
        if not free:
            live.pop(idx)
            continue
        nx, ny = rng.choice(free)
        seen[ny, nx] = True
        grid[2 * ny + 1, 2 * nx + 1] = FLOOR
        grid[cy + ny + 1, cx + nx + 1] = FLOOR                       # the passage between the two cells
        live.append((nx, ny))
    top = rng.randrange(dim_x) if random_entry else 0
    bottom = rng.randrange(dim_x) if random_entry else dim_x - 1
    grid[0, 2 * top + 1] = FLOOR
    grid[-1, 2 * bottom + 1] = FLOOR
    return grid


def find_entries(array):
    """Border cells in state 2, top/bottom row first (column by column), then the side columns; at most two (maze.py:758-777)."""
    H, W = array.shape
    entries = []
    for i in range(W):
        if array[0, i] == FLOOR:
            entries.append((0, i))
        if array[-1, i] == FLOOR:
            entries.append((H - 1, i))
    for j in range(1, H - 1):
        if array[j, 0] == FLOOR:
            entries.append((j, 0))
        if array[j, -1] == FLOOR:
            entries.append((j, W - 1))
    return entries[:2]


def find_path(maze, random_entry=False):
    """Shortest path through state-2 cells between the two openings, written into `maze` as state 1 (in place, as the
    reference does); None when there are not exactly two openings or no path (maze.py:780-818)."""
    if random_entry:
        entries = find_entries(maze)
        if len(entries) != 2:
            return None
        start, end = entries
    else:
        start, end = (0, 1), (14, 13)
    H, W = maze.shape
    parent = {start: None}
    queue = deque([start])
    while queue:
        node = queue.popleft()
        if node == end:
            while node is not None:
                maze[node] = PATH
                node = parent[node]
            return maze
        for dy, dx in _BFS_DIRS:
            nxt = (node[0] + dy, node[1] + dx)
            if 0 <= nxt[0] < H and 0 <= nxt[1] < W and maze[nxt] == FLOOR and nxt not in parent:
                parent[nxt] = node
                queue.append(nxt)
    return None


def maze_gen(limit, size=None, crop=False, random_transform=True, dim_x=7, dim_y=7, pixelSizeOfTile=1, weightHigh=99,
             weightLow=97, device="cuda", seed=None):
    """`limit` solved mazes as an int64 tensor (limit, 1, H, W) on `device` (maze.py:821-858)."""
    if pixelSizeOfTile != 1:
        raise NotImplementedError("pixelSizeOfTile != 1 (every reference config uses 1)")
    rng = random.Random(seed) if seed is not None else _GLOBAL_RNG
    out = []
    for _ in range(int(limit)):
        grid = grow_tree_maze(rng, dim_x, dim_y, weightHigh, weightLow, random_transform)
        if crop:
            grid = grid[1:-1, 1:-1].copy()
        solved = find_path(grid, random_transform)
        if solved is None:                 # cannot happen uncropped; cropped + fixed entries never solves in the reference either
            raise RuntimeError("generated maze has no solution (crop_wall needs random_transform)")
        if random_transform and rng.choice([True, False]):
            solved = np.rot90(solved).copy()
        out.append(torch.from_numpy(solved).unsqueeze(0))
    return torch.stack(out, 0).to(device)


_GLOBAL_RNG = random.Random()


def path_length(maze):
    return np.count_nonzero(maze == PATH), np.count_nonzero(maze == WALL), np.count_nonzero(maze == FLOOR)


def maze_acc(samples, verbose=True):
    """Fraction of samples that are exactly the solved form of their own walls, and those samples (maze.py:866-898).
    Returns the stack of valid mazes (K, 15, 15) -- an empty (0, 15, 15) array when none is valid, where the reference's
    np.stack raises.  The accuracy and state statistics are printed as in the reference and kept in `maze_acc.last`."""
    samples = np.asarray(samples).reshape(-1, 15, 15)
    clean = samples.copy()
    clean[clean == PATH] = FLOOR
    ok, kept, stats = [], [], []
    for i in range(samples.shape[0]):
        solved = find_path(clean[i], True)
        good = solved is not None and bool((solved == samples[i]).all())
        ok.append(1 if good else 0)
        if good:
            kept.append(solved)
            stats.append(path_length(solved))
    acc = float(np.mean(ok)) if ok else 0.0
    mean = np.mean(np.asarray(stats, dtype=np.float64).reshape(-1, 3), axis=0) if stats else np.full(3, np.nan)
    maze_acc.last = {"accuracy": acc, "path_len": float(mean[0]), "wall_len": float(mean[1]), "way_len": float(mean[2])}
    if verbose:
        print(f"Accuracy: From {samples.shape[0]} are {acc * 100}% solvable.")
        for name, v in (("path", mean[0]), ("wall", mean[1]), ("way", mean[2])):
            print(f"Average {name} length: {v} and prob {v * 100 / 225}%")
    return np.stack(kept, 0) if kept else np.zeros((0, 15, 15), dtype=samples.dtype)


@dataset_utils.register_dataset
class Maze3SComplete(Dataset):
    """`cfg.data.limit` mazes generated once and held on the device (maze.py:922-943)."""

    def __init__(self, cfg, device, _=None):
        self.device = device
        self.data = maze_gen(limit=cfg.data.limit, crop=cfg.data.crop_wall, dim_x=7, dim_y=7,
                             random_transform=cfg.data.random_transform, device=device,
                             seed=getattr(cfg.data, "seed", None))

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.data[idx]


@dataset_utils.register_dataset
class Maze3S(Dataset):
    """An endless stream: every item is a freshly generated maze; `len` = batch size, so one epoch is one batch
    (maze.py:945-966)."""

    def __init__(self, cfg, device, _=None):
        self.cfg = cfg
        self.device = device

    def __len__(self):
        return int(self.cfg.data.batch_size)

    def __getitem__(self, idx):
        self.maze = maze_gen(limit=self.cfg.data.limit, device=self.device, crop=self.cfg.data.crop_wall,
                             random_transform=self.cfg.data.random_transform, dim_x=7, dim_y=7)
        return self.maze[0]
