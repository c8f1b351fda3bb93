"""Host-side counts behind profiles/r04_row_requests.txt and DESIGN.md section 5.5 (numpy only, no GPU):

for the voxel sets of a synthetic sweep at the encoder's levels (rank order = sorted by (z, y, x), 16-row groups as the
conv kernels form them), per output row of a SubM 3 x 3 x 3 layer:

  * row requests today (every row of a group that has the tap), with absent rows skipped, and with rows whose +x neighbour
    is the next row taken from the right-hand lane (tap k of row r == tap k - 1 of row r + 1);
  * MFMA row slots (16 x active (group, tap) pairs) today, and with the rows of a 128- / 256-row tile sorted by their
    27-bit tap mask -- how much padding a regrouping of rows could remove.

    python tools/probes/row_stats.py [--points 300000] [--seed 9000]

The level sets are approximations of the encoder's (each level = the cells touched by the previous level's 2 x 2 x 2
neighbourhoods), good to a few per cent of the bench's voxel counts."""
import argparse
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_synthetic():
    spec = importlib.util.spec_from_file_location("synthetic", os.path.join(HERE, "..", "..", "is-fusion_amd", "synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def keys_of(cells, grid):
    return np.unique((cells[:, 2] * grid[1] + cells[:, 1]) * grid[0] + cells[:, 0])


def xyz(key, grid):
    return key % grid[0], (key // grid[0]) % grid[1], key // (grid[0] * grid[1])


def neighbour(key, grid, dz, dy, dx):
    """rank of the (dz, dy, dx) neighbour of every voxel, -1 if absent"""
    x, y, z = xyz(key, grid)
    zz, yy, xx = z + dz, y + dy, x + dx
    good = (zz >= 0) & (zz < grid[2]) & (yy >= 0) & (yy < grid[1]) & (xx >= 0) & (xx < grid[0])
    k2 = (zz * grid[1] + yy) * grid[0] + xx
    pos = np.searchsorted(key, k2)
    pos[pos >= len(key)] = len(key) - 1
    return np.where(good & (key[pos] == k2), pos, -1)


def down(key, grid):
    """the cells of the next level (kernel 3, stride 2, padding 1)"""
    x, y, z = xyz(key, grid)
    g2 = (grid + 1) // 2
    cells = [np.stack([(x + dx) // 2, (y + dy) // 2, (z + dz) // 2], 1) for dz in (0, 1) for dy in (0, 1) for dx in (0, 1)]
    cc = np.concatenate(cells)
    return keys_of(cc[((cc >= 0) & (cc < g2)).all(1)], g2), g2


def group_any(live, n):
    pad = (-n) % 16
    return np.concatenate([live, np.zeros(pad, bool)]).reshape(-1, 16).any(1)


def stats(key, grid, name):
    n = len(key)
    r = np.arange(n)
    xadj = neighbour(key, grid, 0, 0, 1) == r + 1
    masks = np.zeros(n, np.int64)
    now = present = shared = 0
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            prev = None
            for dx in (-1, 0, 1):
                live = neighbour(key, grid, dz, dy, dx) >= 0
                masks |= live.astype(np.int64) << k
                now += group_any(live, n).sum() * 16
                present += live.sum()
                if prev is not None:
                    shared += (live & xadj & (r % 16 < 15) & np.repeat(group_any(prev, n), 16)[:n]).sum()
                prev = live
                k += 1

    def slots(m):
        pad = (-len(m)) % 16
        g = np.bitwise_or.reduce(np.concatenate([m, np.zeros(pad, np.int64)]).reshape(-1, 16), 1)
        return sum(((g >> t) & 1).sum() for t in range(27)) * 16

    line = f"{name}: {n} rows, {present / n:.2f} pairs/row; row requests per row: today {now / n:.2f}, present only " \
           f"{present / n:.2f}, + x-sharing {(present - shared) / n:.2f}; MFMA row slots per row: today {slots(masks) / n:.2f}"
    for tm in (128, 256):
        srt = masks.copy()
        for t0 in range(0, n, tm):
            srt[t0:t0 + tm] = np.sort(srt[t0:t0 + tm])
        line += f", rows of a {tm}-row tile sorted by mask {slots(srt) / n:.2f}"
    print(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=300000)
    ap.add_argument("--seed", type=int, default=9000)
    a = ap.parse_args()
    syn = load_synthetic()
    pts = syn.lidar_sweeps(a.seed, a.points)
    vs = np.array([0.075, 0.075, 0.2])
    lo, hi = np.array(syn.PC_RANGE[:3]), np.array(syn.PC_RANGE[3:])
    grid = np.round((hi - lo) / vs).astype(np.int64)
    c = np.floor((pts[:, :3] - lo) / vs).astype(np.int64)
    key = keys_of(c[((c >= 0) & (c < grid)).all(1)], grid)
    for level in range(4):
        stats(key, grid, f"level {level}")
        key, grid = down(key, grid)


if __name__ == "__main__":
    main()
