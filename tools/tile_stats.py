"""tools/tile_stats.py -- CPU study of the sparse-conv tile geometry on the bench frames (no GPU).

For every resolution level of the isfusion_0075voxel SparseEncoder and for several ROW ORDERS of the active set
(raster (b,z,y,x) = what the round-2 kernels use; bricks of bz x by x bx cells in raster order of the bricks; Morton)
it reports, for the SubM 3x3x3 rulebook of the level:

  issued / algorithmic   MFMA work the f16x3 kernel issues (a 16-row group multiplies a tap when ANY of its rows has
                         a neighbour through it) over the pairs that exist;
  rows / tile            distinct input rows a 128-row tile touches over all 27 taps (what an LDS staging of the tile's
                         input rows would have to hold) and the gathers per distinct row (pairs of the tile / distinct);
  taps / tile            taps with at least one pair in the tile (workgroup-level steps).

    python tools/tile_stats.py [--frames 4] [--points 300000] [--tile 128]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VS = np.array([0.075, 0.075, 0.2], np.float32)
RANGE = np.array([-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], np.float32)
SHAPE0 = (41, 1440, 1440)


def voxelize(pts):
    c = np.floor((pts[:, :3] - RANGE[:3]) / VS).astype(np.int64)   # x, y, z
    ok = (c >= 0).all(1) & (c[:, 0] < 1440) & (c[:, 1] < 1440) & (c[:, 2] < 40)
    c = c[ok]
    return np.unique(np.stack([c[:, 2], c[:, 1], c[:, 0]], 1), axis=0)   # z, y, x


def down(coords, shape, pad, ks=(3, 3, 3), st=(2, 2, 2)):
    """output set of a strided SparseConv3d: every output site reached by an input through some tap."""
    oshape = tuple((shape[j] + 2 * pad[j] - ks[j]) // st[j] + 1 for j in range(3))
    outs = []
    for kz in range(ks[0]):
        for ky in range(ks[1]):
            for kx in range(ks[2]):
                k = np.array([kz, ky, kx])
                num = coords + np.array(pad) - k
                ok = (num % np.array(st) == 0).all(1)
                o = num[ok] // np.array(st)
                ok2 = (o >= 0).all(1) & (o < np.array(oshape)).all(1)
                outs.append(o[ok2])
    return np.unique(np.concatenate(outs), axis=0), oshape


def lin(c, shape):
    return (c[:, 0] * shape[1] + c[:, 1]) * shape[2] + c[:, 2]


def order_key(c, order):
    z, y, x = c[:, 0], c[:, 1], c[:, 2]
    if order == "raster":
        return (z << 40) | (y << 20) | x
    if order.startswith("brick"):
        bz, by, bx = (int(t) for t in order[5:].split("x"))
        brick = ((z // bz) << 40) | ((y // by) << 20) | (x // bx)
        inner = ((z % bz) * by + (y % by)) * bx + (x % bx)
        return brick * (bz * by * bx) + inner
    if order.startswith("ybrick"):   # bricks ordered (by, bx, bz): all z of a (y, x) patch adjacent
        bz, by, bx = (int(t) for t in order[6:].split("x"))
        brick = ((y // by) << 40) | ((x // bx) << 20) | (z // bz)
        inner = ((z % bz) * by + (y % by)) * bx + (x % bx)
        return brick * (bz * by * bx) + inner
    if order == "morton":
        def part(v):
            v = v.astype(np.uint64)
            r = np.zeros_like(v)
            for b in range(12):
                r |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
            return r
        return ((part(z) << np.uint64(2)) | (part(y) << np.uint64(1)) | part(x)).astype(np.int64)
    raise ValueError(order)


def subm_nbr(coords, shape):
    """nbr[k][o] = index (into coords, which must be lin-sorted) of the tap-k input of output o, -1 if none."""
    keys = lin(coords, shape)
    n = coords.shape[0]
    nbr = np.full((27, n), -1, np.int64)
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                q = coords + np.array([dz, dy, dx])
                ok = (q >= 0).all(1) & (q < np.array(shape)).all(1)
                qk = lin(q, shape)
                pos = np.searchsorted(keys, qk)
                pos[pos >= n] = n - 1
                hit = ok & (keys[pos] == qk)
                nbr[k, hit] = pos[hit]
                k += 1
    return nbr


def stats(nbr_raster, perm, tile, group=16):
    """nbr in raster ids; perm[new_row] = raster id.  -> dict of the figures in the module docstring."""
    n = perm.shape[0]
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    nb = nbr_raster[:, perm]                # taps of the rows in the new order (values still raster ids)
    nbn = np.where(nb >= 0, inv[np.maximum(nb, 0)], -1)
    has = nbn >= 0
    pairs = int(has.sum())
    pad = (-n) % group
    hp = np.pad(has, ((0, 0), (0, pad)))
    g_any = hp.reshape(27, -1, group).any(2)
    issued = int(g_any.sum()) * group
    g32 = np.pad(has, ((0, 0), (0, (-n) % 32))).reshape(27, -1, 32).any(2)
    issued32 = int(g32.sum()) * 32
    # per tile: distinct input rows and taps
    ntiles = (n + tile - 1) // tile
    distinct = np.zeros(ntiles, np.int64)
    tp = np.zeros(ntiles, np.int64)
    taps = np.zeros(ntiles, np.int64)
    for t in range(ntiles):
        blk = nbn[:, t * tile:(t + 1) * tile]
        v = blk[blk >= 0]
        distinct[t] = np.unique(v).size
        tp[t] = v.size
        taps[t] = (blk >= 0).any(1).sum()
    return dict(n=n, pairs=pairs, ppv=pairs / n, issue16=issued / pairs, issue32=issued32 / pairs,
                distinct_mean=distinct.mean(), distinct_p99=np.percentile(distinct, 99), distinct_max=distinct.max(),
                reuse=tp.sum() / distinct.sum(), taps=taps.mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--points", type=int, default=300000)
    ap.add_argument("--tile", type=int, default=128)
    ap.add_argument("--orders", default="raster,brick1x4x4,brick2x4x4,brick4x4x4,brick2x2x8,brick2x8x8,ybrick8x4x4,morton")
    args = ap.parse_args()
    from isfusion_amd import synthetic
    orders = args.orders.split(",")
    for f in range(args.frames):
        seed = 1234 + 1000 * 2 + f
        pts = synthetic.lidar_sweeps(seed, args.points)
        c = voxelize(pts)
        shape = SHAPE0
        pads = [(1, 1, 1), (1, 1, 1), (0, 1, 1)]
        for lvl in range(4):
            nbr = subm_nbr(c, shape)
            print(f"frame {seed} level {lvl} shape {shape} N {c.shape[0]}")
            for o in orders:
                perm = np.argsort(order_key(c, o), kind="stable")
                s = stats(nbr, perm, args.tile)
                print(f"  {o:12s} pairs/voxel {s['ppv']:.2f} issue16 {s['issue16']:.3f} issue32 {s['issue32']:.3f} "
                      f"distinct/tile mean {s['distinct_mean']:.0f} p99 {s['distinct_p99']:.0f} max {s['distinct_max']} "
                      f"gathers/distinct {s['reuse']:.2f} taps/tile {s['taps']:.1f}")
            if lvl < 3:
                c, shape = down(c, shape, pads[lvl])


if __name__ == "__main__":
    main()
