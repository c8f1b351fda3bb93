"""Host logic of the sparse-conv tile plan (csrc/isf_spconv16.h: conv16_plan / conv16_tile_rows): a small host program
compiled with hipcc (no GPU needed) enumerates the tiles of a launch exactly as the kernel's workgroups do; every output
row must be covered exactly once, whatever the full / half tile mix, and in one-round launches no CU may be dealt more
16-row groups than ceil(need)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

SRC = r"""
#include <cstdio>
#include <cstdlib>
#include "isf_spconv16.h"
int main(int argc, char** argv) {
  const int n_out = atoi(argv[1]), TM = atoi(argv[2]), ncb = atoi(argv[3]), k = atoi(argv[4]), cus = atoi(argv[5]);
  const bool balance = atoi(argv[6]) != 0;
  const isf::Conv16Plan plan = isf::conv16_plan(n_out, TM, ncb, k, cus, balance);
  printf("plan %d %d %d\n", plan.full, plan.half, plan.part_rows);
  const int parts = ncb == 2 ? 4 : 8;
  for (int part = 0; part < parts; ++part)
    for (int j = 0; j < plan.full + plan.half; ++j) {
      int row0, row_end; bool half;
      if (isf::conv16_tile_rows(plan, TM, n_out, part, j, row0, row_end, half)) {
        const int last = row0 + (half ? TM / 2 : TM);
        printf("tile %d %d %d %d %d\n", part, j, row0, last < row_end ? last : row_end, (int)half);
      }
    }
  return 0;
}
"""


@pytest.fixture(scope="module")
def plan_exe(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    d = tmp_path_factory.mktemp("tile_plan")
    src = d / "plan.hip"
    src.write_text(SRC)
    exe = d / "plan"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "is-fusion_amd", "csrc"),
                    "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True, capture_output=True)
    return str(exe)


def tiles_of(exe, n_out, TM, ncb, k=3, cus=32, balance=1):
    out = subprocess.run([exe, str(n_out), str(TM), str(ncb), str(k), str(cus), str(balance)], check=True,
                         capture_output=True, text=True).stdout.split("\n")
    plan = tuple(int(v) for v in out[0].split()[1:])
    tiles = [tuple(int(v) for v in ln.split()[1:]) for ln in out[1:] if ln.startswith("tile")]
    return plan, tiles


@pytest.mark.parametrize("n_out", [1, 10, 127, 128, 129, 1000, 20345, 36130, 40691, 44704, 119321, 346474])
@pytest.mark.parametrize("TM,ncb,k", [(128, 2, 3), (128, 1, 3), (256, 1, 2), (128, 1, 7)])
@pytest.mark.parametrize("balance", [0, 1])
def test_every_row_is_covered_exactly_once(plan_exe, n_out, TM, ncb, k, balance):
    plan, tiles = tiles_of(plan_exe, n_out, TM, ncb, k, balance=balance)
    covered = sorted((r0, r1) for _, _, r0, r1, _ in tiles if r1 > r0)
    assert covered[0][0] == 0 and covered[-1][1] == n_out
    for (a0, a1), (b0, b1) in zip(covered, covered[1:]):
        assert a1 == b0, (plan, a0, a1, b0, b1)        # no gap, no overlap
    if not balance:
        assert plan[1] == 0                            # uniform tiles only


def test_one_round_launches_are_dealt_evenly(plan_exe):
    """BASELINE configs[1], frame set 0, level 3 (40691 rows, 256 columns): 64 full + 31 half tiles per XCD = 2 full +
    1 half per CU -> 5 row groups per SIMD where uniform tiles give 6 on the CUs that receive three workgroups"""
    plan, tiles = tiles_of(plan_exe, 40691, 128, 2)
    assert plan[:2] == (64, 31)
    # workgroups are dealt to the 32 CUs of an XCD in order: CU c gets tiles j = c, c + 32, c + 64
    groups_per_cu = [0] * 32
    for part, j, r0, r1, half in tiles:
        if part == 0:
            groups_per_cu[j % 32] += (4 if half else 8)
    assert max(groups_per_cu) == 20 and min(groups_per_cu) >= 16          # 20 groups per CU = 5 per SIMD
    uniform, _ = tiles_of(plan_exe, 40691, 128, 2, balance=0)
    assert uniform[0] == 80 and uniform[1] == 0                            # 80 tiles on 32 CUs: 3 on some (6 per SIMD)
    # a frame set whose need rounds up to 6 keeps uniform tiles
    plan6, _ = tiles_of(plan_exe, 44704, 128, 2)
    assert plan6[1] == 0


@pytest.mark.parametrize("TM,ncb,k,bound", [(128, 2, 3, 96 * 256), (256, 1, 2, 256 * 256)])
def test_small_launch_bounds_are_where_the_plan_stops_being_all_half(plan_exe, TM, ncb, k, bound):
    """round 5 (isf_spconv16.hip: launch16_rows / conv16_device_cus): a launch of at most
    96 x CUs rows (256-column layers: 128-row tiles, two column blocks, 3 workgroups per CU) or 256 x CUs rows (the 8-wave
    128-column shape: 256-row tiles, 2 per CU) is cut into HALF tiles only -- the launches the one-group instantiation
    takes over -- and the first row past the bound brings full tiles in.  The one-group launch itself (tiles of TM / 2 rows,
    uniform: half tiles make no sense with one 16-row group per wave) covers every row exactly once."""
    for n_out in (2048, bound // 3, bound - 17, bound):
        plan, _ = tiles_of(plan_exe, n_out, TM, ncb, k)
        assert plan[0] == 0 and plan[1] > 0, (n_out, plan)
    for n_out in (bound + 1, bound + 5000):
        plan, _ = tiles_of(plan_exe, n_out, TM, ncb, k)
        assert plan[0] > 0, (n_out, plan)
    for n_out in (2048, 9000, bound - 17, bound):
        plan, tiles = tiles_of(plan_exe, n_out, TM // 2, ncb, k + 1, balance=0)
        assert plan[1] == 0
        covered = sorted((r0, r1) for _, _, r0, r1, _ in tiles if r1 > r0)
        assert covered[0][0] == 0 and covered[-1][1] == n_out
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))


# ----------------------------------------------------------------------------------------------------------------------
# Unit plan of the one-workgroup-per-CU kernel (csrc/isf_spconv16.h: conv_cu_cut / conv_cu_piece, walked on the host by
# isf_sparse_conv_cu_plan_host -- the same functions the device planner calls; no GPU work).
def _cu_units(work, cus=256):
    from isfusion_amd import spconv as sp
    return sp.cu_plan_host(work, cus)


@pytest.mark.parametrize("case", ["level3", "dense", "sparse", "one", "tiny", "zeros", "big", "spiky", "cus32"])
def test_cu_unit_plan_covers_every_group_once_and_balances(case):
    import numpy as np
    from isfusion_amd import _lib
    rng = np.random.default_rng(sum(map(ord, case)))
    cus = 256
    if case == "level3":      # 40.7 k rows, 9..27 taps per group (profiles/r03_conv_trace.txt: mean 16)
        work = np.clip(rng.normal(16, 5, 2544), 9, 27).astype(np.int32)
    elif case == "dense":
        work = np.full(2544, 27, np.int32)
    elif case == "sparse":    # 3-tap conv_out: the cap, not the work, sizes the units
        work = rng.integers(1, 4, 2258).astype(np.int32)
    elif case == "one":
        work = np.array([5], np.int32)
    elif case == "tiny":
        work = rng.integers(1, 28, 7).astype(np.int32)
    elif case == "zeros":     # rows without any neighbour still belong to exactly one unit
        work = np.zeros(300, np.int32); work[::17] = 9
    elif case == "big":       # several units per CU
        work = rng.integers(1, 28, 20000).astype(np.int32)
    elif case == "spiky":
        work = np.ones(5000, np.int32); work[2000:2100] = 27
    else:
        work = rng.integers(1, 28, 1000).astype(np.int32); cus = 32
    units = _cu_units(work, cus)
    n = len(work)
    assert len(units) <= _lib.load().isf_sparse_conv_cu_max_units(n, cus)
    # contiguous, ascending, every group exactly once, 1..16 groups per unit
    assert units[0, 0] == 0 and units[-1, 0] + units[-1, 1] == n
    assert (units[1:, 0] == units[:-1, 0] + units[:-1, 1]).all()
    assert units[:, 1].min() >= 1 and units[:, 1].max() <= 16
    # balance: no unit carries more than the balanced share + one group's worth of rounding (the cap may only cut work)
    w = np.array([work[a:a + k].sum() for a, k in units])
    r = max(1, -(-n // (cus * 12)))
    share = work.sum() / (cus * r)
    assert w.max() <= share + 27 + 1e-9, (w.max(), share)
    if case in ("level3", "dense"):
        assert len(units) == cus and w.min() >= share - 27          # one unit per CU, all within a group of the mean


def test_cu_unit_plan_is_deterministic_and_monotone_in_cuts():
    import numpy as np
    rng = np.random.default_rng(5)
    work = rng.integers(0, 28, 3333).astype(np.int32)
    a, b = _cu_units(work), _cu_units(work.copy())
    assert np.array_equal(a, b)


# ----------------------------------------------------------------------------------------------------------------------
# Equal-work TILE TABLE of the tile kernel's one-round launches (csrc/isf_spconv16.h: conv16_table_part, walked on the host
# by isf_sparse_conv_tile_table_host -- the function the device planner calls)
@pytest.mark.parametrize("case", ["level3", "dense", "sparse_cap", "tiny", "nofit", "level2_8wave"])
def test_tile_table_covers_every_group_once_and_balances_the_compute_units(case):
    import numpy as np
    from isfusion_amd import spconv as sp
    rng = np.random.default_rng(sum(map(ord, case)))
    cus, wgs, gt, parts = 32, 3, 8, 4
    if case == "level3":          # 40.7 k rows -> 2544 groups, 636 per part, work 9..27
        part_groups, work = 636, np.clip(rng.normal(16, 5, 2544), 9, 27).astype(np.int32)
        dense = np.zeros(2544, bool); dense[300:700] = True; work[dense] = 27          # a dense region
    elif case == "dense":
        part_groups, work = 636, np.full(2544, 27, np.int32)
    elif case == "sparse_cap":    # 760 groups per part on 32 x 24 = 768 slots of capacity: the cap binds
        part_groups, work = 760, rng.integers(1, 28, 3040).astype(np.int32)
    elif case == "tiny":
        part_groups, work = 2, rng.integers(1, 28, 7).astype(np.int32)
    elif case == "nofit":
        part_groups, work = 800, rng.integers(1, 28, 3200).astype(np.int32)
    else:                         # 8-wave 128-column shape: 2 workgroups per CU, 16 groups per tile, 8 parts
        cus, wgs, gt, parts = 32, 2, 16, 8
        part_groups, work = 933, rng.integers(5, 28, 7458).astype(np.int32)
    tiles, fits = sp.tile_table_host(work, part_groups, parts, cus, wgs, gt)
    if case == "nofit":
        assert not fits
        return
    assert fits
    n = len(work)
    seen = np.zeros(n, np.int32)
    for p in range(parts):
        G0, G1 = min(p * part_groups, n), min((p + 1) * part_groups, n)
        cu_work = np.zeros(cus)
        cu_groups = np.zeros(cus, np.int32)
        nxt = G0
        for c in range(cus):
            for k in range(wgs):
                g0, ng = tiles[p, c + cus * k]
                if ng == 0:
                    continue
                assert 1 <= ng <= gt and g0 == nxt and g0 + ng <= G1, (p, c, k, g0, ng)     # contiguous, in CU order
                assert k == 0 or tiles[p, c + cus * (k - 1), 1] == gt                       # full tiles first, remainder last
                seen[g0:g0 + ng] += 1
                cu_work[c] += work[g0:g0 + ng].sum() + 2 * ng
                cu_groups[c] += ng
                nxt = g0 + ng
        assert nxt == G1 and cu_groups.max() <= wgs * gt
        if G1 - G0 >= 4 * cus and case != "sparse_cap":
            tot = cu_work.sum()
            assert cu_work.max() <= tot / cus + 2 * 29 + 1e-9, (case, p, cu_work.max(), tot / cus)   # within ~2 groups of the mean
    assert (seen == 1).all()
