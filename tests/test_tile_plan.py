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
