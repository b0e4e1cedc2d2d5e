"""Inputs beyond the per-launch grid limit take several kernel launches (launch_blocks, launch_batch,
launch_quant, widen, narrow, permute).  The limit is 2^22 workgroups -- 16 Gi elements -- so the offset
arithmetic of that path is driven here with GGQ_MAX_GRID=3 (read once at library load, hence a child
process) on small inputs, against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wide,schedule,grid", [
    ("0", None, "3"), ("1", None, "3"), ("0", "1,8,2", "3"), ("0", "4,8,3", "3"), ("0", "0,0,1", "3"),
    ("0", None, None), ("0", "1,8,2", None), ("0", "2,4,0", None)])
def test_split_launch_paths(oracle, wide, schedule, grid):
    """wide = 1 also forces the 64-bit index kernels that tensors of 2^32 pieces and more take.  `schedule`
    (GGQ_PREFETCH = period,lead,wait) forces the group start -- which the library only uses from 64 Mi elements --
    and short burst periods onto these small inputs, in the single-tensor and in the batch kernel; with a group
    start in the schedule, launch plans also turn into hybrids (tensors above GGQ_HYBRID_MIN_ELEMS leave the tile
    table for launches of their own; tensors that are adjacent in memory count as one run), so the child's plans mix
    all routes; grid = None leaves the grid limit alone."""
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi_launch_child.py")
    env = dict(os.environ, GGQ_WIDE_INDEX=wide, GGQ_HYBRID_MIN_ELEMS="20000")
    env.pop("GGQ_MAX_GRID", None)
    if grid:
        env["GGQ_MAX_GRID"] = grid
    if schedule:
        env["GGQ_PREFETCH"] = schedule
    r = subprocess.run([sys.executable, child], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-2000:], r.stderr[-4000:])
