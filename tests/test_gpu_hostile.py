"""Scene data that points outside its own arrays must not fault the GPU.  The reference follows such ints blindly (undefined behaviour);
here a model block whose pointer, primitive count or material pointers leave their palettes never intersects, an octree leaf beyond the
block palette is air, and an entity BVH whose leaves or triangle materials leave their palettes makes the render call fail with
CHUNKY_E_INVALID (capi.hip derive_records / bvh_leaves_sound; the host parsers themselves run under AddressSanitizer in
tests/test_sanitize.py).  Each batch renders in a child process: a memory fault would abort it."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1000, 2000, 3000])
def test_hostile_scene_data_never_faults(seed):
    proc = subprocess.run([sys.executable, os.path.join(HERE, "hostile_child.py"), str(seed), "24"], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, (proc.stdout[-2000:], proc.stderr[-3000:])
    out = json.loads(proc.stdout.strip().splitlines()[-1])
    assert out["rendered"] + out["refused"] == 24 * 5 and out["rendered"] > 0
    assert out["kernels_differ"] == [], out["kernels_differ"]  # [seed, what was damaged, kernel variant]
