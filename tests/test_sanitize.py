"""Host-side code that runs on caller-supplied ints, under AddressSanitizer + UBSan on the CPU (GPU sanitizers do not exist on this
pool): the octree re-layout of every scene upload (csrc/widetree.cpp) on random well-formed trees — every cell's leaf value and level
equal to the reference descent (K/octree.h:81-89) — and on damaged ones (wild branches, cycles, branches below level 0, pointers that do
not fit), which have to be refused or expressed without a single out-of-bounds access."""
import json
import os
import subprocess

from chunkyclplugin_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_widetree_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "widetree_fuzz")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-Wall", "-Wextra", "-Werror",
           os.path.join(ROOT, "tests", "sanitize", "widetree_fuzz.cpp"), os.path.join(native.CSRC, "widetree.cpp"), "-o", exe]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    proc = subprocess.run([exe, "1500"], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    out = json.loads(proc.stdout.strip().splitlines()[-1])
    assert out["rounds"] == 1500 and out["expressed"] > 1000 and out["refused"] > 100 and out["cells_compared"] > 10 ** 7


def test_capi_host_parsers_under_asan_ubsan(tmp_path):
    """derive_records / build_quad_aux / build_bvh_records / bvh_leaves_sound / list_emitters of capi.hip, compiled for the host only"""
    exe = str(tmp_path / "capi_host_fuzz")
    cmd = ["hipcc", "-x", "hip", "--offload-host-only", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-ffp-contract=off", os.path.join(ROOT, "tests", "sanitize", "capi_host_fuzz.cpp"), os.path.join(native.CSRC, "widetree.cpp"), "-o", exe]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-3000:]
    proc = subprocess.run([exe, "3000"], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [json.loads(x) for x in proc.stdout.strip().splitlines()]
    assert lines[0]["blocks_switched_off"] > 500 and lines[0]["on_records"] > 1000 and lines[0]["on_packed_path"] > 1000
    assert lines[1]["records_built"] > 1000 and lines[1]["refused"] > 100
    assert lines[2]["emitter_rounds"] == 3000
