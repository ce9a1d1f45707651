"""The N > 1 path with the HIP kernels under a real process group: bench.py itself, launched by torch.distributed.run
as 2 (and 4) ranks that share GPU 0 (`--one-device --backend gloo`: RCCL refuses two ranks on one device, so the
framebuffer reduce is staged through the host by parallel.reduce_framebuffer; everything else — one process per rank,
chunky_render_set_shard, the per-rank framebuffer tensor, the read-back collective, the timing protocol — is the code
the 2/4/8-GPU runs use).  Rank 0's reduced framebuffer must equal the single-process image bit for bit, and whole
rows of it must equal the oracle.

The ranks are started as child processes of a launcher that never touches the GPU (torch.distributed.run); nothing
here replaces a process that has initialised HIP."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import binding

from chunkyclplugin_amd import native, scenes
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, CHUNKS, PASSES = 640, 360, 8, 48


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(world, dump, extra=(), launcher=True, steps=1, passes=PASSES, small=True, backend="gloo"):
    launch = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] if launcher else [sys.executable])
    size = ["--passes", str(passes), "--width", str(W), "--height", str(H), "--chunks", str(CHUNKS)] if small else []
    detail = dump + ".detail.json"
    cmd = [*launch, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", "0",
           *size, "--one-device", "--backend", backend, "--no-cpu", "--dump", dump, "--detail", detail, *extra]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="8")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    return _line_and_detail(p.stdout, detail)


def _line_and_detail(stdout, detail):
    """The ONE stdout line must stay inside what the driver keeps of stdout (round 5's 20 KB line was lost); everything else is
    in the detail file.  Returns the detail object after checking that the line is its summary."""
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]          # rank 0 prints ONE JSON line
    assert len(lines[0].encode()) < 4000, len(lines[0])
    line, full = json.loads(lines[0]), json.load(open(detail))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == full[k], k
    assert "dropped" not in line, line["dropped"]
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["launch_ms"] == full["roofline"]["launch_ms"]
    assert line["config"]["workload"] and line["collective"]["ranks"] == full["collective"]["ranks"] and line["rccl_ranks"] == full["rccl_ranks"]
    if "image_check" in full:
        assert line["image_check"]["bit_identical"] == full["image_check"]["bit_identical"]
    if "other_configs" in full:
        assert [o["baseline_config"] for o in line["other_configs"]] == [o["baseline_config"] for o in full["other_configs"]]
    return full


@pytest.fixture(scope="module")
def single_rank_image(gpu_instance):
    sc = scenes.cached_outdoor_world(chunks=CHUNKS, height=256, width=W, img_height=H)
    loader = HipSceneLoader(gpu_instance)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, W, H)
    r.set_camera(sc.projector_type, sc.camera)
    r.render_passes(native.java_random_ints(PASSES))
    img = r.read()
    r.close()
    loader.close()
    return sc, img


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_device_reduce_to_the_single_rank_image(tmp_path, port, single_rank_image, world):
    sc, want = single_rank_image
    dump = str(tmp_path / f"fb{world}.npy")
    line = _run_bench(world, dump)
    assert line["n_gpus"] == world and line["scaling"] == "strong" and line["unit"] == "Msamples/s"
    assert len(line["per_rank"]["kernel_ms"]) == world and min(line["per_rank"]["kernel_ms"]) > 0
    got = np.load(dump)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    # and the reduced image is the oracle's on whole rows (so "equal to the single-rank image" is not two wrongs)
    rows = (5, H // 3, H // 2, H - 2)
    gids = np.concatenate([np.arange(y * W, (y + 1) * W) for y in rows]).astype(np.int32)
    ref = port.render_gids(sc, native.java_random_ints(PASSES), gids, threads=binding.usable_threads()).reshape(-1, 3)[gids]
    np.testing.assert_array_equal(got.reshape(-1, 3)[gids].view(np.uint32), ref.view(np.uint32))


def test_bench_starts_its_own_ranks_without_a_launcher(tmp_path, single_rank_image):
    """`python bench.py --gpus 2` as the driver may call it: the parent never touches the GPU, starts two child ranks itself
    and relays rank 0's one JSON line; the reduced image is the single-process image."""
    _sc, want = single_rank_image
    dump = str(tmp_path / "fb_self.npy")
    line = _run_bench(2, dump, launcher=False)
    coll = dict(line["collective"])
    assert len(coll.pop("readback_ms")) == 1
    assert line["n_gpus"] == 2 and coll == {"backend": "gloo", "called_from": "torch.distributed", "ranks": 2, "devices": [0], "launcher": "bench.py self-spawn"}
    assert line["rccl_ranks"] == 0  # gloo rig: no RCCL ranks claimed
    np.testing.assert_array_equal(np.load(dump).view(np.uint32), want.view(np.uint32))


def test_single_rank_through_the_spawn_path(tmp_path, single_rank_image):
    """N = 1 as a child rank (--spawn): same image, and the line says one rank and no collective."""
    _sc, want = single_rank_image
    dump = str(tmp_path / "fb_one.npy")
    line = _run_bench(1, dump, extra=("--spawn",), launcher=False)
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 0 and line["collective"]["launcher"] == "bench.py self-spawn"  # one process: no communicator
    np.testing.assert_array_equal(np.load(dump).view(np.uint32), want.view(np.uint32))


def test_read_back_every_merge_interval(tmp_path, gpu_instance):
    """The reference reads the buffer back every 1024 spp and starts the running mean again (OpenClPathTracingRenderer.java:158-178):
    three steps of 600 passes on two ranks cross the interval once — two read-back reduces in the timed region, and the final
    image is the mean of the passes after the restart only (seeds 1200..1799 from spp 0)."""
    dump = str(tmp_path / "fb_interval.npy")
    line = _run_bench(2, dump, steps=3, passes=600)
    assert line["config"]["readbacks_timed"] == 2 and line["config"]["spp_timed"] == 1800
    assert [len(x) for x in line["per_rank"]["reduce_ms"]] == [2, 2]
    sc = scenes.cached_outdoor_world(chunks=CHUNKS, height=256, width=W, img_height=H)
    loader = HipSceneLoader(gpu_instance)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, W, H)
    r.set_camera(sc.projector_type, sc.camera)
    r.render_passes(native.java_random_ints(1800)[1200:])
    np.testing.assert_array_equal(np.load(dump).view(np.uint32), r.read().view(np.uint32))
    r.close()
    loader.close()


def test_headline_line_checks_its_own_image(tmp_path):
    """The driver's run, shortened: the headline workload at full size on two ranks (sharing the one GPU), one step.  After the
    timed region bench.py renders 4 passes through the same shards and the same collective and compares whole rows with the
    reference build's (tests/golden/timed_rows.npz): the line carries the verdict."""
    line = _run_bench(2, str(tmp_path / "fb_full.npy"), small=False)
    assert line["config"]["baseline_config"] == 2 and line["config"]["passes_per_step"] == 256
    chk = line["image_check"]
    assert chk["bit_identical"] and chk["pixels"] == 16 * 1920 and chk["passes"] == 8 and chk["pixels_differing"] == 0, chk
    # and rank 0 then opened "all GPUs" behind one context (here: two members on GPU 0) while rank 1 waited on the store
    g = line["group_check"]
    assert "error" not in g, g
    assert g["members"] == 2 and g["peer_status"] == [0, 0] and g["gather_ms"] > 0 and g["image_check"]["bit_identical"], g


def _run_single(args):
    import tempfile
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    with tempfile.TemporaryDirectory() as td:
        detail = os.path.join(td, "detail.json")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu", "--detail", detail, *args],
                           capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        return _line_and_detail(p.stdout, detail)


def test_group_mode_of_the_bench():
    """`bench.py --group 2`: ONE process, two members behind one context (both on GPU 0 here) — what a JVM binds.  The line
    reports each member's peer-access status and the gather milliseconds, checks its image against the reference's rows, and
    runs the reference's whole loop (end_to_end) on the group."""
    line = _run_single(["--group", "2"])
    assert line["n_gpus"] == 2 and line["group"]["peer_status"] == [0, 0] and len(line["group"]["gather_ms"]) == 1
    assert line["image_check"]["bit_identical"], line["image_check"]
    e = line["end_to_end"]
    assert e["readbacks"] == 1 and e["value"] > 0 and e["cold_value"] > 0 and e["finite"], e
    # two members on one device cannot share an RCCL communicator: the line says the exchange ran on peer copies, and why
    c = line["collective"]
    assert c["backend"] == "peer-copy" and c["called_from"].startswith("libchunky_hip") and "share device 0" in c["transport"]["detail"], c
    assert line["rccl_ranks"] == 0 and line["readback"]["in_value"] and len(line["readback"]["to_host_ms"]) == 1


def test_group_mode_of_the_bench_reports_rccl_from_the_library():
    """`bench.py --group 1`: the one-member group has a real RCCL communicator, created by libchunky_hip itself (no torch.distributed
    in the process): the line reports collective.backend "rccl" called from C++, as an 8-GPU group will."""
    line = _run_single(["--group", "1"])
    c = line["collective"]
    assert c["backend"] == "rccl" and c["called_from"].startswith("libchunky_hip") and c["transport"]["name"] == "rccl-sendrecv", c
    assert "rccl 2." in c["transport"]["detail"] and line["rccl_ranks"] == 1 and line["n_gpus"] == 1
    assert line["image_check"]["bit_identical"], line["image_check"]


def test_default_line_times_every_baseline_config():
    """The driver's command, shortened: the N = 1 default line carries `other_configs` — configs[1], [3], [4], each with its value,
    launch time, contract roofline and an image check against the reference build's rows — and pays the read-back inside `value`."""
    line = _run_single([])
    legs = {o["baseline_config"]: o for o in line["other_configs"]}
    assert sorted(legs) == [1, 3, 4], line["other_configs"]
    for cfg, o in legs.items():
        assert "error" not in o, o
        assert o["value"] > 0 and o["launch_ms"] > 0 and o["steps"] >= 3 and o["image_check"]["bit_identical"], o
        assert o["roofline"]["frac"] > 0 and o["roofline"]["bound"] == "hbm" and o["roofline"]["frac_above_one"] == (o["roofline"]["frac"] > 1), o["roofline"]
    assert legs[4]["roofline"]["kernel"] == "render_pool<17,16,bvh>+fold_kernel"
    rb = line["readback"]
    assert rb["in_value"] and rb["count"] == 1 and rb["to_host_ms"][0] > 0 and line["value_hbm_resident"] >= line["value"]


@pytest.mark.parametrize("config,kernel", [(1, "render_pool<17,64,sorted>+fold_kernel"), (3, "render_pool<17,64>+fold_kernel"), (4, "render_pool<17,16,bvh>+fold_kernel")])
def test_other_baseline_configs_through_the_bench(config, kernel):
    """`bench.py --config n`: the other BASELINE configurations with the same JSON schema, each checking its own image."""
    line = _run_single(["--config", str(config)])
    assert line["config"]["baseline_config"] == config and line["roofline"]["kernel"] == kernel, line["roofline"]
    assert line["image_check"]["bit_identical"], line["image_check"]
    assert line["end_to_end"]["value"] > 0 and line["roofline"]["frac"] > 0 and line["value"] > 0
    if config == 3:  # stated with "NEE on": the emitter-sampling extension is reported beside the reference's light transport
        x = line["extension_emitter_nee"]
        assert 0 < x["value"] < line["value"] and x["kernel"].startswith("render_pool<17,32,ext>"), x
    if config == 4:  # scenes with entities also report the behind-the-ray cull (an extension) beside the reference's walk
        x = line["extension_behind_cull"]
        assert x["value"] > 1.3 * line["value"] and x["image_check"]["bit_identical"], x
    else:
        assert "extension_behind_cull" not in line


def test_a_failing_rccl_communicator_does_not_lose_the_run(tmp_path, single_rank_image):
    """Failure injection for the first contact with real multi-GPU hardware: two ranks on ONE device with the real backend (nccl =
    RCCL), which refuses duplicate GPUs at its first collective.  bench.py reports the error in the line, stages the read-backs
    through the host over its gloo control group, and still delivers the right image."""
    _sc, want = single_rank_image
    dump = str(tmp_path / "fb_rccl_fail.npy")
    line = _run_bench(2, dump, backend="nccl")
    coll = line["collective"]
    assert coll["backend"] == "gloo" and line["rccl_ranks"] == 0 and coll["rccl_failed"], coll
    np.testing.assert_array_equal(np.load(dump).view(np.uint32), want.view(np.uint32))
