#!/usr/bin/env python3
"""bench.py — Msamples/s of the path-tracing hot path on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts one child process per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] — the synthetic 32x32-chunk outdoor world at
1920x1080, draw-depth 256, sun + sky, seeds from java.util.Random(0) — because that is the scene
the metric is quoted on and it fits one GPU.  A "step" is `--passes` passes (samples per pixel)
over the whole image, one launch (default 256, the most a launch carries, so the default 4 steps are
the 1024 spp BASELINE.json quotes the configuration at; every launch ends with a ~0.75 ms tail in which
the longest paths of its last samples finish, so fewer, longer launches waste less — 1 % at N = 1, 7 %
of an eighth share); scene upload is outside the timed region, the framebuffer lives in HBM.

N > 1: one process per GPU, the scene replicated, the image cut into 16x16-pixel blocks dealt
round-robin (chunky_render_set_shard), no collective on the data path, ONE RCCL reduce of the
per-rank framebuffers to rank 0 inside the timed region (the read-back).  Total work is fixed as
N grows => "scaling": "strong".

The JSON line also carries:
  roofline     — the contract figure: achieved = ALGORITHMIC bytes per sample of the reference's
                 access stream (counted by the CPU oracle on a row-sample of this same view) x samples per launch /
                 mean launch duration from HIP events on the launch stream, against the 8 TB/s HBM peak.  It is a
                 work-rate convention: the scene is cache-resident, so physical HBM traffic is ~1 % of it.  `traffic`
                 (PMC FETCH_SIZE x2 + WRITE_SIZE per launch) and `valu` (VALU issue share and lane utilisation — the
                 real limiter) come from the committed PMC summary named in `pmc_source` and are attached only when
                 that summary was collected for the kernel / launch shape of this run; otherwise they are null.
                 `limits` (same source, same condition) is the steering metric: valu_lane_frac = VALU issue share x lane
                 utilisation, the L1's tag look-ups per cycle, L2 request bandwidth against its peak, the wait share, and the
                 limiter they add up to — `frac` itself is saturated by the cache-resident scene and ranks nothing.
  per_rank     — N > 1: every rank's kernel milliseconds (HIP events) and the milliseconds of the read-back reduce.
  cpu_baseline — the C restatement of the reference kernel (oracle/port.c, kind "port") timed on
                 this box's host cores on a bounded row-sample of the same workload (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def sample_rows(height: int, n_rows: int):
    step = max(height // n_rows, 1)
    return list(range(step // 2, height, step))[:n_rows]


def oracle_row_sample(sc, seeds, rows, threads, count: bool):
    """Run the CPU oracle on whole rows of the 1080p view. Returns (samples, seconds, counters)."""
    from oracle import binding
    port = binding.port()
    h = binding.SceneHandle(sc)
    res = np.zeros(3 * sc.width * sc.height, np.float32)
    gids = (np.asarray(rows, np.int64)[:, None] * sc.width + np.arange(sc.width)[None, :]).reshape(-1).astype(np.int32)
    if count:
        port.counters(enable=True, reset=True)
        port.counters(reset=True)
    t0 = time.perf_counter()
    port.render_gids(h, seeds, gids, res=res, threads=threads)
    dt = time.perf_counter() - t0
    c = port.counters(enable=False, reset=True) if count else None
    return gids.size * len(seeds), dt, c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--passes", type=int, default=256, help="passes (spp) per step; one launch carries up to 256")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--chunks", type=int, default=32)
    ap.add_argument("--kernel", type=int, default=0, help="kernel variant (CHUNKY_OPT_KERNEL)")
    ap.add_argument("--tile", type=int, default=0, help="shard tiles: 0 = 16x16-pixel blocks (default), n > 0 = runs of n pixel indices")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the oracle row-sample too: nothing under oracle/ is loaded, built or spawned (profiler runs)")
    ap.add_argument("--dump", default="", help="rank 0 writes the final (reduced) framebuffer to this .npy file")
    ap.add_argument("--emulate-world", type=int, default=0, help="rig: render only rank 0's share of an N-GPU split on one GPU")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for rigs")
    ap.add_argument("--one-device", action="store_true", help="rig: every rank uses GPU 0 (1-GPU box, with --backend gloo)")
    ap.add_argument("--spawn", action="store_true", help="run even N = 1 as a child rank through the self-spawn path (rig)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (args.gpus > 1 or args.spawn) and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: this process has not touched the GPU (numpy only so far) and never will — it
        # starts one CHILD per GPU with the torch.distributed environment, relays rank 0's JSON line and exits with the
        # children's code.  (Under torch.distributed.run RANK is set and this branch is not taken.)
        from chunkyclplugin_amd import parallel
        sys.exit(parallel.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if world != args.gpus:
        args.gpus = world

    import torch
    import torch.distributed as dist
    from chunkyclplugin_amd import native, parallel, scenes
    from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader, RendererInstance

    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    sc = scenes.cached_outdoor_world(chunks=args.chunks, height=256, width=args.width, img_height=args.height)
    n_pix = sc.width * sc.height
    inst = RendererInstance.get(local_rank)
    loader = HipSceneLoader(inst)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    r.set_option(native.OPT_KERNEL, args.kernel)
    r.set_shard(rank, args.emulate_world or world, args.tile)
    fb = torch.zeros(3 * n_pix, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()  # the fill runs on torch's stream, the passes on the library's: order them (chunky_hip.h)
    r.set_device_buffer(fb.data_ptr())

    total_passes = (args.warmup + args.steps) * args.passes
    seeds = native.java_random_ints(total_passes)

    def barrier():
        r.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if world > 1:  # the read-back collective once, untimed, on a scratch buffer: communicator and channel set-up are not the path
        parallel.reduce_framebuffer(torch.zeros_like(fb) if args.backend == "nccl" else torch.zeros(16), dst=0)
    spp = 0
    for _ in range(args.warmup):
        r.render_passes(seeds[spp:spp + args.passes], first_buffer_spp=spp, sync=False)
        spp += args.passes
    barrier()
    r.kernel_time()  # discard warmup launches
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.render_passes(seeds[spp:spp + args.passes], first_buffer_spp=spp, sync=False)
        spp += args.passes
    r.sync()
    t_reduce = time.perf_counter()
    parallel.reduce_framebuffer(fb, dst=0)  # the read-back collective (one RCCL reduce; no-op at N=1)
    torch.cuda.synchronize()
    reduce_ms = (time.perf_counter() - t_reduce) * 1e3
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kernel_ms, launches = r.kernel_time()
    info = r.kernel_info()
    per_rank = None
    devices_seen = [local_rank]
    if world > 1:
        objs = [None] * world
        dist.all_gather_object(objs, (local_rank, torch.cuda.get_device_name(local_rank)))
        devices_seen = [o[0] for o in objs]
    if world > 1:
        mine = torch.tensor([kernel_ms, reduce_ms], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = {"kernel_ms": [round(float(g[0]), 3) for g in gathered], "reduce_ms": [round(float(g[1]), 3) for g in gathered],
                    "note": "kernel_ms = sum of this rank's launches in the timed region (HIP events); reduce_ms = host time of "
                            "the one read-back reduce incl. waiting for the slowest rank"}
    if rank == 0 and args.dump:
        np.save(args.dump, fb.cpu().numpy())

    if rank == 0:
        local_slots = int(parallel.owned_gids(n_pix, 0, args.emulate_world or world, args.tile, sc.width).size)  # pixels rank 0 renders
        # an emulated share renders only rank 0's tiles: count what was rendered, and say so
        samples = (min(local_slots, n_pix) if args.emulate_world else n_pix) * args.steps * args.passes
        value = samples / dt / 1e6
        # ---- roofline: algorithmic bytes of the reference access stream on this view ---------------
        threads = os.cpu_count() or 1
        # what a launch really carried: chunky_render_passes cuts a step into launches of at most info["passes_per_launch"]
        # passes (256 unless the staged samples would not fit); the timed region's samples over its launches is exact either way
        launch_ms = kernel_ms / max(launches, 1)
        samples_per_launch = min(local_slots, n_pix) * args.steps * args.passes // max(launches, 1)
        passes_per_launch = min(args.passes, info["passes_per_launch"])
        bytes_per_sample, n_s, rows = None, 0, []
        if not args.no_roofline:
            from oracle import binding
            binding.port(build=not args.no_cpu)  # --no-cpu runs (profiler passes) never spawn a compiler
            rows = sample_rows(sc.height, 36)
            n_s, _, ctr = oracle_row_sample(sc, seeds[:1], rows, threads, count=True)
            bytes_per_sample = binding.algorithmic_bytes(ctr)
        achieved = bytes_per_sample * samples_per_launch / (launch_ms * 1e-3) / 1e9 if (launch_ms > 0 and bytes_per_sample) else 0.0
        kernel_name = ("render_pool<%d,%d>+fold_kernel" % (info["tree"], info["pool"]) if info["pool"] >= 0 else
                       "render_waves<%d,%d,%s>" % (info["tree"], info["group"], "bvh" if info["bvh"] else "no-bvh"))
        traffic, valu, pmc_source, limits = None, None, None, None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp):
            try:
                pm = json.load(open(tp))
                # only for the launch shape the counters were collected on
                if (pm.get("kernel_info") == [info["tree"], info["group"], int(info["bvh"]), info["pool"]] and pm.get("passes_per_launch") == passes_per_launch
                        and pm.get("samples_per_launch") == samples_per_launch and args.kernel == 0):
                    traffic = pm.get("hbm_bytes_per_launch")
                    valu = pm.get("valu")
                    limits = pm.get("limits")
                    pmc_source = pm.get("source")
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/s, 32x32-chunk scene @1920x1080", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: synthetic {args.chunks}x{args.chunks}-chunk outdoor world, "
                                   f"{sc.width}x{sc.height}, draw-depth 256, sun+sky, {args.passes} spp per step",
                       "passes_per_step": args.passes, "spp_timed": args.steps * args.passes,
                       "octree_ints": int(sc.octree.size), "octree_depth": int(sc.octree_depth),
                       "parallelism": f"image tiles ({'16x16-pixel blocks' if args.tile == 0 else f'runs of {args.tile} px'}) round-robin over {world} GPU(s), scene replicated, "
                                      f"one RCCL reduce per read-back", "kernel_variant": args.kernel},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "achieved_is": "algorithmic bytes of the reference access stream / launch time (SURVEY 8d), "
                                        "not physical HBM traffic: the scene is cache-resident",
                         "physical_frac": round(traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic and launch_ms > 0 else None,
                         "valu": valu, "limits": limits, "pmc_source": pmc_source,
                         "algorithmic_bytes_per_sample": round(bytes_per_sample, 1) if bytes_per_sample else None,
                         "kernel": kernel_name, "launches": launches, "launch_ms": round(launch_ms, 4),
                         "samples_per_launch": samples_per_launch,
                         "counted_on": f"{n_s} samples ({len(rows)} rows of this view, seed 0)"},
        }
        if args.emulate_world:
            out["emulated_world"] = args.emulate_world
            out["metric"] += f" — EMULATED rank-0 share of a {args.emulate_world}-GPU split on one GPU (not a multi-GPU result)"
        # what the collective actually ran on: the backend and world size torch.distributed reports ("nccl" IS RCCL on ROCm)
        out["rccl_ranks"] = dist.get_world_size() if (world > 1 and dist.get_backend() == "nccl") else (1 if world == 1 else 0)
        out["collective"] = {"backend": dist.get_backend() if world > 1 else None, "ranks": dist.get_world_size() if world > 1 else 1,
                             "devices": sorted(set(devices_seen)),
                             "launcher": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else
                                         ("bench.py self-spawn" if "RANK" in os.environ else "none")}
        if per_rank:
            out["per_rank"] = per_rank
        if world == 1 and not args.no_cpu and not args.no_roofline:
            # bounded CPU leg: the SAME full-resolution view, whole image, P passes with P sized from a
            # calibration run so the leg costs about --cpu-seconds of wall time on all host cores
            all_rows = sample_rows(sc.height, sc.height)
            p_cpu, done, spent = 2, 0, 0.0
            for _ in range(3):  # grow the pass count until the leg costs about --cpu-seconds
                done, spent, _c = oracle_row_sample(sc, seeds[:p_cpu], all_rows, threads, count=False)
                if spent >= 0.7 * args.cpu_seconds or p_cpu >= 64:
                    break
                p_cpu = int(min(64, max(p_cpu + 1, round(p_cpu * args.cpu_seconds / max(spent, 1e-3)))))
            out["cpu_baseline"] = {"value": round(done / spent / 1e6, 4), "unit": "Msamples/s", "cores": threads,
                                   "kind": "port",
                                   "sample": f"{done} samples = the same {sc.width}x{sc.height} view, {p_cpu} pass(es), "
                                             f"{spent:.1f} s of oracle/port.c (C restatement of the reference kernel) "
                                             f"with OpenMP on all {threads} host cores"}
            out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)

    r.close()
    loader.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
