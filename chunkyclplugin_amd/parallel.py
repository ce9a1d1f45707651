"""Image-tile sharding across the GPUs of one node — host side.

The reference is single-device; this is new.  One process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI), the scene replicated, pixels sharded:

* the image is cut into tiles — blocks of 16 x 16 pixels (`tile` 0, the default of bench.py: the shape
  the pool kernel renders in) or runs of `tile` consecutive pixel indices; tile t belongs to rank
  t % world (`chunky_render_set_shard` applies the same rule on the device: `pool_slot_gid` /
  `shard_gid` in csrc/path_state.hpp);
* every rank renders into a full-size framebuffer that stays zero outside its own tiles;
* ONE collective per read-back: `reduce(SUM)` to rank 0.  Tiles are disjoint and x + 0 = x exactly,
  so the result is bit-identical to the 1-GPU image.  Nothing is exchanged per pass.
"""
from __future__ import annotations

import numpy as np


def local_slots(n_pixels: int, rank: int, world: int, tile: int, width: int = 0) -> int:
    """Pixel slots (whole tiles, padding included) owned by `rank` — ShardView.n_local on the device.
    tile = 0: 16 x 16 blocks of an image `width` pixels wide; tile > 0: runs of `tile` pixel indices."""
    if world == 1:
        return n_pixels
    if tile == 0:
        assert width > 0 and n_pixels % width == 0, "block shards need the image width"
        n_blocks = ((width + 15) // 16) * ((n_pixels // width + 15) // 16)
        return max((n_blocks - rank + world - 1) // world, 0) * 256
    n_tiles = (n_pixels + tile - 1) // tile
    mine = (n_tiles - rank + world - 1) // world
    return max(mine, 0) * tile


def owned_gids(n_pixels: int, rank: int, world: int, tile: int = 256, width: int = 0) -> np.ndarray:
    """Global pixel indices rendered by `rank`, in the order the device queue hands them out."""
    slots = np.arange(local_slots(n_pixels, rank, world, tile, width), dtype=np.int64)
    if world == 1:
        return slots.astype(np.int32)
    if tile == 0:
        height, bw = n_pixels // width, (width + 15) // 16
        b = (slots // 256) * world + rank
        x, y = (b % bw) * 16 + (slots % 16), (b // bw) * 16 + (slots % 256) // 16
        ok = (x < width) & (y < height)
        return (y * width + x)[ok].astype(np.int32)
    t, w = slots // tile, slots % tile
    gid = (t * world + rank) * tile + w
    return gid[gid < n_pixels].astype(np.int32)


HOST_STAGING = False  # set by a caller whose RCCL communicator failed: CUDA tensors are then reduced through the host (gloo)


def reduce_framebuffer(fb, dst: int = 0, group=None):
    """The read-back collective: sum the per-rank framebuffers (disjoint tiles, zero elsewhere)
    onto rank `dst`.  `fb` is a torch tensor (CUDA for RCCL, CPU for gloo); `group` = a gloo group to use when the default
    group's RCCL communicator is not usable (HOST_STAGING); returns fb."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if fb.is_cuda and (HOST_STAGING or dist.get_backend(group) == "gloo"):  # no usable RCCL (test rigs; a failed communicator): through the host
            host = fb.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
            fb.copy_(host)
        else:
            dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return fb


def spawn_ranks(script: str, argv, world: int, timeout: float | None = None) -> int:
    """Start `world` copies of `script argv...` as CHILD processes, one per GPU, with the torch.distributed
    environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT), relay rank 0's stdout
    and return the worst exit code.  The caller must not have touched the GPU: nothing is exec'ed in place —
    the parent only waits (a process that has initialised HIP must never be replaced by another program).
    If a rank fails, the others are terminated by their own PIDs (never by pattern)."""
    import os
    import socket
    import subprocess
    import sys
    import tempfile
    import time

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, rc = [], 0
    with tempfile.TemporaryFile("w+") as out0:
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            env.setdefault("OMP_NUM_THREADS", str(max((os.cpu_count() or world) // world, 1)))
            procs.append(subprocess.Popen([sys.executable, script, *argv], env=env,
                                          stdout=out0 if rank == 0 else subprocess.DEVNULL))
        t0 = time.monotonic()
        try:
            while any(p.poll() is None for p in procs):
                if any(p.poll() not in (None, 0) for p in procs):
                    break  # a rank failed: the others would wait for it in a collective
                if timeout is not None and time.monotonic() - t0 > timeout:
                    rc = 124
                    break
                time.sleep(0.05)
        finally:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        for p in procs:
            rc = rc or (p.returncode or 0)
        out0.seek(0)
        sys.stdout.write(out0.read())
        sys.stdout.flush()
    return rc
