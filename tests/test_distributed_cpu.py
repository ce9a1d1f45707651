"""The N>1 path on CPU: two processes over gloo.  Each rank renders only the pixels it owns (the
oracle stands in for the device kernel here — tests may use it as the renderer of record), the
product's sharding rule + read-back collective (chunkyclplugin_amd/parallel.py) assemble the image
on rank 0, which must equal the single-process image bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_scenes as gs
from chunkyclplugin_amd import parallel, scenes


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tile, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import binding
        sc = gs.make("outdoor")
        n = sc.width * sc.height
        seeds = scenes.java_random_ints(3)
        fb = np.zeros(3 * n, np.float32)
        gids = parallel.owned_gids(n, rank, world, tile, sc.width)
        binding.port().render_gids(sc, seeds, gids, res=fb, threads=2)
        t = torch.from_numpy(fb)
        parallel.reduce_framebuffer(t, dst=0)
        if rank == 0:
            np.save(out_path, t.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tile", [0, 256, 100])
def test_two_ranks_reduce_to_the_single_rank_image(tmp_path, port, tile):
    out = str(tmp_path / "fb.npy")
    mp.spawn(_worker, args=(2, _free_port(), tile, out), nprocs=2, join=True)
    got = np.load(out)
    sc = gs.make("outdoor")
    want = port.render_passes(sc, scenes.java_random_ints(3))
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("world,tile,n", [(1, 256, 1000), (2, 256, 3072), (3, 64, 2000), (8, 256, 1920 * 1080), (8, 100, 777)])
def test_tiles_partition_the_image(world, tile, n):
    seen = np.zeros(n, np.int32)
    for r in range(world):
        g = parallel.owned_gids(n, r, world, tile)
        assert parallel.local_slots(n, r, world, tile) >= g.size
        seen[g] += 1
        if world > 1 and g.size:
            assert ((g // tile) % world == r).all()
    assert (seen == 1).all()


@pytest.mark.parametrize("world,width,height", [(2, 64, 40), (3, 100, 37), (8, 1920, 1080), (5, 16, 16), (4, 15, 3)])
def test_blocks_partition_the_image(world, width, height):
    """tile = 0: the 16 x 16 blocks of the image (edge blocks partial), block b to rank b % world."""
    n = width * height
    seen = np.zeros(n, np.int32)
    for r in range(world):
        g = parallel.owned_gids(n, r, world, 0, width)
        assert parallel.local_slots(n, r, world, 0, width) >= g.size
        seen[g] += 1
        if g.size:
            b = (g // width // 16) * ((width + 15) // 16) + (g % width) // 16
            assert (b % world == r).all()
    assert (seen == 1).all()


def test_spawn_ranks_relays_rank0_and_exit_codes(capfd):
    """`python bench.py --gpus N` without a launcher: parallel.spawn_ranks starts N children with the torch.distributed
    environment, relays rank 0's line and reports failure without hanging when one rank dies."""
    import json
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spawn_child.py")
    assert parallel.spawn_ranks(child, [], 3, timeout=120) == 0
    lines = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    msg = json.loads(lines[0])
    assert msg == {"world": 3, "sum": 3.0, "local_rank": "0", "master": "127.0.0.1"}
    assert parallel.spawn_ranks(child, ["fail"], 2, timeout=120) != 0
