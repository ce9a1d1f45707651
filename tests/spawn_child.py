"""Child of tests/test_distributed_cpu.py::test_spawn_ranks: joins the gloo group the environment describes,
all-reduces its rank and lets rank 0 print one JSON line (the shape of bench.py's N > 1 start-up, without a GPU)."""
import json
import os
import sys

import torch
import torch.distributed as dist

if len(sys.argv) > 1 and sys.argv[1] == "fail" and os.environ["RANK"] == "1":
    sys.exit(7)  # a rank that dies before the rendezvous: the parent must not hang on the others
dist.init_process_group("gloo")
t = torch.tensor([float(dist.get_rank())])
dist.all_reduce(t)
if dist.get_rank() == 0:
    print(json.dumps({"world": dist.get_world_size(), "sum": t.item(), "local_rank": os.environ["LOCAL_RANK"],
                      "master": os.environ["MASTER_ADDR"]}), flush=True)
dist.destroy_process_group()
