"""Parameter sweep: independent members of one model stepped together on a GPU
(``Ensemble``), members sharded over ranks when launched with torch.distributed
(``python -m torch.distributed.run --nproc-per-node 8 examples/parameter_sweep.py``)."""
import os
import sys

import numpy as np
from triflow_amd import Model
from triflow_amd.ensemble import Ensemble, shard_members
from triflow_amd.workloads import BENCH_MODELS

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
members = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
if world > 1:
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
    dist.init_process_group("nccl")
mine = shard_members(members, rank, world)            # member m -> rank m mod world

model = Model(*BENCH_MODELS["M3_film"])
x = np.linspace(0, 100, N, endpoint=False)
phase = 2 * np.pi * np.array(mine)[:, None] / members
h = 1 + 0.1 * np.cos(2 * np.pi * 4 * x[None, :] / 100 + phase)
fields = dict(h=h, q=h ** 3, T=np.sin(2 * np.pi * x[None, :] / 100 + phase))
pars = dict(c=0.5 + np.array(mine) / members, We=0.005 * (1 + np.array(mine) % 8), eps=.5, k=.05)

ens = Ensemble(model, x, fields, pars, periodic=True, scheme="ROS2",
               device=int(os.environ.get("LOCAL_RANK", -1)) if world > 1 else -1)
for _ in range(20):
    ens.step(1e-3)
ens.sync()
state = ens.state()                                   # [nvar][members on this rank][N]
for j, m in enumerate(mine):
    print("rank %d member %2d: c = %.3f  mean h = %.12f" % (rank, m, pars["c"][j], state[0, j].mean()))
