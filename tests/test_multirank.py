"""N > 1 path on CPU: two Gloo ranks shard an ensemble by member index, rank 0
broadcasts the parameter table, every rank integrates its own members (through
the host emulation of the kernels) and the union equals the single-process
result bit for bit.  There is no per-step communication to test: members are
independent."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch.distributed as dist
from functools import partial
from oracle import corpus
from tests.emu.build_emu import EmuBackend
from triflow_amd import Model
from triflow_amd.compilers import hip_compiler
from triflow_amd.ensemble import Ensemble, broadcast_table, shard_members

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n_members, N = 5, 96
table = np.zeros((n_members, 2))
if rank == 0:
    m = np.arange(n_members)
    table = np.stack([0.5 + m / 64.0, 0.005 * (1 + m % 8)], axis=1)
table = broadcast_table(table)
mine = shard_members(n_members, rank, world)
name, fd, pars, dt, _ = corpus.config_inputs(3, N)
model = Model(*corpus.model_args(name), compiler=partial(hip_compiler, backend=EmuBackend()))
fields = {{k: np.repeat(fd[k][None, :], len(mine), axis=0) * (1 + 0.01 * np.array(mine)[:, None])
          for k in ("h", "q", "T")}}
pars = dict(pars, c=table[mine, 0], We=table[mine, 1])
ens = Ensemble(model, fd["x"], fields, pars, True, scheme="ROS2", m1=8, m_upper=3)
for _ in range(3):
    ens.step(dt)
ens.sync()
np.save(os.path.join({out!r}, "rank%d.npy" % rank), ens.state())
np.save(os.path.join({out!r}, "members%d.npy" % rank), np.array(mine))
dist.barrier()
dist.destroy_process_group()
'''


def run_world(world, out):
    script = os.path.join(out, "worker.py")
    with open(script, "w") as f:
        f.write(WORKER.format(root=ROOT, out=out))
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, script], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out_text, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out_text.decode()[-3000:]


def gather(world, out):
    res = {}
    for rank in range(world):
        state = np.load(os.path.join(out, "rank%d.npy" % rank))        # [nvar][nsys][N]
        for j, m in enumerate(np.load(os.path.join(out, "members%d.npy" % rank))):
            res[int(m)] = state[:, j, :]
    return res


def test_two_ranks_equal_one(tmp_path):
    d1, d2 = str(tmp_path / "w1"), str(tmp_path / "w2")
    os.makedirs(d1)
    os.makedirs(d2)
    run_world(1, d1)
    run_world(2, d2)
    one, two = gather(1, d1), gather(2, d2)
    assert sorted(one) == sorted(two) == list(range(5))
    for m in one:
        assert np.isfinite(one[m]).all()
        assert np.array_equal(one[m], two[m]), m
    # members really differ (the parameter table arrived)
    assert not np.array_equal(one[0], one[3])


def test_shard_members():
    from triflow_amd.ensemble import shard_members
    owned = [shard_members(64, r, 8) for r in range(8)]
    assert sorted(sum(owned, [])) == list(range(64))
    assert all(len(o) == 8 for o in owned)
    assert owned[3][:3] == [3, 11, 19]
