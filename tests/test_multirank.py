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
n_members, N = int(os.environ.get("TF_TEST_MEMBERS", "5")), int(os.environ.get("TF_TEST_NODES", "96"))
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


def run_world(world, out, members=5, nodes=96, port=29517):
    script = os.path.join(out, "worker.py")
    with open(script, "w") as f:
        f.write(WORKER.format(root=ROOT, out=out))
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   TF_TEST_MEMBERS=str(members), TF_TEST_NODES=str(nodes))
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


def test_eight_ranks_config4_table(tmp_path):
    """The shape of BASELINE config 4 (64 members over 8 ranks, 8 per rank, user_guide.rst:125-138)
    at a tiny grid: every member is integrated by exactly one rank and equals the one-process
    result bit for bit; the parameter table reaches every rank by the one broadcast."""
    from triflow_amd.ensemble import shard_members
    d1, d8 = str(tmp_path / "w1"), str(tmp_path / "w8")
    os.makedirs(d1)
    os.makedirs(d8)
    run_world(1, d1, members=64, nodes=40, port=29521)
    run_world(8, d8, members=64, nodes=40, port=29523)
    one, eight = gather(1, d1), gather(8, d8)
    assert sorted(one) == sorted(eight) == list(range(64))
    for rank in range(8):
        assert list(np.load(os.path.join(d8, "members%d.npy" % rank))) == shard_members(64, rank, 8)
    for m in one:
        assert np.isfinite(one[m]).all()
        assert np.array_equal(one[m], eight[m]), m
    assert not np.array_equal(one[0], one[9])


COLD_CACHE_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
from triflow_amd import Model, compilers, workloads
compilers.CACHE_DIR = {cache!r}
model = Model(*workloads.model_args("M2_diff"), hold_compilation=True)
hsaco, spec = compilers.build_code_object(model, 0, seg=4)
with open(hsaco, "rb") as f:
    assert f.read(4) == b"\x7fELF"
print("RESULT", hsaco, compilers.BUILD_COUNT)
'''


def test_ranks_on_a_cold_cache_compile_once(tmp_path):
    """8 ranks of a sweep that start on a cold code-object cache (``bench.py --gpus 8`` on a fresh
    node): one of them runs hipcc, the others wait for the lock and load the same file; nothing
    half written, no temporary left behind."""
    cache = str(tmp_path / "cache")
    script = str(tmp_path / "worker.py")
    with open(script, "w") as f:
        f.write(COLD_CACHE_WORKER.format(root=ROOT, cache=cache))
    procs = [subprocess.Popen([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for _ in range(8)]
    paths, builds = set(), 0
    for p in procs:
        text, _ = p.communicate(timeout=600)
        assert p.returncode == 0, text.decode()[-3000:]
        line = [ln for ln in text.decode().splitlines() if ln.startswith("RESULT")][-1].split()
        paths.add(line[1])
        builds += int(line[2])
    assert len(paths) == 1 and builds == 1, (paths, builds)
    left = sorted(os.listdir(cache))
    assert not [n for n in left if n.endswith(".tmp") or n.endswith(".lock")], left
    assert sum(n.endswith(".hsaco") for n in left) == 1, left


def test_shard_members():
    from triflow_amd.ensemble import shard_members
    owned = [shard_members(64, r, 8) for r in range(8)]
    assert sorted(sum(owned, [])) == list(range(64))
    assert all(len(o) == 8 for o in owned)
    assert owned[3][:3] == [3, 11, 19]
