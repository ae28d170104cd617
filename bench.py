#!/usr/bin/env python
"""Headline benchmark: implicit time-steps/s of the 3-variable falling-film
system (BASELINE config 3: M3, N = 1e6, periodic, upwind + ROS2) on MI355X,
with the F+J stencil sweep priced against the HBM roofline and the reference's
numpy-compiler path (CPU oracle) timed on the same box.

  python bench.py [--gpus N] [--steps K] [--warmup W]
                  [--members-per-gpu M] [--scheme ROS2|RODASPR|Theta|BDF2] [--config 2|3|5]

N > 1: launched by the driver through torch.distributed.run, one rank per GPU;
every rank integrates its own members of the parameter sweep of BASELINE
config 4 (member m -> rank m % N, no per-step communication), rank 0
broadcasts the parameter table once over RCCL.  Weak scaling: the per-GPU work
(--members-per-gpu members of 1e6 nodes) is fixed as N grows.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md)


def sweep_bytes_per_node(model):
    """Algorithmic bytes of the F+J sweep:
    8*(nvar + nh + vector parameters) read + 8*nvar F + 8*nnz J."""
    nvar = model._nvar
    nh = len(model._help_funcs)
    nnz = len(model._J_sparse_array)
    return 8 * (nvar + nh) + 8 * nvar + 8 * nnz


def member_table(n_members, base):
    """Parameter sweep of BASELINE config 4: c = 0.5 + m/64,
    We = 0.005 * (1 + m % 8), initial-condition phase 2*pi*m/64."""
    m = np.arange(n_members)
    return np.stack([0.5 + m / 64.0, 0.005 * (1 + m % 8), 2 * np.pi * m / 64.0], axis=1)


def build_problem(cfg, N, table):
    from triflow_amd import workloads
    name, fd, pars, dt, scheme = workloads.config_inputs(cfg, N)
    nm = table.shape[0]
    fields = {k: np.repeat(v[None, :], nm, axis=0) for k, v in fd.items() if k != "x"}
    pars = dict(pars)
    if cfg == 3:
        x = fd["x"]
        h = 1 + 0.1 * np.cos(2 * np.pi * 4 * x[None, :] / 100 + table[:, 2:3])
        fields = dict(h=h, q=h ** 3,
                      T=np.repeat(np.sin(2 * np.pi * x / 100)[None, :], nm, axis=0))
        if nm > 1 or os.environ.get("WORLD_SIZE", "1") != "1":
            pars["c"] = table[:, 0].copy()
            pars["We"] = table[:, 1].copy()
    return name, fd["x"], fields, pars, dt, scheme


def _cpu_sample(job):
    """One worker of the CPU baseline: `nsteps` steps of one member; returns seconds."""
    cfg, N, scheme_name, fair, nsteps = job
    from oracle import numpy_path as ora          # the checker, timed as the CPU baseline
    from triflow_amd import Model, workloads
    name, fd, pars, dt, _ = workloads.config_inputs(cfg, N)
    model = Model(*workloads.model_args(name),
                  compiler=ora.fair_numpy_compiler if fair else ora.numpy_compiler)
    scheme = {"ROS2": ora.ROS2, "RODASPR": lambda m: ora.RODASPR(m, time_stepping=False),
              "Theta": ora.Theta, "BDF2": ora.BDF2}[scheme_name](model)
    fields = model.fields_template(**fd)
    t = 0.0
    t0 = time.perf_counter()
    for _ in range(nsteps):
        t, fields = scheme(t, fields, dt, pars)
    return time.perf_counter() - t0


def cpu_baseline(cfg, N, scheme_name, fair=False, workers=1):
    """The reference's algorithm (oracle = NumPy/SciPy port of the numpy-compiler
    path + SuperLU) on this box's host cores, full size.  The algorithm is single
    threaded by construction; ``workers`` > 1 integrates that many independent
    ensemble members in parallel processes (the reference's own advice for sweeps,
    user_guide.rst:125-138) and reports their aggregate rate.  ``fair``: the same
    arithmetic without the reference's per-row ``np.stack`` interleave
    (compilers.py:288), so that the comparison is not inflated by that pathology."""
    nsteps = 2 if fair else 3
    job = (cfg, N, scheme_name, fair, nsteps)
    if workers <= 1:
        el = _cpu_sample(job)
    else:
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(workers) as pool:
            el = max(pool.map(_cpu_sample, [job] * workers))       # slowest member, stepping only
    return dict(value=max(workers, 1) * nsteps / el, unit="steps/s", cores=max(workers, 1), kind="port",
                sample="%d %s steps of the same workload (N=%d)%s, NumPy %s / SciPy SuperLU, "
                       "%s, %.1f s" % (nsteps, scheme_name, N,
                                       " by each of %d member processes" % workers if workers > 1 else "",
                                       np.__version__,
                                       "one thread per member" if workers > 1 else "single thread", el))


def scheme_api_rate(model, cfg, N, scheme_name, dt, steps=20):
    """The same workload driven through the reference-style Python protocol
    ``t, fields = scheme(t, fields, dt, pars)`` with device-resident containers
    (informational; the timed region above uses the ensemble C-ABI loop)."""
    from triflow_amd import schemes, workloads
    name, fd, pars, _, _ = workloads.config_inputs(cfg, N)
    scheme = {"ROS2": schemes.ROS2, "Theta": schemes.Theta, "BDF2": schemes.BDF2,
              "RODASPR": lambda m: schemes.RODASPR(m, time_stepping=False)}[scheme_name](model)
    fields, t = model.fields_template(**fd), 0.0
    for _ in range(3):
        t, fields = scheme(t, fields, dt, pars)
    fields._device_backing().stepper.solver.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        t, fields = scheme(t, fields, dt, pars)
    fields._device_backing().stepper.solver.sync()
    return steps / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--members-per-gpu", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=(2, 3, 5))
    ap.add_argument("--nodes", type=int, default=0, help="override N (default: BASELINE size)")
    ap.add_argument("--scheme", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-workers", type=int, default=0,
                    help="also time the CPU baseline with this many member processes (ensemble runs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    # rehearsal on a one-GPU box: TRIFLOW_BENCH_BACKEND=gloo puts every rank on GPU 0
    backend = os.environ.get("TRIFLOW_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(device_index)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from triflow_amd import Model, workloads
    from triflow_amd.ensemble import Ensemble, broadcast_table, shard_members

    n_members = world * args.members_per_gpu
    table = member_table(n_members, None) if rank == 0 else np.zeros((n_members, 3))
    table = broadcast_table(table)                    # the one collective of the run
    mine = shard_members(n_members, rank, world)
    name, x, fields, pars, dt, default_scheme = build_problem(
        args.config, args.nodes or None, table[mine])
    scheme = args.scheme or default_scheme
    N = x.size
    model = Model(*workloads.model_args(name))
    ens = Ensemble(model, x, fields, pars, bool(pars["periodic"]), scheme=scheme,
                   device=device_index, hook=None, nstate=2)
    solver = ens.solver

    def barrier():
        ens.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ens.step(dt)
    # HIP events (kernel begin/end timestamps) on the roofline kernel only, so the
    # timed region is not perturbed by instrumenting all ~70 launches per step
    solver.timing(kernels=["tfk_sweep_fj"])
    solver.timing_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ens.step(dt)
    barrier()
    elapsed = time.perf_counter() - t0
    sweep_ms, sweep_n = solver.timing_report().get("tfk_sweep_fj", (0.0, 0))
    # untimed pass with every launch instrumented: per-kernel breakdown
    nprof = min(args.steps, 10)
    solver.timing(True)
    solver.timing_reset()
    for _ in range(nprof):
        ens.step(dt)
    ens.sync()
    report = solver.timing_report()
    solver.timing(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64,
                            device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    state = ens.state()
    if not np.isfinite(state).all():
        raise RuntimeError("non-finite state after the timed steps")

    if rank == 0:
        bytes_per_launch = sweep_bytes_per_node(model) * N * len(mine)
        achieved = bytes_per_launch / (sweep_ms / sweep_n * 1e-3) / 1e9 if sweep_n else None
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "sweep_traffic.json")
        if os.path.exists(tfile) and args.config == 3 and args.members_per_gpu == 1 \
                and not args.nodes:
            with open(tfile) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        out = {
            "metric": "implicit time-steps/s, 1e6-node 3-var system (member-steps/s over all GPUs)",
            "value": n_members * args.steps / elapsed,
            "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config %d: %s, N=%d nodes, %s, fixed dt=%g, %s"
                                   % (args.config, name, N,
                                      "periodic" if pars["periodic"] else "clamped", dt, scheme),
                       "members_per_gpu": args.members_per_gpu,
                       "parallelism": "ensemble members sharded by rank, no data-path collective",
                       "solver_levels": solver.describe()["chunks"]},
            "roofline": {"bound": "hbm", "kernel": "tfk_sweep_fj",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": (sweep_ms / sweep_n) if sweep_n else None},
            "kernels_ms_per_step": {k: round(v[0] / nprof, 5) for k, v in report.items()},
        }
        if world == 1 and args.members_per_gpu == 1:
            out["scheme_api_steps_per_s"] = scheme_api_rate(model, args.config, N, scheme, dt)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, N, scheme)
            out["cpu_baseline_fair"] = cpu_baseline(args.config, N, scheme, fair=True)
            if args.cpu_workers > 1:
                out["cpu_baseline_members"] = cpu_baseline(args.config, N, scheme, workers=args.cpu_workers)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
