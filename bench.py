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
import datetime
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


def step_bytes_per_node(model, scheme, stages):
    """Algorithmic bytes of one implicit step ("implicit step"):
    F+J sweep, (s-1) F-only sweeps, one factorisation (write LU + read LU:
    2*8*W*nvar^2), s solves (8*W*nvar^2 + 16*nvar each) and (s-1) band products
    J@v (8*nnz + 16*nvar); Theta / BDF-2: sweep + read U + factor/solve streaming
    + rhs in, x out, U out (+ the history read/write for BDF-2)."""
    nvar = model._nvar
    nh = len(model._help_funcs)
    nnz = len(model._J_sparse_array)
    W = model._window_range
    sweep = sweep_bytes_per_node(model)
    if scheme in ("Theta", "BDF2"):
        return sweep + 8 * nvar + 16 * W * nvar ** 2 + 24 * nvar + (16 * nvar if scheme == "BDF2" else 0)
    fsweep = 8 * (nvar + nh) + 8 * nvar
    return (sweep + (stages - 1) * fsweep + 16 * W * nvar ** 2
            + stages * (8 * W * nvar ** 2 + 16 * nvar) + (stages - 1) * (8 * nnz + 16 * nvar))


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


def config_hook(cfg):
    """The hook of a BASELINE configuration (config 5: Dirichlet values A[0] = A[-1] = 1), as the
    declarative object that the device applies in place and the oracle calls like any hook."""
    if cfg == 5:
        from triflow_amd.device import DirichletHook
        return DirichletHook(A={0: 1.0, -1: 1.0})
    return None


def _cpu_sample(job):
    """One worker of the CPU baseline: `nsteps` steps of one member; returns (seconds, final
    state in the reference's node-major order or None)."""
    cfg, N, scheme_name, fair, nsteps, want_state = job
    from oracle import numpy_path as ora          # the checker, timed as the CPU baseline
    from triflow_amd import Model, workloads
    name, fd, pars, dt, _ = workloads.config_inputs(cfg, N)
    model = Model(*workloads.model_args(name),
                  compiler=ora.fair_numpy_compiler if fair else ora.numpy_compiler)
    scheme = {"ROS2": ora.ROS2, "RODASPR": lambda m: ora.RODASPR(m, time_stepping=False),
              "ROS3PRw": lambda m: ora.ROS3PRw(m, time_stepping=False),
              "ROS3PRL": lambda m: ora.ROS3PRL(m, time_stepping=False),
              "Theta": ora.Theta, "BDF2": ora.BDF2}[scheme_name](model)
    fields = model.fields_template(**fd)
    hook = config_hook(cfg)
    kw = dict(hook=hook) if hook is not None else {}
    t = 0.0
    t0 = time.perf_counter()
    for _ in range(nsteps):
        t, fields = scheme(t, fields, dt, pars, **kw)
    el = time.perf_counter() - t0
    return el, (np.array(fields.uflat) if want_state else None)


#: steps of the CPU baseline sample = steps of the parity check of the same run
CPU_STEPS = 3


def _cpu_worker(job, conn):
    try:
        conn.send(_cpu_sample(job))
    finally:
        conn.close()


def _run_cpu_workers(job, n):
    """`n` oracle processes on the same job; returns (results, exit codes of the workers that died).
    Processes of their own: SciPy's SuperLU dies (SIGSEGV inside gssv, 32-bit work-array sizes) on the
    2e7 unknowns of config 5 at full size -- the reference's own solver call, schemes.py:557 -- and that
    must not take the device's result down with it; the exit code says what happened."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    procs = []
    for _ in range(n):
        recv, send = ctx.Pipe(duplex=False)
        p = ctx.Process(target=_cpu_worker, args=(job, send))
        p.start()
        send.close()
        procs.append((p, recv))
    res, died = [], []
    for p, recv in procs:
        try:
            res.append(recv.recv())              # (before join: a large state must be drained first)
        except EOFError:
            pass
        p.join()
        if p.exitcode != 0:
            died.append(p.exitcode)
    return res, died


def cpu_baseline(cfg, N, scheme_name, fair=False, workers=1, want_state=False):
    """The reference's algorithm (oracle = NumPy/SciPy port of the numpy-compiler
    path + SuperLU) on this box's host cores, full size.  The algorithm is single
    threaded by construction; ``workers`` > 1 integrates that many independent
    ensemble members in parallel processes (the reference's own advice for sweeps,
    user_guide.rst:125-138) and reports their aggregate rate.  ``fair``: the same
    arithmetic without the reference's per-row ``np.stack`` interleave
    (compilers.py:288), so that the comparison is not inflated by that pathology."""
    nsteps = 2 if fair else CPU_STEPS
    job = (cfg, N, scheme_name, fair, nsteps, want_state and workers <= 1)
    state = None
    # The checker runs in processes of its own: SciPy's SuperLU dies (SIGSEGV inside gssv, 32-bit
    # work-array sizes) on the 2e7 unknowns of config 5 at full size -- the reference's own solver
    # call, schemes.py:557 -- and that must not take the device's result down with it.
    res, died = _run_cpu_workers(job, max(workers, 1))
    if died:
        import signal
        segv = all(code == -signal.SIGSEGV for code in died)
        if segv and N >= 2000:
            # SuperLU cannot factorise a system of this size; the cost of the reference's algorithm is
            # linear in N: half the nodes, half the rate (the state of that run is handed back so that
            # the parity check can be made at that size)
            half = cpu_baseline(cfg, N // 2, scheme_name, fair=fair, workers=workers, want_state=want_state)
            half, half_state = half if want_state else (half, None)
            if half.get("value"):
                half.update(value=half["value"] / 2.0, extrapolated_from_nodes=half.get("extrapolated_from_nodes", N // 2),
                            worker_exit="SIGSEGV",
                            sample="the oracle process died with SIGSEGV at N=%d (SciPy SuperLU cannot factorise a system "
                                   "of this size: inside gssv, the reference's own solver call); measured at N=%d and "
                                   "halved (the algorithm is linear in N): %s" % (N, N // 2, half["sample"]))
            return (half, half_state) if want_state else half
        how = ", ".join(sorted(set("exit code %d" % c if c >= 0 else "signal %s" % signal.Signals(-c).name for c in died)))
        out = dict(value=None, unit="steps/s", cores=max(workers, 1), kind="port", worker_exit=how,
                   sample="%d %s steps of the same workload (N=%d): the oracle process ended with %s "
                          "(nothing extrapolated)" % (nsteps, scheme_name, N, how))
        return (out, None) if want_state else out
    el = max(r[0] for r in res)                     # slowest member, stepping only
    if workers <= 1:
        state = res[0][1]
    out = dict(value=max(workers, 1) * nsteps / el, unit="steps/s", cores=max(workers, 1), kind="port",
                sample="%d %s steps of the same workload (N=%d)%s, NumPy %s / SciPy SuperLU, "
                       "%s, %.1f s" % (nsteps, scheme_name, N,
                                       " by each of %d member processes" % workers if workers > 1 else "",
                                       np.__version__,
                                       "one thread per member" if workers > 1 else "single thread", el))
    return (out, state) if want_state else out


#: Full-size parity of the headline run: the device state after CPU_STEPS steps from the initial
#: condition against the oracle state of the cpu_baseline leg (the same steps, the same inputs).
#: Bounds = 100 x the value measured on MI355X at full size (config 3: 2.0e-10 after three steps,
#: backward error 4e-13, cond(I - gamma dt J) ~ 1e11 at dx = 1e-4): above them
#: the run fails.
PARITY_BOUND = {2: 2e-7, 3: 2e-8, 5: 2e-9}


def device_parity(ens, dt, nsteps, ref_uflat):
    ens.restart()
    for _ in range(nsteps):
        ens.step(dt)
    ens.sync()
    st = ens.state()                                  # [nvar][nsys][N]
    u = np.ascontiguousarray(st[:, 0, :].T).reshape(-1)
    omega, refined = ens.solver.backward_error()
    return dict(steps=nsteps, rel_err=float(np.abs(u - ref_uflat).max() / np.abs(ref_uflat).max()),
                backward_error=float(omega), refined=bool(refined),
                against="oracle/numpy_path.py (reference algorithm, NumPy + SuperLU) on the same inputs, "
                        "full size, max-norm relative difference of the state")


def scheme_api_rate(model, cfg, N, scheme_name, dt, steps=20):
    """The same workload driven through the reference-style Python protocol
    ``t, fields = scheme(t, fields, dt, pars)`` with device-resident containers
    (informational; the timed region above uses the ensemble C-ABI loop)."""
    from triflow_amd import schemes, workloads
    name, fd, pars, _, _ = workloads.config_inputs(cfg, N)
    adaptive = {n: (lambda m, n=n: getattr(schemes, n)(m, time_stepping=False)) for n in ("ROS3PRw", "ROS3PRL", "RODASPR")}
    scheme = dict({"ROS2": schemes.ROS2, "Theta": schemes.Theta, "BDF2": schemes.BDF2}, **adaptive)[scheme_name](model)
    fields, t = model.fields_template(**fd), 0.0
    for _ in range(3):
        t, fields = scheme(t, fields, dt, pars)
    fields._device_backing().stepper.solver.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        t, fields = scheme(t, fields, dt, pars)
    fields._device_backing().stepper.solver.sync()
    return steps / (time.perf_counter() - t0)


def _device_identity(torch, index):
    """What identifies the physical GPU a rank runs on (for the N > 1 line)."""
    props = torch.cuda.get_device_properties(index)
    ident = getattr(props, "uuid", None)
    bus = getattr(props, "pci_bus_id", None)
    return "%s|uuid=%s|pci=%s" % (props.name, ident, bus)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=0,
                    help="the timed K-step block is repeated this many times, each bracketed by "
                         "barrier + synchronize; value / ms_per_step are those of the median block.  "
                         "0 (default): at least 25 blocks and at least 2500 steps in all, so that the "
                         "GPU is busy for more than a second on the default workload")
    ap.add_argument("--members-per-gpu", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=(2, 3, 5))
    ap.add_argument("--nodes", type=int, default=0, help="override N (default: BASELINE size)")
    ap.add_argument("--scheme", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plain", action="store_true",
                    help="only the timed loop (profiler runs): no per-kernel pass, no Python-protocol rate")
    ap.add_argument("--cpu-workers", type=int, default=0,
                    help="also time the CPU baseline with this many member processes (ensemble runs)")
    args = ap.parse_args()

    if args.repeats <= 0:          # (a function of the arguments only: the same on every rank)
        args.repeats = max(25, -(-2500 // max(args.steps, 1)))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        raise SystemExit("bench.py --gpus %d needs %d ranks (torch.distributed.run --nproc-per-node), "
                         "WORLD_SIZE is %d" % (args.gpus, args.gpus, world))
    import torch
    dist = None
    # rehearsal on a one-GPU box: TRIFLOW_BENCH_BACKEND=gloo puts every rank on GPU 0
    backend = os.environ.get("TRIFLOW_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else 0
    # TRIFLOW_BENCH_FORCE_DIST=1: a one-rank run still goes through the process group (RCCL init,
    # broadcast, all-gather, barrier) -- the N > 1 code path on a one-GPU box (tests/test_gpu_parity.py)
    multi = world > 1 or os.environ.get("TRIFLOW_BENCH_FORCE_DIST") == "1"
    if multi:
        import torch.distributed as dist
        torch.cuda.set_device(device_index)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")          # (torch.distributed.run sets both)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(hours=1),
                                    device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(hours=1))

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
                   device=device_index, hook=config_hook(args.config),
                   nstate=4 if scheme == "BDF2" else 3)   # (BDF-2: three rotating slots, the history is read in place)
    solver = ens.solver

    def barrier():
        ens.sync()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ens.step(dt)
    # HIP events (kernel begin/end timestamps) on the roofline kernel only, so the
    # timed region is not perturbed by instrumenting every launch of a step
    sweep_kernel = {"Theta": "tfk_sweep_fj_theta", "BDF2": "tfk_sweep_fj_bdf2"}.get(scheme, "tfk_sweep_fj")
    solver.timing(kernels=[sweep_kernel])
    solver.timing_reset()
    blocks = []
    # The film model's waves steepen: at dt = 1e-3 config 3 stays smooth for about 6000 steps
    # (tools/gpu_soak.py), after which the state itself blows up.  Runs longer than MAX_RUN_STEPS
    # go back to the initial state (Ensemble.restart: one device-to-device copy of the state,
    # 24 MB per member, queued on the solver's stream) -- between timed blocks, and inside a block
    # only if a single block is longer than that.
    MAX_RUN_STEPS = 4000
    run_steps = args.warmup
    for _ in range(max(args.repeats, 1)):
        if run_steps + min(args.steps, MAX_RUN_STEPS) > MAX_RUN_STEPS and run_steps > 0:
            ens.restart()
            run_steps = 0
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):           # EXACTLY K steps between the two barriers
            if run_steps >= MAX_RUN_STEPS:
                ens.restart()
                run_steps = 0
            run_steps += 1
            ens.step(dt)
        barrier()
        blocks.append(time.perf_counter() - t0)
    sweep_ms, sweep_n = solver.timing_report().get(sweep_kernel, (0.0, 0))
    solver.timing(False)
    ens.check()                                # a singular / unstable factorisation raises here
    report, nprof = {}, 0
    if not args.plain:
        # untimed pass with every launch instrumented: per-kernel breakdown
        if run_steps + 10 > MAX_RUN_STEPS:
            ens.restart()
        nprof = min(args.steps, 10)
        solver.timing(True)
        solver.timing_reset()
        for _ in range(nprof):
            ens.step(dt)
        ens.sync()
        report = solver.timing_report()
        solver.timing(False)
    blocks = np.asarray(blocks)
    seen = None
    if multi:
        tdev = "cuda" if backend == "nccl" else "cpu"
        tb = torch.tensor(blocks, dtype=torch.float64, device=tdev)
        gathered = [torch.zeros_like(tb) for _ in range(world)]
        dist.all_gather(gathered, tb)
        per_rank = np.stack([g.cpu().numpy() for g in gathered])         # [rank][block]
        blocks = per_rank.max(axis=0)                                    # MAX over ranks, per block
        seen = [None] * world
        dist.all_gather_object(seen, dict(rank=rank, local_rank=local_rank, members=len(mine),
                                          device=_device_identity(torch, device_index),
                                          pid=os.getpid()))
    state = ens.state()
    if not np.isfinite(state).all():
        raise RuntimeError("non-finite state after the timed steps")

    # BASELINE config 4 proper on the 8-GPU node: 64 members, member m on rank m % 8, 8 members per
    # rank through the same launches (the line's `value` stays the 1-member-per-GPU weak-scaling
    # figure, so that N = 1 agrees with the one-GPU run).  TRIFLOW_BENCH_CONFIG4=1: rehearsal with
    # any number of ranks (8 members per rank).
    config4 = None
    want4 = (world == 8 and args.members_per_gpu == 1 and args.config == 3 and scheme == "ROS2"
             and not args.nodes) or os.environ.get("TRIFLOW_BENCH_CONFIG4") == "1"
    if multi and want4:
        n4 = 8 * world
        table4 = member_table(n4, None) if rank == 0 else np.zeros((n4, 3))
        table4 = broadcast_table(table4)
        mine4 = shard_members(n4, rank, world)
        _, x4, fields4, pars4, dt4, _ = build_problem(3, args.nodes or None, table4[mine4])
        ens4 = Ensemble(model, x4, fields4, pars4, True, scheme="ROS2", device=device_index, nstate=3)

        def barrier4():
            ens4.sync()
            dist.barrier()
            torch.cuda.synchronize()
        k4, nb4 = 20, 7
        for _ in range(3):
            ens4.step(dt4)
        blocks4 = []
        for _ in range(nb4):
            barrier4()
            t0 = time.perf_counter()
            for _ in range(k4):
                ens4.step(dt4)
            barrier4()
            blocks4.append(time.perf_counter() - t0)
        ens4.check()
        if not np.isfinite(ens4.state()).all():
            raise RuntimeError("config 4: non-finite state")
        tb4 = torch.tensor(blocks4, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        g4 = [torch.zeros_like(tb4) for _ in range(world)]
        dist.all_gather(g4, tb4)
        per_rank4 = np.stack([g.cpu().numpy() for g in g4])
        el4 = float(np.median(per_rank4.max(axis=0)))
        config4 = {"workload": "BASELINE config 4: %d members of config 3 (c = 0.5 + m/64, We = 0.005 (1 + m %% 8), "
                               "phase 2 pi m/64), member m on rank m %% %d" % (n4, world),
                   "members": n4, "members_per_rank": [len(shard_members(n4, r, world)) for r in range(world)],
                   "steps": k4, "blocks": nb4, "ms_per_step": el4 / k4 * 1e3,
                   "member_steps_per_s": n4 * k4 / el4,
                   "per_rank": [round(float(len(mine4) * k4 / np.median(r)), 2) for r in per_rank4]}
        ens4.close()

    if rank == 0:
        elapsed = float(np.median(blocks))
        stages = len(ens.tab.b) if ens.tab is not None else 1
        bytes_per_launch = sweep_bytes_per_node(model) * N * len(mine)
        achieved = bytes_per_launch / (sweep_ms / sweep_n * 1e-3) / 1e9 if sweep_n else None
        # Theta / BDF-2 run the sweep fused with their right-hand side: the kernel writes rhs (8*nvar)
        # where the plain sweep writes F (F stays inside rhs) and, for BDF-2, reads the history
        # U_{n-1} in its state slot (8*nvar; with two rotating slots it would also copy it)
        fused = sweep_kernel in ("tfk_sweep_fj_theta", "tfk_sweep_fj_bdf2")
        fused_extra = {"tfk_sweep_fj_theta": 0, "tfk_sweep_fj_bdf2": 8 if ens._nrot >= 3 else 16}.get(sweep_kernel, 0) * model._nvar
        fused_bytes = (sweep_bytes_per_node(model) + fused_extra) * N * len(mine)
        step_bytes = step_bytes_per_node(model, scheme, stages) * N * len(mine)
        step_gbs = step_bytes / elapsed * args.steps / 1e9
        traffic, traffic_src = None, None
        tfile = os.path.join(ROOT, "profiles", "sweep_traffic.json")
        if os.path.exists(tfile) and args.config == 3 and args.members_per_gpu == 1 \
                and not args.nodes and scheme == "ROS2":
            with open(tfile) as f:
                tj = json.load(f)
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_src = "profiles/sweep_traffic.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE " \
                          "passes of this command, not collected in this run" % tj.get("round", "r01")
        out = {
            "metric": "implicit time-steps/s, 1e6-node 3-var system (member-steps/s over all GPUs)",
            "value": n_members * args.steps / elapsed,
            "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "timing": {"repeats": len(blocks), "statistic": "median of the K-step blocks (max over ranks per block)",
                       "block_ms_min": float(blocks.min()) * 1e3, "block_ms_max": float(blocks.max()) * 1e3},
            "config": {"workload": "BASELINE config %d: %s, N=%d nodes, %s, fixed dt=%g, %s"
                                   % (args.config, name, N,
                                      "periodic" if pars["periodic"] else "clamped", dt, scheme),
                       "members_per_gpu": args.members_per_gpu,
                       "parallelism": "ensemble members sharded by rank, no data-path collective",
                       "solver_levels": solver.describe()["chunks"]},
            "roofline": {"bound": "hbm", "kernel": sweep_kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": (sweep_ms / sweep_n) if sweep_n else None},
            "roofline_step": {"bound": "hbm", "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": step_gbs / HBM_PEAK_GBS,
                              "algorithmic_bytes_per_step": step_bytes,
                              "formula": "SURVEY 8(d) implicit-step bytes: %d B/node" %
                                         step_bytes_per_node(model, scheme, stages)},
        }
        if fused and sweep_n:
            out["roofline"]["fused_bytes_per_launch"] = fused_bytes
            out["roofline"]["fused_frac"] = fused_bytes / (sweep_ms / sweep_n * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["fused_note"] = ("%s = F+J sweep + the scheme's right-hand side in one pass: 'frac' prices it by "
                                             "SURVEY's F+J formula, 'fused_frac' by the bytes it moves" % sweep_kernel)
        if report:
            out["kernels_ms_per_step"] = {k: round(v[0] / nprof, 5) for k, v in report.items()}
        if multi:
            devices = [s["device"] for s in seen]
            out["backend"] = "%s (%s)" % (backend, "RCCL over xGMI" if backend == "nccl" else "rehearsal")
            out["ranks_seen"] = len(seen)
            out["devices_seen"] = devices
            out["members_per_rank"] = [s["members"] for s in seen]
            out["steps_per_s_per_rank"] = [round(float(args.members_per_gpu * args.steps / np.median(r)), 2)
                                           for r in per_rank]
            if backend == "nccl" and len(set(devices)) != world:
                raise RuntimeError("ranks share a GPU: %s" % devices)
        if config4 is not None:
            out["config4"] = config4
            if not args.no_cpu_baseline:
                # the reference's own advice for sweeps (one process per member), on this node's cores
                import psutil
                workers = max(1, min(n4, os.cpu_count() or 1,
                                     int(psutil.virtual_memory().available // (6 << 30))))
                config4["cpu_baseline_members"] = cpu_baseline(3, N, "ROS2", workers=workers)
        if world == 1 and not args.plain and getattr(solver, "constant_jacobian", False):
            # a constant-matrix model (config 2) keeps its factorisation between steps; the reference
            # factorises in every step (schemes.py:148-149, 557): the same workload that way, so that
            # the two rates are never confused
            os.environ["TRIFLOW_REUSE_FACTOR"] = "0"
            try:
                ens_f = Ensemble(model, x, fields, pars, bool(pars["periodic"]), scheme=scheme,
                                 device=device_index, hook=config_hook(args.config), nstate=3)
            finally:
                del os.environ["TRIFLOW_REUSE_FACTOR"]
            for _ in range(args.warmup):
                ens_f.step(dt)
            rates = []
            for _ in range(7):
                ens_f.sync()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    ens_f.step(dt)
                ens_f.sync()
                rates.append(args.steps * len(mine) / (time.perf_counter() - t0))
            ens_f.close()
            out["factorising_every_step"] = {"value": float(np.median(rates)), "unit": "steps/s",
                                             "note": "`value` above reuses the factorisation of the constant matrix "
                                                     "(tf_set_constant_jacobian); this is the rate with one "
                                                     "factorisation per step, as the reference does"}
        if world == 1 and args.members_per_gpu == 1 and not args.plain:
            out["scheme_api_steps_per_s"] = scheme_api_rate(model, args.config, N, scheme, dt)
        parity_failed = None
        if world == 1 and not args.no_cpu_baseline:
            # member 0 of a one-member run is the BASELINE configuration itself: the oracle's
            # state after its timed steps pins the device state of this very run
            check = args.members_per_gpu == 1
            out["cpu_baseline"], ref_state = cpu_baseline(args.config, N, scheme, want_state=True)
            n_ref = out["cpu_baseline"].get("extrapolated_from_nodes", N)
            if check and ref_state is None:
                out["parity"] = {"skipped": "no oracle state: " + out["cpu_baseline"]["sample"]}
            elif check:
                pens = ens
                if n_ref != N:
                    # the oracle only ran at n_ref nodes (see cpu_baseline.sample): the device takes the
                    # same steps at that size
                    _, xh, fh, ph, _, _ = build_problem(args.config, n_ref, table[mine])
                    pens = Ensemble(model, xh, fh, ph, bool(ph["periodic"]), scheme=scheme, device=device_index,
                                    hook=config_hook(args.config), nstate=4 if scheme == "BDF2" else 3)
                out["parity"] = device_parity(pens, dt, CPU_STEPS, ref_state)
                out["parity"]["nodes"] = n_ref
                if pens is not ens:
                    pens.close()
                bound = PARITY_BOUND[args.config]
                out["parity"]["bound"] = bound
                if not out["parity"]["rel_err"] <= bound:
                    parity_failed = "parity: rel_err %.3e > %.1e" % (out["parity"]["rel_err"], bound)
            del ref_state
            out["cpu_baseline_fair"] = cpu_baseline(args.config, N, scheme, fair=True)
            workers = args.cpu_workers or (min(64, os.cpu_count() or 1) if args.members_per_gpu > 1 else 0)
            if workers > 1:
                out["cpu_baseline_members"] = cpu_baseline(args.config, N, scheme, workers=workers)
        print(json.dumps(out))
        if parity_failed:
            raise SystemExit(parity_failed)
    if multi:
        dist.barrier()            # (rank 0 may still have been timing the CPU baseline: the group's timeout allows for it)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
