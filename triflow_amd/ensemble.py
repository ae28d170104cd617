"""Ensembles / parameter sweeps: many independent 1-D systems in one solver.

The reference has no batch dimension; its guidance for parametric studies is to
pickle the model and run members in separate processes
(``source_doc/source/user_guide.rst:125-138``).  Here the members of a sweep
that live on one GPU share one ``tf_solver`` (``nsys`` systems: the chunks of
all members sit side by side in the partition-interleaved planes and go through
the same kernel launches), and a sweep is sharded over the GPUs of a node by
member index: member ``m`` belongs to rank ``m % world_size``.  Members never
exchange data; the only collective is the one-off broadcast of the parameter
table from rank 0 (``broadcast_table``).
"""

import numpy as np

from .tableaux import TABLEAUX


def shard_members(n_members, rank, world_size):
    """Indices of the ensemble members owned by ``rank`` (round robin)."""
    return list(range(rank, n_members, world_size))


def broadcast_table(table, src=0):
    """Rank ``src``'s float64 table to every rank through ``torch.distributed``
    (RCCL when the process group is 'nccl', Gloo on CPU)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(table, dtype=np.float64)
    device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.as_tensor(np.ascontiguousarray(table, dtype=np.float64)).to(device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


class Ensemble:
    """``nsys`` members of the same model on one GPU, stepped together.

    ``fields``: dict name -> array ``[nsys][N]`` (dependent variables and help
    functions); ``x``: ``[N]`` or ``[nsys][N]``; ``pars``: dict name -> scalar,
    ``[nsys]`` (one value per member) or ``[nsys][N]``.
    """

    def __init__(self, model, x, fields, pars, periodic, scheme="ROS2", theta=1.0,
                 hook=None, device=-1, **solver_opts):
        cm = getattr(model, "_device", None)
        if cm is None:
            raise RuntimeError("Ensemble needs a model compiled with the HIP compiler")
        self.model, self.compiled = model, cm
        dep = list(model._dep_vars)
        first = np.asarray(fields[dep[0]], dtype=float)
        self.nsys, self.N = (1, first.size) if first.ndim == 1 else first.shape
        x = np.broadcast_to(np.asarray(x, dtype=float), (self.nsys, self.N))
        values = []
        mask = 0
        for k, name in enumerate(cm.pars):
            v = np.asarray(pars[name], dtype=float)
            if v.ndim == 2 or (v.ndim == 1 and v.size == self.N and self.nsys != self.N):
                v = np.broadcast_to(v, (self.nsys, self.N))
                mask |= 1 << k
            values.append(v)
        # a solver of its own (not the per-shape cache of CompiledModel.solver): two ensembles
        # of the same shape must not share state slots, parameters or boundary data
        from ._capi import DeviceSolver
        from .compilers import HipBackend
        seg = cm.sweep_segment(self.N * self.nsys) if isinstance(cm.backend, HipBackend) else None
        self.solver = DeviceSolver(cm.device_model(mask, int(device), seg), self.N, nsys=self.nsys,
                                   periodic=periodic, device=device, **solver_opts)
        s = self.solver
        s.set_dx((x[:, -1] - x[:, 0]) / (self.N - 1))
        s.set_x(x)
        for k, v in enumerate(values):
            s.set_param(k, v)
        spec = s.model.spec
        dxs = (x[:, -1] - x[:, 0]) / (self.N - 1)
        if spec["host_consts"]:
            from . import codegen
            per_member = [codegen.eval_host_constants(
                spec, dxs[e], [np.asarray(v)[e] if np.ndim(v) >= 1 and np.shape(v)[0] == self.nsys
                               else v for v in values]) for e in range(self.nsys)]
            for j in range(len(spec["host_consts"])):
                s.set_param(spec["npar_model"] + j, np.array([pm[j] for pm in per_member]))
        if cm.nh:
            s.set_helpers(np.array([np.broadcast_to(fields[k], (self.nsys, self.N))
                                    for k in model._help_funcs]))
        s.set_state(0, np.array([np.broadcast_to(fields[k], (self.nsys, self.N)) for k in dep]))
        # a third (or later) state slot keeps the initial state on the device for restart()
        self._keep = s.nstate - 1 if s.nstate >= 3 else None
        if self._keep is not None:
            s.copy_state(0, self._keep)
        if hook is not None:
            s.set_dirichlet(hook.entries(dep, 0.0))
        self.scheme, self.theta = scheme, theta
        self.tab = TABLEAUX.get(scheme)
        self.cur, self.t = 0, 0.0
        # BDF-2 with three rotating slots or more: U_{n-1} is still in the slot the previous step
        # started from -- the step reads it there and the history is never copied (tf_step_bdf2_from)
        self._nrot = s.nstate if self._keep is None else self._keep
        self._bdf_prev, self._bdf_dt = -1, None

    def step(self, dt):
        """One fixed step of every member (asynchronous: returns after the launches)."""
        s, src = self.solver, self.cur
        dst = (src + 1) % (s.nstate if self._keep is None else self._keep)
        if self.scheme == "Theta":
            s.step_theta(src, dst, dt, self.theta)
        elif self.scheme == "BDF2" and self._nrot >= 3:
            same = self._bdf_dt is not None and abs(self._bdf_dt - dt) <= 1e-12 * abs(dt)
            s.step_bdf2_from(src, dst, self._bdf_prev if same else -1, dt)
            self._bdf_prev, self._bdf_dt = src, dt
        elif self.scheme == "BDF2":
            s.step_bdf2(src, dst, dt)
        else:
            s.step_row(src, dst, dt, self.tab.alpha, self.tab.gamma, self.tab.b, None,
                       hook_after=True, want_err=False)
        self.cur = dst
        self.t += dt

    def restart(self):
        """Back to the initial state and t = 0 (parameters, hook and factorisation plan stay; a
        device-to-device copy, queued like a step; needs ``nstate >= 3``): long benchmark runs
        restart instead of integrating a model past the time its solution stays smooth."""
        if self._keep is None:
            raise RuntimeError("Ensemble.restart needs nstate >= 3 (the last slot keeps the initial state)")
        self.solver.copy_state(self._keep, 0)
        if self.scheme == "BDF2":
            self.solver.bdf2_reset()
            self._bdf_prev, self._bdf_dt = -1, None
        self.cur, self.t = 0, 0.0

    def sync(self):
        """Wait for the queued steps; raises if a factorisation met a singular pivot
        block or lost accuracy (``RuntimeError``)."""
        self.solver.sync()

    check = sync

    def close(self):
        self.solver.close()

    def state(self):
        """Dependent variables, ``[nvar][nsys][N]``."""
        return self.solver.get_state(self.cur)
