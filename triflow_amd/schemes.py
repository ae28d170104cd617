"""Temporal schemes running on the GPU (seam #2 of the reference).

Same classes, constructor arguments and call protocol as
``triflow/core/schemes.py``::

    scheme = schemes.RODASPR(model, tol=1e-1)
    t, fields = scheme(t, fields, dt, pars, hook=hook)

but a step is a handful of kernel launches on the state that already lives in
HBM (``device.py``): the F/J stencil sweep, the banded factorisation of
``I - gamma dt J``, the stage solves and the vector algebra.  The host sees at
most one scalar per step (the embedded error estimate of the adaptive
Rosenbrock schemes).  ``hook`` semantics and call sites are the reference's
(``schemes.py:139,145,224,549,558``); see ``device.py`` for how declarative
and arbitrary Python hooks are served.

* Theta        ``schemes.py:502-559``   -> ``tf_step_theta``
* ROW_general  ``schemes.py:69-238``    -> ``tf_step_row`` (+ host step control)
* ROS2 / ROS3PRw / ROS3PRL / RODASPR    tableaux of ``schemes.py:241-427``
* time_stepping ``schemes.py:33-66``    step-doubling wrapper (host logic)
* scipy_ode    ``schemes.py:430-499``   SciPy integrators fed by ``model.F`` / ``model.J``
* BDF2         new (named by BASELINE.json, absent from the reference)
"""

import logging
import os
import weakref
from functools import wraps

import numpy as np

from .device import DirichletHook, null_hook, stepper_for
from .tableaux import TABLEAUX

log = logging.getLogger(__name__)
log.addHandler(logging.NullHandler())

__all__ = ["null_hook", "DirichletHook", "time_stepping", "Theta", "ROW_general", "ROS2",
           "ROS3PRw", "ROS3PRL", "RODASPR", "BDF2", "scipy_ode"]


def _is_device_hook(hook):
    return hook is null_hook or isinstance(hook, DirichletHook) or \
        getattr(hook, "__name__", "") == "null_hook"


def _device_step(model, t, fields, pars, hook, launch, dt=None):
    """Common prologue of every scheme: resident source slot (uploading when the
    caller handed over host data), hook at ``t``, one device step into a free
    slot.  ``launch(solver, src, dst)`` issues the kernels; returns
    ``(new_fields, pars, launch result)``."""
    if _is_device_hook(hook):
        if isinstance(hook, DirichletHook):
            pars = hook.update_pars(t, pars)          # hook(t, ...) may return new parameters
        stepper = stepper_for(model, fields, pars)
        stepper.bind(fields, pars)
        stepper.set_hook(hook, t, None if dt is None else t + dt)
        src = stepper.acquire(fields)
        template = fields
    else:
        # arbitrary Python hook: run it on a host copy (reference: fields =
        # fields.copy(); fields, pars = hook(t, fields, pars))
        template, pars = hook(t, fields.copy(), pars)
        stepper = stepper_for(model, template, pars)
        stepper.bind(template, pars)
        stepper.set_hook(null_hook)
        src = stepper.acquire(template)
    dst = stepper.free_slot(exclude=(src,))
    out = launch(stepper.solver, src, dst)
    return stepper.wrap(template, dst), pars, out


def _difference_norms(a, b, ord):
    """``||a[var] - b[var]||_ord`` for every dependent variable; computed by a
    reduction kernel when both containers still live on the same GPU solver."""
    def probe(f):                       # foreign containers: host norms
        backing = getattr(f, "_device_backing", lambda: None)()
        if backing is not None and backing.valid() and f._pending_point_writes():
            backing.stepper.acquire(f)      # node assignments of a Python hook: applied on the device
            backing = f._device_backing()
        return backing
    ba, bb = probe(a), probe(b)
    if ba is not None and bb is not None and ba.stepper is bb.stepper \
            and ba.valid() and bb.valid() and ord in (2, np.inf):
        return list(ba.stepper.solver.diff_norms(ba.slot, bb.slot, ord)[0])
    return [np.linalg.norm(np.asarray(a[key]) - np.asarray(b[key]), ord)
            for key in b.dependent_variables]


#: A whole trial of the step-doubling controller as one library call (``tf_step_doubling``).  Off by
#: default since round 3: the launches are asynchronous either way and the GPU is the bottleneck, so
#: the fused call has never been faster than the loop below (config 2, N = 1e6: 0.670 against 0.657 ms
#: per trial; config 3: equal -- profiles/r03_step_doubling_trial.txt).  ``TRIFLOW_FUSED_TRIAL=1``
#: (or setting this flag) switches it on; results are bit-identical (``check_fused_step_doubling``).
FUSED_TRIALS = os.environ.get("TRIFLOW_FUSED_TRIAL", "0") == "1"


def _fused_trial(scheme, t, fields, dt_, m, pars, hook, ord):
    """One trial of the step-doubling controller as ONE call into the device library
    (``tf_step_doubling``: the coarse step, the ten fine steps and the norm of their
    difference are queued back to back, the host waits once) -- possible when the scheme
    is a fixed-step device scheme and the hook needs no host work between sub-steps
    (none, or a :class:`DirichletHook` with constant values).  Returns
    ``(fields after the fine steps, err)`` or ``None`` when the trial has to be driven
    step by step from Python (same results, eleven host round trips)."""
    desc = getattr(scheme, "_device_desc", lambda: None)()
    if desc is None or ord not in (2, np.inf) or not _is_device_hook(hook):
        return None
    if isinstance(hook, DirichletHook) and (hook.time_dependent or hook.parameters is not None):
        return None
    stepper = stepper_for(scheme._model, fields, pars)
    if stepper.nstate < 5 or stepper.solver.nsys != 1:
        return None
    stepper.bind(fields, pars)
    stepper.set_hook(hook, t, t)
    src = stepper.acquire(fields)
    coarse = stepper.free_slot(exclude=(src,))
    tmp = stepper.free_slot(exclude=(src, coarse))
    dst = stepper.free_slot(exclude=(src, coarse, tmp))
    err = stepper.solver.step_doubling(src, dst, tmp, coarse, dt_, m, desc, ord)[0]
    return stepper.wrap(fields, dst), err


def time_stepping(scheme, tol=1e-1, ord=2, m=10, reject_factor=2):
    """Step-doubling control around any scheme (``schemes.py:33-66``): a coarse
    step ``m*dt`` against ten fine steps (the reference's literal 10), error
    ``max_var ||coarse - fine||_ord / (m**2 - 1)``, new ``dt`` from
    ``sqrt(dt**2 * tol / err)``, retried while it shrinks by more than
    ``reject_factor``.  The internal ``dt`` persists across calls.  A trial runs as one
    device call when it can (:func:`_fused_trial`)."""
    state = {"dt": None}

    def one_step(t, fields, dt, pars, hook):
        dt_ = dt
        while True:
            fused = _fused_trial(scheme, t, fields, dt_, m, pars, hook, ord) if FUSED_TRIALS else None
            if fused is not None:
                fields, err = fused
                for _ in range(10):
                    t = t + dt_                  # the reference's ten additions, same bits
            else:
                _, coarse = scheme(t, fields, m * dt_, pars, hook)
                for _ in range(10):
                    t, fields = scheme(t, fields, dt_, pars, hook)
                err = max(_difference_norms(coarse, fields, ord)) / (m ** 2 - 1)
            dt_ = np.sqrt(dt ** 2 * tol / err)
            if dt_ < dt / reject_factor:
                continue        # as in the reference, the retry starts from the advanced state
            return t, fields, dt_

    @wraps(scheme)
    def adaptive_scheme(t, fields, dt, pars, hook=null_hook):
        target = t + dt
        state["dt"] = state["dt"] if state["dt"] else dt
        while t + state["dt"] <= target:
            t, fields, state["dt"] = one_step(t, fields, state["dt"] / m, pars, hook)
        if t < target:
            t, fields = scheme(t, fields, target - t, pars, hook)
        return t, fields
    return adaptive_scheme


class Theta:
    """Theta scheme (0: forward Euler, 1: backward Euler, 0.5: Crank-Nicolson),
    ``schemes.py:502-559``.  ``solver`` (seam #3) defaults to the on-device
    banded solver; a callable ``solver(A, b) -> x`` makes the scheme assemble
    ``A`` and ``b`` from the device-evaluated ``model.F`` / ``model.J`` and
    hand them to it, like the reference does."""

    def __init__(self, model, theta=1, solver=None):
        self._model, self._theta, self._solver = model, theta, solver

    def _device_desc(self):
        """What ``tf_step_doubling`` needs to run this scheme's steps itself."""
        return None if self._solver is not None else dict(kind="theta", theta=self._theta)

    def __call__(self, t, fields, dt, pars, hook=null_hook):
        if self._solver is not None:
            return self._host_solver_step(t, fields, dt, pars, hook)
        new, pars, _ = _device_step(
            self._model, t, fields, pars, hook,
            lambda solver, src, dst: solver.step_theta(src, dst, dt, self._theta), dt=dt)
        if not _is_device_hook(hook):
            new, _ = hook(t + dt, new, pars)
        return t + dt, new

    def _host_solver_step(self, t, fields, dt, pars, hook):
        import scipy.sparse as sps
        fields = fields.copy()
        fields, pars = hook(t, fields, pars)
        F = self._model.F(fields, pars)
        J = self._model.J(fields, pars)
        U = fields.uflat
        B = dt * (F - self._theta * J @ U) + U
        A = sps.identity(U.size, format="csc") - self._theta * dt * J
        fields.fill(self._solver(A, B))
        fields, _ = hook(t + dt, fields, pars)
        return t + dt, fields


class ROW_general:
    """Rosenbrock-Wanner schemes (``schemes.py:69-238``): one factorisation of
    ``I - gamma_00 dt J`` per step, ``s`` stage solves."""

    def __init__(self, model, alpha, gamma, b, b_pred=None, time_stepping=False,
                 tol=None, max_iter=None, dt_min=None, safety_factor=0.9,
                 recompute_target=True):
        self._model = model
        self._alpha = np.asarray(alpha, dtype=float)
        self._gamma = np.asarray(gamma, dtype=float)
        self._b = np.asarray(b, dtype=float)
        self._b_pred = None if b_pred is None else np.asarray(b_pred, dtype=float)
        self._s = len(b)
        self._time_control = time_stepping
        self._tol, self._max_iter, self._dt_min = tol, max_iter, dt_min
        self._safety_factor = safety_factor
        self._recompute_target = recompute_target
        self._internal_dt = None
        self._internal_iter = None
        self._interp_cache = None
        self._err = None

    def _device_desc(self):
        """What ``tf_step_doubling`` needs to run this scheme's (fixed) steps itself."""
        if self._time_control:
            return None          # the embedded-error control decides step by step on the host
        return dict(kind="row", alpha=self._alpha, gamma=self._gamma, b=self._b, hook_after=True)

    def __call__(self, t, fields, dt, pars, hook=null_hook):
        if self._time_control:
            return self._variable_step(t, fields, dt, pars, hook=hook)
        t, fields, _ = self._fixed_step(t, fields, dt, pars, hook=hook, hook_after=True,
                                        want_err=False)
        return t, fields

    def _fixed_step(self, t, fields, dt, pars, hook=null_hook, hook_after=False,
                    want_err=True, err_slot=None):
        """``schemes.py:142-174``; returns ``(t + dt, fields, err)`` with
        ``err = ||U - U_pred||_inf`` when the tableau has ``b_pred``.
        ``hook_after`` is the extra hook call of the fixed-step ``__call__``
        (``schemes.py:137-140``).  ``err_slot``: the step is only queued, its estimate stays on
        the device and ``err`` is a callable that fetches it (``tf_step_row_queued``)."""
        device_hook = _is_device_hook(hook)

        def launch(solver, src, dst):
            if err_slot is not None:
                solver.step_row_queued(src, dst, dt, self._alpha, self._gamma, self._b, self._b_pred,
                                       hook_after=hook_after and device_hook, err_slot=err_slot)
                return lambda: solver.read_err(err_slot)
            return solver.step_row(src, dst, dt, self._alpha, self._gamma, self._b,
                                   self._b_pred, hook_after=hook_after and device_hook,
                                   want_err=want_err and self._b_pred is not None)
        new, pars, err = _device_step(self._model, t, fields, pars, hook, launch, dt=dt)
        if hook_after and not device_hook:
            new, pars = hook(t + dt, new, pars)
        return t + dt, new, err

    #: The accepted trial of ``_variable_step`` that reaches the target is followed, in the reference,
    #: by a second step from the same state with ``dt = target - t`` (``schemes.py:212-217``).  Once
    #: the controller's own step exceeds the caller's ``dt`` -- the steady state of a smooth run: the
    #: trial is then made with the caller's ``dt`` (``schemes.py:188-193``) -- that second step repeats the
    #: trial with a ``dt`` that differs from it by the rounding of ``t + dt`` alone, i.e. by at most an
    #: ulp of ``t``: the state it would produce differs from the trial's by ~1e-16 relative, five
    #: orders below the solver's own forward error.  The trial's state is then taken as the landing
    #: step's (with the closing hook applied): half the work per accepted ``dt`` of the default
    #: ``Simulation`` path (profiles/r04_default_user_path.txt).  Set to False for the literal sequence.
    REUSE_TRIAL_AS_LANDING = True

    #: While the host looks at the error estimate of a trial the GPU would sit idle -- and in the steady
    #: state of a smooth run the next thing it will be asked for is known: the same step from the state
    #: this trial produced (the next call's first trial, ``dt`` the caller's).  That step is queued
    #: *before* the estimate is read (``tf_step_row_queued``), and the next call takes it if it asks for
    #: exactly that (same container, ``t``, ``dt``, parameters, hook); otherwise it is dropped.  Nothing
    #: is returned unverified: every trial's estimate is read before the controller acts on it, in the
    #: reference's order (profiles/r04_default_user_path.txt: 682 -> see there, accepted dt/s).
    QUEUE_NEXT_TRIAL = True

    def _trial(self, t, fields, dt, pars, hook, target):
        """One trial of ``_variable_step``: ``_fixed_step`` -- taken from the step the previous call
        queued ahead when it is that very step -- with the likely next one queued behind it."""
        spec, self._spec = getattr(self, "_spec", None), None
        usable = (self.QUEUE_NEXT_TRIAL and self.REUSE_TRIAL_AS_LANDING and self._recompute_target
                  and self._b_pred is not None
                  and (hook is null_hook or getattr(hook, "__name__", "") == "null_hook")
                  and getattr(fields, "_device_backing", lambda: None)() is not None)
        if not usable:
            return self._fixed_step(t, fields, dt, pars, hook)
        # (dt "the same": a driver that lands on t + dt asks for target - t, the last call's dt give or
        # take the rounding of the sum -- the tolerance of the landing step, REUSE_TRIAL_AS_LANDING)
        if spec is not None and spec["fields"] is fields and spec["t"] == t and spec["pars"] is pars \
                and abs(spec["dt"] - dt) <= np.spacing(abs(t + dt)):
            new_t, new_fields, fetch, slot = t + dt, spec["new_fields"], spec["fetch"], spec["slot"]
        else:
            slot = 1
            new_t, new_fields, fetch = self._fixed_step(t, fields, dt, pars, hook, err_slot=slot)
        # the trial that would follow if this one is accepted and lands on the target with the
        # controller's step still above the caller's dt: queued now, looked at by the next call
        if new_t >= target and dt == getattr(self, "_call_dt", None) \
                and abs((target - t) - dt) <= np.spacing(abs(target)):
            nslot = 2 if slot == 1 else 1
            nt, nf, nfetch = self._fixed_step(target, new_fields, dt, pars, hook, err_slot=nslot)
            self._spec = dict(fields=new_fields, t=target, dt=dt, pars=pars, new_t=nt, new_fields=nf,
                              fetch=nfetch, slot=nslot)
        err = fetch()
        if err > self._tol:
            self._spec = None              # rejected: what was queued behind it started from a state nobody keeps
        return new_t, new_fields, err

    def _landing_from_trial(self, new_fields, dt_used, t, target, pars, hook):
        if not self.REUSE_TRIAL_AS_LANDING or abs((target - t) - dt_used) > np.spacing(abs(target)):
            return None
        if hook is null_hook or getattr(hook, "__name__", "") == "null_hook":
            return new_fields
        if isinstance(hook, DirichletHook) and hook.parameters is None:
            backing = getattr(new_fields, "_device_backing", lambda: None)()
            if backing is None or not backing.valid():
                return None
            dep = list(self._model._dep_vars)
            backing.stepper.solver.poke(backing.slot, hook.entries(dep, target))   # hook(t + dt, ...)
            return new_fields
        return None

    def _variable_step(self, t, fields, dt, pars, hook=null_hook):
        """Embedded-error step control (``schemes.py:176-238``)."""
        if self._b_pred is None:
            raise NotImplementedError("time stepping needs the b predictor coefficients")
        if self._tol is None:
            raise ValueError("time stepping needs a tolerance")
        target = t + dt
        self._internal_iter = 0
        if self._interp_cache is not None:
            t0, t1, U0, U1 = self._interp_cache
            if t0 <= target <= t1:
                fields.fill(U0 + (U1 - U0) * ((target - t0) / (t1 - t0)))
                return target, fields
        start = 1e-6 if self._internal_dt is None else self._internal_dt
        self._call_dt = dt
        dt = self._internal_dt = min(start, dt) if self._recompute_target else start
        while True:
            self._err = None
            while self._err is None or self._err > self._tol:
                dt_used = dt
                new_t, new_fields, self._err = self._trial(t, fields, dt, pars, hook, target)
                log.debug("error: %s", self._err)
                with np.errstate(divide="ignore"):      # err == 0 -> dt = inf, as in NumPy
                    dt = self._internal_dt = (self._safety_factor * dt
                                              * np.sqrt(self._tol / self._err))
            if new_t >= target:
                self._internal_iter += 1
                if self._recompute_target:
                    landed = self._landing_from_trial(new_fields, dt_used, t, target, pars, hook)
                    if landed is not None:
                        return target, landed
                    # land exactly on the target; the closing hook call of
                    # schemes.py:224 is the step's hook_after
                    t, fields, self._err = self._fixed_step(t, fields, target - t, pars, hook,
                                                            hook_after=True)
                else:
                    U0, U1 = fields.uflat, new_fields.uflat
                    self._interp_cache = (t, new_t, U0, U1)
                    fields = fields.copy()
                    fields.fill(U0 + (U1 - U0) * ((target - t) / (new_t - t)))
                    fields, pars = hook(t, fields, pars)
                return target, fields
            t, fields = new_t, new_fields
            self._internal_iter += 1
            if self._internal_iter > (self._max_iter if self._max_iter
                                      else self._internal_iter + 1):
                raise RuntimeError("Rosebrock internal iteration "
                                   "above max iterations authorized")
            if dt < (self._dt_min if self._dt_min else dt * .5):
                raise RuntimeError("Rosebrock internal time step "
                                   "less than authorized")


class ROS2(ROW_general):
    """Second order, fixed step (``schemes.py:241-256``)."""

    def __init__(self, model):
        tab = TABLEAUX["ROS2"]
        super().__init__(model, tab.alpha, tab.gamma, tab.b, time_stepping=False)


def _adaptive_row(name, doc):
    tab = TABLEAUX[name]

    def __init__(self, model, tol=1e-1, time_stepping=True, max_iter=None, dt_min=None,
                 recompute_target=True):
        ROW_general.__init__(self, model, tab.alpha, tab.gamma, tab.b, b_pred=tab.b_pred,
                             time_stepping=time_stepping, tol=tol, max_iter=max_iter,
                             dt_min=dt_min, recompute_target=recompute_target)
    return type(name, (ROW_general,), {"__init__": __init__, "__doc__": doc})


ROS3PRw = _adaptive_row("ROS3PRw", "Third order, 3 stages, adaptive (schemes.py:259-300).")
ROS3PRL = _adaptive_row("ROS3PRL", "Fourth order, 4 stages, adaptive (schemes.py:303-353).")
RODASPR = _adaptive_row("RODASPR", "Sixth order, 6 stages, adaptive (schemes.py:356-427).")


class BDF2:
    """Linearly implicit two-step BDF, fixed step (new; scheme protocol of
    ``schemes.py:523-559``)::

        (I - 2/3 dt J(U_n)) (U_{n+1} - U_n) = 1/3 (U_n - U_{n-1}) + 2/3 dt F(U_n)

    The first call, any call whose ``dt`` differs from the previous one, and any call
    whose ``fields`` are not the container the previous call of *this* object returned
    (a restart, another trajectory) is the backward-Euler form
    ``(I - dt J)(U_{n+1} - U_n) = dt F``.  The history ``U_{n-1}`` lives on the device
    and belongs to this scheme object: several BDF2 objects can step on the same model
    (they share one ``tf_solver``) without seeing each other's history."""

    _ids = iter(range(1, 2 ** 62))

    def __init__(self, model):
        self._model = model
        self._owner = next(BDF2._ids)      # names this object's history buffer on the device
        self._last = None                  # weak reference to the container last returned
        self._solvers = []

    def __call__(self, t, fields, dt, pars, hook=null_hook):
        continuing = self._last is not None and self._last() is fields

        def launch(solver, src, dst):
            if not any(s is solver for s in self._solvers):
                self._solvers.append(solver)
            solver.step_bdf2(src, dst, dt, owner=self._owner, continuing=continuing)
        new, pars, _ = _device_step(self._model, t, fields, pars, hook, launch, dt=dt)
        if not _is_device_hook(hook):
            new, _ = hook(t + dt, new, pars)
        try:
            self._last = weakref.ref(new)
        except TypeError:                  # a foreign container type without weak references
            self._last = lambda new=new: new
        return t + dt, new

    def reset(self):
        """Forget the history: the next call starts with the backward-Euler form."""
        self._last = None

    def __del__(self):
        for solver in getattr(self, "_solvers", ()):
            try:
                solver.bdf2_release(self._owner)
            except Exception:
                pass


class _HostOdeProblem:
    """dU/dt = F(U) as SciPy's integrators want it: flat vectors in and out, evaluated by
    the device-compiled ``model.F`` / ``model.J`` (seam #1, one PCIe round trip per call).
    One instance per ``scipy_ode.__call__``: it owns the working container and the hook."""

    def __init__(self, model, work, pars, hook):
        self.model, self.work, self.pars, self.hook = model, work, pars, hook

    def _state(self, t, U):
        self.work.fill(U)
        self.work, pars = self.hook(t, self.work, self.pars)
        return self.work, pars

    def rhs(self, t, U):
        return self.model.F(*self._state(t, U))

    def dense_jacobian(self, t, U):
        return self.model.J(*self._state(t, U), sparse=False)


class scipy_ode:
    """Any ``scipy.integrate.ode`` integrator as a scheme (``schemes.py:430-499``; outside
    the accelerated path: the integrator runs on the host, dense Jacobian when
    ``jac=True``).  Kept so that scripts naming it keep running."""

    def __init__(self, model, jac=False, integrator="vode", **integrator_kwargs):
        self._model, self._use_jac = model, bool(jac)
        self._integrator, self._options = integrator, dict(integrator_kwargs)

    def __call__(self, t, fields, dt, pars, hook=null_hook):
        from scipy.integrate import ode
        start, pars = hook(t, fields, pars)
        problem = _HostOdeProblem(self._model, start, pars, hook)
        driver = ode(problem.rhs, problem.dense_jacobian if self._use_jac else None)
        driver.set_integrator(self._integrator, **self._options)
        driver.set_initial_value(start.uflat, t)
        end = problem.work
        end.fill(driver.integrate(t + dt))
        end, _ = hook(t + dt, end, pars)
        return t + dt, end
