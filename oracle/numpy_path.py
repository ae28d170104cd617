"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy/SciPy restatement of the reference's numpy-compiler hot path, used only
as the checker in ``tests/``, in ``__graft_entry__.smoke()`` and as the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``triflow_amd/`` imports
this module; the product path has no CPU fallback.

Parity status: PINNED.  ``oracle/gen_golden.py`` imported the reference's own
``compilers.py`` / ``routines.py`` / ``model.py`` / ``schemes.py`` from
``/root/reference`` in the build container and stored their outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against those vectors (F/J bit-exact, steps to 1e-12).  ``BDF2`` is the one
exception: the reference has no such scheme, its parity is
unpinned by the reference and is anchored on convergence order and on
``scipy.integrate.ode('vode', method='bdf')`` instead.

Every function cites the reference lines it follows
(paths relative to ``/root/reference``).
"""

from functools import partial

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spsla
from sympy import lambdify


# --------------------------------------------------------------------------
# compiler: triflow/core/compilers.py:181-332
# --------------------------------------------------------------------------
def _lambdify_modules():
    """Name overrides handed to ``lambdify`` (compilers.py:196-220).

    ``Heaviside`` is identically one there (``np.where(a < 0, 1, 1)``,
    compilers.py:204-205); SymPy >= 1.9 prints it with a second argument, hence
    the ``*_``.  ``amax`` / ``amin`` take one tuple argument (compilers.py:196-202);
    newer SymPy prints ``reduce(maximum, [...])`` and never calls them.
    """
    def np_min(args):
        a, b = args
        return np.where(a < b, a, b)

    def np_max(args):
        a, b = args
        return np.where(a < b, b, a)

    def np_heaviside(a, *_):
        return np.where(a < 0, 1, 1)

    return [{"amax": np_max, "amin": np_min, "Heaviside": np_heaviside}, "numpy"]


def numpy_compiler(model, faithful_interleave=True):
    """``compiler(model) -> (F_function, J_function)`` (compilers.py:181-224)."""
    f_func = lambdify(model._symbolic_args, model.F_array.tolist(),
                      modules=_lambdify_modules())
    j_func = lambdify(model._symbolic_args, model._J_sparse_array.tolist(),
                      modules=_lambdify_modules())
    return (partial(compute_F, model, f_func, faithful_interleave=faithful_interleave),
            partial(compute_J, model, j_func))


def fair_numpy_compiler(model):
    """Same arithmetic without the reference's per-row ``np.stack`` (see compute_F)."""
    return numpy_compiler(model, faithful_interleave=False)


def stencil_views(model, *input_args):
    """Ghost-cell padding and shifted views (compilers.py:227-278).

    ``dx = (x[-1] - x[0]) / (N - 1)``; a ``dx`` entry of the parameter dict is
    ignored (compilers.py:234-237).  Periodic: wrap ``mp`` cells; otherwise
    replicate the edge values (compilers.py:257-264).
    """
    names = [*model._indep_vars, *model._dep_vars, *model._help_funcs,
             *model._pars, "periodic"]
    env = dict(zip(names, input_args))
    x = env["x"]
    N = x.size
    env["dx"] = (x[-1] - x[0]) / (N - 1)
    periodic = env["periodic"]
    lo, hi = model._bounds
    mp = (model._window_range - 1) // 2
    for name in model._symb_vars_with_spatial_diff_order:
        a = env[name]
        if periodic:
            ext = np.concatenate([a[lo:], a, a[:hi]])
        else:
            ext = np.concatenate([[a[0]] * mp, a, [a[-1]] * mp])
        for off in range(lo, hi + 1):
            key = name if off == 0 else "%s_%s%i" % (name, "m" if off < 0 else "p", abs(off))
            env[key] = ext[off - lo: ext.size + off - hi]
    return env, N, mp, periodic


def compute_F(model, f_func, *input_args, faithful_interleave=True):
    """compilers.py:281-289.  Output ``F[node * nvar + eq]``."""
    env, N, mp, periodic = stencil_views(model, *input_args)
    F = f_func(*[env[key] for key in model._args])
    F = np.concatenate([np.broadcast_to(f, (N,)) for f in F]).reshape((model._nvar, N)).T
    if faithful_interleave:
        # the reference iterates the N rows in Python here (compilers.py:288);
        # kept so that the timed CPU baseline is the reference's algorithm
        return np.stack(F).flatten()
    return np.ascontiguousarray(F).reshape(-1)


def jacobian_values(model, j_func, *input_args):
    """The ``[N, nnzJ]`` value table of compilers.py:295-301."""
    env, N, mp, periodic = stencil_views(model, *input_args)
    J = j_func(*[env[key] for key in model._args])
    J = np.stack([np.repeat(j, N) if np.ndim(j) == 0 else j for j in J])
    return J.T.reshape(N, -1), N, mp, periodic


def jacobian_pattern(nvar, window, sparse_indices, N, periodic):
    """(rows, cols) of every stored value (compilers.py:303-328).

    Symbolic entry ``k``: equation ``k % nvar``, column slot ``c = k // nvar``,
    variable ``c % nvar``, node offset ``c // nvar - mp``.  The value of node
    ``p`` lands in row ``p*nvar + eq`` and column ``g(p + off)*nvar + var`` where
    ``g`` wraps (periodic) or clamps to ``[0, N-1]``.
    """
    mp = (window - 1) // 2
    k = np.asarray(sparse_indices)
    eq, slot = k % nvar, k // nvar
    var, off = slot % nvar, slot // nvar - mp
    node = np.arange(N)[:, None]
    neigh = node + off[None, :]
    neigh = neigh % N if periodic else np.clip(neigh, 0, N - 1)
    rows = node * nvar + eq[None, :]
    cols = neigh * nvar + var[None, :]
    return rows.ravel(), cols.ravel()


def compute_J(model, j_func, *input_args):
    """compilers.py:292-332: COO -> CSC, duplicates summed (compilers.py:330-331)."""
    vals, N, mp, periodic = jacobian_values(model, j_func, *input_args)
    # the reference rebuilds this step-invariant pattern on every call
    # (compilers.py:303-328); so does its timed restatement
    rows, cols = jacobian_pattern(model._nvar, model._window_range,
                                  model._sparse_indices[0], N, periodic)
    n = N * model._nvar
    return sps.csc_matrix((vals.ravel(), (rows, cols)), shape=(n, n))


# --------------------------------------------------------------------------
# schemes: triflow/core/schemes.py
# --------------------------------------------------------------------------
def null_hook(t, fields, pars):
    return fields, pars


def time_stepping(scheme, tol=1e-1, ord=2, m=10, reject_factor=2):
    """Step-doubling wrapper (schemes.py:33-66).  The fine loop runs the
    literal 10 sub-steps (schemes.py:40), whatever ``m`` is."""
    state = {"dt": None}

    def one_step(t, fields, dt, pars, hook):
        dt_ = dt
        while True:
            _, coarse = scheme(t, fields, m * dt_, pars, hook)
            for _ in range(10):
                t, fields = scheme(t, fields, dt_, pars, hook)
            err = max(np.linalg.norm(coarse[key] - fields[key], ord) / (m ** 2 - 1)
                      for key in fields.dependent_variables)
            dt_ = np.sqrt(dt ** 2 * tol / err)
            if dt_ < dt / reject_factor:
                continue
            return t, fields, dt_

    def adaptive(t, fields, dt, pars, hook=null_hook):
        target = t + dt
        state["dt"] = state["dt"] if state["dt"] else dt
        while t + state["dt"] <= target:
            t, fields, state["dt"] = one_step(t, fields, state["dt"] / m, pars, hook)
        if t < target:
            t, fields = scheme(t, fields, target - t, pars, hook)
        return t, fields
    return adaptive


class Theta:
    """schemes.py:502-559."""

    def __init__(self, model, theta=1, solver=spsla.spsolve):
        self._model, self._theta, self._solver = model, theta, solver

    def __call__(self, t, fields, dt, pars, hook=null_hook):
        fields = fields.copy()
        fields, pars = hook(t, fields, pars)
        F = self._model.F(fields, pars)
        J = self._model.J(fields, pars)
        U = fields.uflat
        B = dt * (F - self._theta * J @ U) + U              # schemes.py:553
        A = sps.identity(U.size, format="csc") - self._theta * dt * J
        fields.fill(self._solver(A, B))
        fields, _ = hook(t + dt, fields, pars)
        return t + dt, fields


class ROW_general:
    """Rosenbrock-Wanner family (schemes.py:69-238)."""

    def __init__(self, model, alpha, gamma, b, b_pred=None, time_stepping=False,
                 tol=None, max_iter=None, dt_min=None, safety_factor=0.9,
                 recompute_target=True):
        self._model = model
        self._alpha, self._gamma, self._b, self._b_pred = alpha, gamma, b, b_pred
        self._s = len(b)
        self._time_control = time_stepping
        self._tol, self._max_iter, self._dt_min = tol, max_iter, dt_min
        self._safety_factor = safety_factor
        self._recompute_target = recompute_target
        self._internal_dt = None
        self._interp_cache = None
        self.n_fixed_steps = 0

    def __call__(self, t, fields, dt, pars, hook=null_hook):
        if self._time_control:
            return self._variable_step(t, fields, dt, pars, hook)
        t, fields, _ = self._fixed_step(t, fields, dt, pars, hook)
        fields, pars = hook(t, fields, pars)                 # schemes.py:137-140
        return t, fields

    def _fixed_step(self, t, fields, dt, pars, hook=null_hook):
        """schemes.py:142-174."""
        self.n_fixed_steps += 1
        fields = fields.copy()
        fields, pars = hook(t, fields, pars)
        J = self._model.J(fields, pars)
        U0 = fields.uflat
        A = sps.eye(U0.size, format="csc") - self._gamma[0, 0] * dt * J
        luf = spsla.factorized(A)
        ks = []
        stage = fields.copy()
        for i in range(self._s):
            stage.fill(U0 + sum(self._alpha[i, j] * ks[j] for j in range(i)))
            F = self._model.F(stage, pars)
            rhs = dt * F
            if i > 0:
                rhs = rhs + dt * (J @ sum(self._gamma[i, j] * ks[j] for j in range(i)))
            ks.append(luf(rhs))
        U = U0 + sum(bi * ki for bi, ki in zip(self._b, ks))
        err = None
        if self._b_pred is not None:
            # U_pred is built from the *updated* U (schemes.py:167-170), so the
            # estimate is || sum_i b_pred_i k_i ||_inf
            U_pred = U + sum(bi * ki for bi, ki in zip(self._b_pred, ks))
            err = np.linalg.norm(U - U_pred, np.inf)
        fields.fill(U)
        return t + dt, fields, err

    def _variable_step(self, t, fields, dt, pars, hook=null_hook):
        """Accept/reject loop (schemes.py:176-238)."""
        from scipy.interpolate import interp1d
        target = t + dt
        n_iter = 0
        try:
            fields.fill(self._interp_cache(target))
            return target, fields
        except (TypeError, ValueError):
            pass
        start = 1e-6 if self._internal_dt is None else self._internal_dt
        dt = self._internal_dt = min(start, dt) if self._recompute_target else start
        while True:
            err = None
            while err is None or err > self._tol:
                new_t, new_fields, err = self._fixed_step(t, fields, dt, pars, hook)
                dt = self._internal_dt = (self._safety_factor * dt
                                          * np.sqrt(self._tol / err))
            if new_t >= target:
                if self._recompute_target:
                    t, fields, err = self._fixed_step(t, fields, target - t, pars, hook)
                else:
                    self._interp_cache = interp1d(
                        [t, new_t], [fields.uflat[None], new_fields.uflat[None]], axis=0)
                    fields.fill(self._interp_cache(target))
                fields, pars = hook(t, fields, pars)
                return target, fields
            t, fields = new_t, new_fields.copy()
            n_iter += 1
            if n_iter > (self._max_iter if self._max_iter else n_iter + 1):
                raise RuntimeError("Rosebrock internal iteration "
                                   "above max iterations authorized")
            if dt < (self._dt_min if self._dt_min else dt * .5):
                raise RuntimeError("Rosebrock internal time step "
                                   "less than authorized")


def _row_subclass(name, fixed_only=False):
    from triflow_amd.tableaux import TABLEAUX
    tab = TABLEAUX[name]

    if fixed_only:
        def __init__(self, model):
            ROW_general.__init__(self, model, tab.alpha, tab.gamma, tab.b,
                                 time_stepping=False)
    else:
        def __init__(self, model, tol=1e-1, time_stepping=True, max_iter=None,
                     dt_min=None, recompute_target=True):
            ROW_general.__init__(self, model, tab.alpha, tab.gamma, tab.b,
                                 b_pred=tab.b_pred, time_stepping=time_stepping,
                                 tol=tol, max_iter=max_iter, dt_min=dt_min,
                                 recompute_target=recompute_target)
    return type(name, (ROW_general,), {"__init__": __init__})


ROS2 = _row_subclass("ROS2", fixed_only=True)      # schemes.py:241-256
ROS3PRw = _row_subclass("ROS3PRw")                 # schemes.py:259-300
ROS3PRL = _row_subclass("ROS3PRL")                 # schemes.py:303-353
RODASPR = _row_subclass("RODASPR")                 # schemes.py:356-427


class BDF2:
    """Linearly-implicit two-step BDF (NOT in the reference; named by
    BASELINE.json; parity unpinned by the reference).

    One Newton iteration from the extrapolated predictor ``U* = U_n``::

        (I - 2/3 dt J(U_n)) (U_{n+1} - U_n) = 1/3 (U_n - U_{n-1}) + 2/3 dt F(U_n)

    which follows from ``U_{n+1} = 4/3 U_n - 1/3 U_{n-1} + 2/3 dt F(U_{n+1})``
    with ``F(U_{n+1}) ~ F(U_n) + J(U_n)(U_{n+1} - U_n)``.  The first call (no
    history) is a backward-Euler step in the same linearised form, i.e.
    ``Theta(theta=1)``.  The history is dropped when ``dt`` changes.
    Follows the scheme protocol of schemes.py:523-559.
    """

    def __init__(self, model, solver=spsla.spsolve):
        self._model, self._solver = model, solver
        self._prev = None      # (U_{n-1}, dt)

    def __call__(self, t, fields, dt, pars, hook=null_hook):
        fields = fields.copy()
        fields, pars = hook(t, fields, pars)
        F = self._model.F(fields, pars)
        J = self._model.J(fields, pars)
        U = fields.uflat
        Id = sps.identity(U.size, format="csc")
        if self._prev is not None and np.isclose(self._prev[1], dt, rtol=1e-12, atol=0):
            A = Id - (2 / 3) * dt * J
            B = (1 / 3) * (U - self._prev[0]) + (2 / 3) * dt * F
        else:
            A = Id - dt * J
            B = dt * F
        self._prev = (U, dt)
        fields.fill(U + self._solver(A, B))
        fields, _ = hook(t + dt, fields, pars)
        return t + dt, fields
