"""Symbolic front-end: PDE strings -> per-node stencil IR.

This is the caller side of the hot path.  It
re-derives, with SymPy, the same intermediate representation the reference
builds in ``triflow/core/model.py:138-297`` so that the HIP compiler plugin and
the CPU oracle see identical symbolic input:

``_args``            ordered argument names  (reference ``model.py:317-328``)
``F_array``          nvar expanded stencil expressions (``model.py:265``)
``J_array``          nvar*nvar*W symbolic Jacobian, Fortran-flattened
                     (``model.py:271-285``)
``_sparse_indices``  positions of the structurally non-null Jacobian entries
                     (``model.py:288-291``)
``_bounds`` / ``_window_range`` / ``_nvar``  stencil geometry
                     (``model.py:244-247, 380-386``)

The discretisation rules are the reference's: centred differences of order
1-4 (``model.py:401-439``) and ``upwind(a, U, acc)`` with one-sided
differences of accuracy 1-3 (``model.py:441-478``).
"""

import logging
import pickle
from functools import partial

import numpy as np
import sympy as sp

from .fields import BaseFields
from .routines import F_Routine, J_Routine

log = logging.getLogger(__name__)
log.addHandler(logging.NullHandler())

#: forward-difference step of the ``fdiff_jac=True`` Jacobian (reference model.py:22)
EPS = 1e-6

_MAX_NAMESPACE_ORDER = 9  # reference generates dx..dxxxxxxxxx (model.py:58)

# centred finite-difference weights {derivative order: {node offset: weight}},
# the denominators are dx**order (reference model.py:405-436)
_HALF = 1 / 2
_CENTRED = {
    1: {-1: -_HALF, 1: _HALF},
    2: {-1: 1, 0: -2, 1: 1},
    3: {-2: -_HALF, -1: 1, 1: -1, 2: _HALF},
    4: {-2: 1, -1: -4, 0: 6, 1: -4, 2: 1},
}


def _as_tuple(arg):
    """str-or-iterable coercion of the constructor inputs (model.py:163-169)."""
    if arg is None:
        return ()
    if isinstance(arg, str):
        return (arg,)
    return tuple(arg)


def _node_symbol(name, offset):
    """Discrete unknown ``U_m2, U_m1, U, U_p1, U_p2`` (model.py:388-399)."""
    if offset == 0:
        return sp.Symbol(name)
    return sp.Symbol("%s_%s%i" % (name, "m" if offset < 0 else "p", abs(offset)))


def _rebuild_model(eqs, dep, pars, helps, bdcs, compiler):
    return Model(eqs, dep, pars, helps, bdcs, compiler=compiler)


class Model:
    """Finite-difference model of ``dU/dt = F(U)`` on a uniform 1-D grid.

    Same constructor as the reference (``model.py:138-150``).  ``compiler``
    selects the plugin that turns the symbolic ``F``/``J`` into numeric
    callables (seam #1, ``model.py:152-155, 299-311``): this framework has one
    backend, the MI355X HIP compiler, which is also what the reference's
    string spellings ``"theano"`` and ``"numpy"`` resolve to here; any callable
    ``compiler(model) -> (F_function, J_function)`` is used as is (that is how
    the tests inject the CPU oracle).
    """

    def __init__(self, differential_equations, dependent_variables,
                 parameters=None, help_functions=None, bdc_conditions=None,
                 compiler="hip", simplify=False, fdiff_jac=False, double=True,
                 hold_compilation=False):
        self._double = double
        self._compiler_spec = compiler
        self._diff_eqs = _as_tuple(differential_equations)
        self._indep_vars = ("x",)
        self._dep_vars = _as_tuple(dependent_variables)
        self._pars = _as_tuple(parameters)
        self._help_funcs = _as_tuple(help_functions)
        self._bdcs = _as_tuple(bdc_conditions)
        self._nvar = len(self._dep_vars)
        self._symb_t = sp.Symbol("t")

        self._parse()
        # stencil footprint per field name: {(symbol, offset)}, grown while the
        # derivatives are replaced (reference model.py:229-232)
        self._symb_vars_with_spatial_diff_order = {
            name: {(sp.Function(name), 0)}
            for name in self._dep_vars + self._help_funcs}
        discrete_eqs = self._discretise(self._symb_diff_eqs)
        self._dbdcs = self._discretise(self._symb_bdcs)

        lo, hi = 0, 0
        for name in self._dep_vars:   # help functions do not widen the window (model.py:244)
            offs = [o for _, o in self._symb_vars_with_spatial_diff_order[name]]
            lo, hi = min(lo, min(offs)), max(hi, max(offs))
        self._bounds = (lo, hi)
        self._window_range = hi - lo + 1

        def grid(names):
            # Fortran order: offset-major, field-minor (model.py:252-262)
            return np.array([_node_symbol(n, o)
                             for o in range(lo, hi + 1) for n in names], dtype=object)

        unknowns = grid(self._dep_vars)
        self._discrete_variables = grid(self._dep_vars + self._help_funcs)

        self.F_array = np.array(discrete_eqs)
        if simplify:
            self.F_array = np.array([e.simplify() for e in self.F_array.tolist()])
        if fdiff_jac:
            rows = [[(e.subs(u, u + EPS) - e) / EPS for u in unknowns]
                    for e in discrete_eqs]
        else:
            rows = [[e.diff(u) for u in unknowns] for e in discrete_eqs]
        self.J_array = np.array(rows).flatten("F")
        if simplify:
            self.J_array = np.array([e.expand().simplify()
                                     for e in self.J_array.tolist()])
        self._sparse_indices = np.where(self.J_array != 0)
        self._J_sparse_array = self.J_array[self._sparse_indices]

        if not hold_compilation:
            self.compile(compiler)

    # ------------------------------------------------------------------ parsing
    def _parse(self):
        """strings -> SymPy (reference model.py:25-74, 480-542)."""
        x = sp.Symbol("x")
        fields = self._dep_vars + self._help_funcs
        namespace = {"x": x}
        for order in range(1, _MAX_NAMESPACE_ORDER + 1):
            namespace["d" + "x" * order] = partial(
                lambda n, expr: sp.Derivative(expr, x, n), order)
            for name in fields:
                namespace["d%s%s" % ("x" * order, name)] = sp.Derivative(
                    sp.Function(name)(x), x, order)

        self._symb_indep_vars = (x,)
        self._symb_dep_vars = tuple(sp.Function(n)(x) for n in self._dep_vars)
        self._symb_help_funcs = tuple(sp.Function(n)(x) for n in self._help_funcs)
        self._symb_pars = sp.symbols(self._pars)
        # only the dependent variables are promoted to functions of x before
        # ``doit`` (the reference zips dep-var symbols only, model.py:515-521)
        promote = dict(zip(map(sp.Symbol, self._dep_vars),
                           self._symb_dep_vars + self._symb_help_funcs))

        def parse(equations):
            try:
                return tuple(sp.sympify(eq, locals=namespace).xreplace(promote).doit()
                             for eq in equations)
            except (TypeError, sp.SympifyError):
                raise ValueError("badly formated differential equations")

        self._symb_diff_eqs = parse(self._diff_eqs)
        self._symb_bdcs = parse(self._bdcs)

    # ----------------------------------------------------------- discretisation
    def _touch(self, name, offsets):
        for off in offsets:
            if off:
                self._symb_vars_with_spatial_diff_order[name].add(
                    (_node_symbol(name, off), off))

    def _finite_diff_scheme(self, U, order):
        """Centred stencil of a derivative (reference model.py:401-439)."""
        weights = _CENTRED.get(int(order))
        if weights is None:
            raise NotImplementedError(
                "Finite difference up to 5th order not implemented yet")
        name = str(U)
        self._touch(name, weights)
        num = sum(w * _node_symbol(name, off) for off, w in weights.items())
        return num / sp.Symbol("dx") ** order

    def _upwind_scheme(self, a, U, accuracy):
        """``Max(a,0)*D-(U) + Min(a,0)*D+(U)`` (reference model.py:441-478)."""
        dx = sp.Symbol("dx")
        name = str(U)
        u = partial(_node_symbol, name)
        if accuracy == 1:
            self._touch(name, (-1, 1))
            back = (u(0) - u(-1)) / dx
            fwd = (u(1) - u(0)) / dx
        elif accuracy == 2:
            self._touch(name, (-2, -1, 1, 2))
            back = (3 * u(0) - 4 * u(-1) + u(-2)) / (2 * dx)
            fwd = (-3 * u(0) + 4 * u(1) - u(2)) / (2 * dx)
        elif accuracy == 3:
            self._touch(name, (-2, -1, 1, 2))
            back = (2 * u(1) + 3 * u(0) - 6 * u(-1) + u(-2)) / (6 * dx)
            fwd = (-2 * u(-1) - 3 * u(0) + 6 * u(1) - u(2)) / (6 * dx)
        else:
            raise NotImplementedError("Upwind up to 2nd order not implemented yet")
        return sp.Max(a, 0) * back + sp.Min(a, 0) * fwd

    def _discretise(self, equations):
        """Derivative -> stencil, functions -> node symbols, upwind, expand
        (reference model.py:544-577)."""
        x = self._symb_indep_vars[0]
        to_symbol = [(f, sp.Symbol(str(f.func)))
                     for f in self._symb_dep_vars + self._symb_help_funcs]
        out = []
        for eq in equations:
            expr = eq
            for deriv in eq.find(sp.Derivative):
                var = sp.Symbol(str(deriv.args[0].func))
                order = 0
                for wrt in deriv.args[1:]:
                    sym, count = (wrt, 1) if isinstance(wrt, sp.Symbol) else wrt
                    if sym == x:
                        order = count
                expr = expr.replace(deriv, self._finite_diff_scheme(var, order))
            expr = expr.subs(to_symbol)
            expr = expr.replace(sp.Function("upwind"), self._upwind_scheme)
            out.append(expr.expand())
        return tuple(out)

    # ---------------------------------------------------------------- compiling
    def compile(self, compiler="hip"):
        """Plugin seam #1 (reference model.py:299-311)."""
        if isinstance(compiler, str):
            from .compilers import resolve_compiler
            compiler = resolve_compiler(compiler)
        F_function, J_function = compiler(self)
        fields = self._dep_vars + self._help_funcs
        self.F = F_Routine(self.F_array, fields, self._pars, F_function)
        self.J = J_Routine(self._J_sparse_array, fields, self._pars, J_function)
        # device-resident fast path used by this framework's own schemes
        self._device = getattr(F_function, "device_model", None)

    @property
    def fields_template(self):
        return BaseFields.factory1D(self._dep_vars, self._help_funcs)

    @property
    def _symbolic_args(self):
        return [*self._symb_indep_vars, *self._discrete_variables,
                *self._symb_pars, sp.Symbol("dx")]

    @property
    def _args(self):
        return [str(a) for a in self._symbolic_args]

    # -------------------------------------------------------------- persistence
    def save(self, filename):
        """Pickle the model (reference model.py:330-344)."""
        with open(filename, "wb") as f:
            pickle.dump(self, f)

    @staticmethod
    def load(filename):
        """Reference model.py:361-378."""
        with open(filename, "rb") as f:
            return pickle.load(f)

    def __reduce__(self):
        # The reference rebuilds from the constructor strings and silently
        # falls back to its default compiler (model.py:77-80, 579-583); the
        # compiler choice is kept here.
        return (_rebuild_model, (self._diff_eqs, self._dep_vars, self._pars,
                                 self._help_funcs, self._bdcs, self._compiler_spec))

    def __repr__(self):
        return ("{equations}\n\nVariables\n---------\n"
                "unknowns:       {vars}\nhelpers:        {helps}\n"
                "parameters:     {pars}").format(
            equations="\n".join(self._diff_eqs),
            vars=", ".join(self._dep_vars),
            helps=", ".join(self._help_funcs) if self._pars else None,
            pars=", ".join(self._pars) if self._pars else None)
