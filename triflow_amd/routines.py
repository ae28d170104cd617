"""``model.F`` / ``model.J``: the call boundary the compiler plugin sits behind.

Same adapter protocol as the reference's ``triflow/core/routines.py:8-91``:
the compiled callable is invoked as

    ``_ufunc(x, *dependent_variables, *help_functions, *parameters, periodic)``

with every physical parameter broadcast to one value per node (a scalar or a
per-node array is accepted, reference ``routines.py:40-43``) and ``periodic``
passed through untouched.  A missing parameter raises ``KeyError`` exactly as
there (``self.pars`` always ends with ``'periodic'``, ``routines.py:11``).
"""

import numpy as np
import sympy as sp


class ModelRoutine:
    def __init__(self, matrix, args, pars, ufunc, reduced=False):
        self.pars = list(pars) + ["periodic"]
        self.matrix = matrix
        self.args = args
        self._ufunc = ufunc

    def __repr__(self):
        return sp.Matrix(self.matrix.tolist()).__repr__()

    def _positional(self, fields, pars):
        x = np.asarray(fields["x"])
        uargs = [x, *[np.asarray(fields[key]) for key in self.args]]
        pargs = [pars[key] if key == "periodic" else pars[key] + x * 0
                 for key in self.pars]
        return uargs, pargs


class F_Routine(ModelRoutine):
    """Right-hand side ``F(U)``: flat float64 array, ``F[node * nvar + eq]``
    (reference ``routines.py:37-45``)."""

    def __call__(self, fields, pars):
        uargs, pargs = self._positional(fields, pars)
        return self._ufunc(*uargs, *pargs)

    def diff_approx(self, fields, pars, eps=1e-3):
        """Dense forward-difference Jacobian, one column per unknown
        (debug helper, reference ``routines.py:47-61``)."""
        U = fields.uflat
        J = np.zeros((U.size, U.size))
        F = self(fields, pars)
        for i in range(U.size):
            shifted = fields.copy()
            Up = shifted.uflat
            Up[i] += eps
            shifted.fill(Up)
            J[i] = (self(shifted, pars) - F) / eps
        return J.T


class J_Routine(ModelRoutine):
    """Jacobian ``dF/dU``: ``scipy.sparse.csc_matrix`` (or dense when
    ``sparse=False``), reference ``routines.py:82-91``."""

    def __call__(self, fields, pars, sparse=True):
        uargs, pargs = self._positional(fields, pars)
        J = self._ufunc(*uargs, *pargs)
        return J if sparse else J.todense()
