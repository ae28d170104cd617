"""Load the reference's hot-path modules, unmodified, from /root/reference.

ORACLE TOOLING -- only usable in the build container (the reference does not
travel to the GPU box; nothing in ``-m gpu`` tests, ``smoke()`` or ``bench.py``
imports this).  Recipe of SURVEY.md §8(c): ``compilers.py``, ``routines.py``,
``model.py``, ``schemes.py`` and ``simulation.py`` are executed from where they
lie under synthetic parent packages, so the reference's ``__init__`` files
(which pull xarray / streamz / holoviews, absent here) never run.  In-memory
stand-ins are provided only for *absent third-party packages of the host
program* (the Dataset container, ``toolz.memoize``, ``pendulum``, ``streamz``),
never for the arithmetic.
"""

import importlib.util
import os
import sys
import types

REF = "/root/reference"


def available():
    return os.path.exists(os.path.join(REF, "triflow/core/compilers.py"))


def _module(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def _exec(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_loaded = None


def load():
    """Returns a namespace with ``Model, schemes, compilers, routines, Simulation``."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present")
    sys.dont_write_bytecode = True        # /root/reference is read-only territory
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from triflow_amd.fields import BaseFields      # xarray-free container stand-in

    def memoize(func):
        cache = {}

        def wrapper(*args):
            key = tuple(id(a) if not isinstance(a, (int, float, str)) else a for a in args)
            if key not in cache:
                cache[key] = func(*args)
            return cache[key]
        return wrapper

    pkg = _module("triflow")
    pkg.__path__ = []
    core = _module("triflow.core")
    core.__path__ = []
    plugins = _module("triflow.plugins")
    plugins.__path__ = []
    _module("triflow.core.fields", BaseFields=BaseFields)
    _module("toolz", memoize=memoize)

    class _Now:
        def subtract(self, **kw):
            return self

        def diff(self):
            return "n/a"

        def to_cookie_string(self):
            return "n/a"
    _module("pendulum", now=lambda: _Now())

    class _Stream:
        def emit(self, x):
            pass
    _module("streamz", Stream=_Stream)
    _module("triflow.plugins.container", TriflowContainer=object)

    theano_stub = None
    try:
        import theano  # noqa: F401
    except ImportError:
        # compilers.py imports theano lazily inside theano_compiler only
        theano_stub = True

    compilers = _exec("triflow.core.compilers", "triflow/core/compilers.py")
    routines = _exec("triflow.core.routines", "triflow/core/routines.py")
    model = _exec("triflow.core.model", "triflow/core/model.py")
    schemes = _exec("triflow.core.schemes", "triflow/core/schemes.py")
    core.schemes = schemes
    simulation = _exec("triflow.core.simulation", "triflow/core/simulation.py")

    _loaded = types.SimpleNamespace(
        Model=model.Model, schemes=schemes, compilers=compilers,
        routines=routines, Simulation=simulation.Simulation,
        model_module=model, theano_absent=theano_stub)
    return _loaded


def numpy_model(ref, *args, **kwargs):
    """A reference ``Model`` compiled with the reference's numpy compiler.

    SymPy >= 1.9 prints ``Heaviside(x, 1/2)``; the reference's one-argument
    ``np_Heaviside`` (compilers.py:204-205) then raises TypeError in J for
    state-dependent upwind velocities.  For those models only, the Jacobian
    lambda is rebuilt with the same module dictionary but a Heaviside that
    ignores the extra argument and is handed to the reference's own
    module-level ``compute_J_numpy`` (version hazards).
    """
    import numpy as np
    from functools import partial
    from sympy import Heaviside, lambdify
    model = ref.Model(*args, compiler=ref.compilers.numpy_compiler, **kwargs)
    if any(e.has(Heaviside) for e in model._J_sparse_array.tolist()):
        def np_Min(args):
            a, b = args
            return np.where(a < b, a, b)

        def np_Max(args):
            a, b = args
            return np.where(a < b, b, a)

        j_func = lambdify(model._symbolic_args, model._J_sparse_array.tolist(),
                          modules=[{"amax": np_Max, "amin": np_Min,
                                    "Heaviside": lambda a, *_: np.where(a < 0, 1, 1)},
                                   "numpy"])
        model.J._ufunc = partial(ref.compilers.compute_J_numpy, model, j_func)
    return model
