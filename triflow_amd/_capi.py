"""ctypes binding of ``libtriflow_hip.so`` (C ABI in ``include/triflow_hip.h``).

Thin by design: numpy arrays in, numpy arrays out, every non-zero return code
becomes a ``RuntimeError`` carrying ``tf_last_error()``.
"""

import ctypes as C
import os

import numpy as np

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


class ModelSpec(C.Structure):
    _fields_ = [("nvar", C.c_int32), ("nh", C.c_int32), ("npar", C.c_int32),
                ("mp", C.c_int32), ("nnz", C.c_int32), ("seg", C.c_int32),
                ("sweep_block", C.c_int32), ("uses_x", C.c_int32),
                ("parvec_mask", C.c_uint32)]


class SolverOpts(C.Structure):
    _fields_ = [("m1", C.c_int32), ("m_upper", C.c_int32), ("nstate", C.c_int32),
                ("refine", C.c_int32), ("device", C.c_int32), ("berr_every", C.c_int32),
                ("reserved", C.c_int32)]


class Scheme(C.Structure):
    _fields_ = [("kind", C.c_int32), ("stages", C.c_int32), ("theta", C.c_double),
                ("alpha", c_double_p), ("gamma", c_double_p), ("b", c_double_p),
                ("hook_after", C.c_int32), ("reserved", C.c_int32)]


#: name -> (restype, argtypes); every symbol declared in include/triflow_hip.h
SIGNATURES = {
    "tf_last_error": (C.c_char_p, []),
    "tf_runtime_info": (C.c_int, [c_int32_p, c_int32_p]),
    "tf_set_device": (C.c_int, [C.c_int32]),
    "tf_model_create": (C.c_int, [C.POINTER(ModelSpec), C.c_void_p, C.c_size_t,
                                  C.POINTER(C.c_void_p)]),
    "tf_model_add_alternate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64]),
    "tf_model_destroy": (None, [C.c_void_p]),
    "tf_solver_create": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                   C.POINTER(SolverOpts), C.POINTER(C.c_void_p)]),
    "tf_solver_destroy": (None, [C.c_void_p]),
    "tf_solver_describe": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p, C.c_int32, c_int64_p]),
    "tf_set_state": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_double_p]),
    "tf_get_state": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_double_p]),
    "tf_set_state_flat": (C.c_int, [C.c_void_p, C.c_int32, c_double_p]),
    "tf_get_state_flat": (C.c_int, [C.c_void_p, C.c_int32, c_double_p]),
    "tf_copy_state": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "tf_set_helpers": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_double_p]),
    "tf_set_param_scalar": (C.c_int, [C.c_void_p, C.c_int32, c_double_p]),
    "tf_set_param_vector": (C.c_int, [C.c_void_p, C.c_int32, c_double_p]),
    "tf_set_dx": (C.c_int, [C.c_void_p, c_double_p]),
    "tf_set_x": (C.c_int, [C.c_void_p, c_double_p]),
    "tf_set_dirichlet": (C.c_int, [C.c_void_p, C.c_int32, c_int32_p, c_int64_p, c_double_p]),
    "tf_set_dirichlet_values": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "tf_poke": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_int32_p, c_int64_p, c_double_p]),
    "tf_peek": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_int32_p, c_int64_p, c_double_p]),
    "tf_eval": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "tf_eval_repeat": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_double_p]),
    "tf_get_F": (C.c_int, [C.c_void_p, c_double_p]),
    "tf_get_J": (C.c_int, [C.c_void_p, c_double_p]),
    "tf_set_csc_map": (C.c_int, [C.c_void_p, c_int32_p, C.c_int64]),
    "tf_set_constant_jacobian": (C.c_int, [C.c_void_p, C.c_int32]),
    "tf_get_J_mapped": (C.c_int, [C.c_void_p, c_double_p]),
    "tf_factor": (C.c_int, [C.c_void_p, C.c_double]),
    "tf_solve": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "tf_matvec": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "tf_step_theta": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double]),
    "tf_step_row": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32,
                              c_double_p, c_double_p, c_double_p, c_double_p, C.c_int32,
                              c_double_p]),
    "tf_step_row_queued": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32,
                                     c_double_p, c_double_p, c_double_p, c_double_p, C.c_int32, C.c_int32]),
    "tf_read_err": (C.c_int, [C.c_void_p, C.c_int32, c_double_p]),
    "tf_step_bdf2": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double]),
    "tf_bdf2_reset": (C.c_int, [C.c_void_p]),
    "tf_step_bdf2_from": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double]),
    "tf_step_bdf2_owned": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int64,
                                     C.c_int32]),
    "tf_bdf2_release": (C.c_int, [C.c_void_p, C.c_int64]),
    "tf_diff_norm": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_double_p]),
    "tf_step_doubling": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                   C.c_int32, C.c_int32, C.POINTER(Scheme), C.c_int32, c_double_p]),
    "tf_backward_error": (C.c_int, [C.c_void_p, c_double_p, c_int32_p]),
    "tf_monitor_error": (C.c_int, [C.c_void_p, c_double_p]),
    "tf_solver_counters": (C.c_int, [C.c_void_p, c_int64_p, c_int64_p, c_int64_p]),
    "tf_sync": (C.c_int, [C.c_void_p]),
    "tf_timing_enable": (C.c_int, [C.c_void_p, C.c_int64]),
    "tf_timing_reset": (C.c_int, [C.c_void_p]),
    "tf_timing_get": (C.c_int, [C.c_void_p, C.c_int32, c_double_p, c_int64_p]),
    "tf_debug_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int32]),
    "tf_solver_kernel_block": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "tf_kernel_count": (C.c_int, []),
    "tf_kernel_name": (C.c_char_p, [C.c_int32]),
}


def _dptr(arr):
    return arr.ctypes.data_as(c_double_p)


def _f64(a, shape=None):
    out = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and out.shape != tuple(shape):
        raise ValueError("expected array of shape %r, got %r" % (tuple(shape), out.shape))
    return out


class Library:
    """A loaded ``libtriflow_hip.so`` with typed entry points."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise RuntimeError(
                "HIP runtime library %s is missing -- build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % path)
        self.path = path
        self.dll = C.CDLL(path)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(self.dll, name)
            fn.restype = restype
            fn.argtypes = argtypes

    def call(self, name, *args):
        rc = getattr(self.dll, name)(*args)
        if rc != 0:
            raise RuntimeError("%s: %s" % (name, self.dll.tf_last_error().decode()))

    def runtime_info(self):
        dev, cnt = C.c_int32(0), C.c_int32(0)
        self.call("tf_runtime_info", C.byref(dev), C.byref(cnt))
        return bool(dev.value), cnt.value

    def set_device(self, ordinal):
        self.call("tf_set_device", int(ordinal))

    def kernel_names(self):
        return [self.dll.tf_kernel_name(k).decode() for k in range(self.dll.tf_kernel_count())]


class DeviceModel:
    """``tf_model``: a loaded per-model code object."""

    def __init__(self, lib, spec, code, alt_code=None, alt_kernels=()):
        self.lib = lib
        self.spec = dict(spec)
        cs = ModelSpec(spec["nvar"], spec["nh"], spec["npar"], spec["mp"], spec["nnz"],
                       spec["seg"], spec["sweep_block"], spec["uses_x"], spec["parvec_mask"])
        self._code = C.create_string_buffer(code, len(code)) if code else None
        handle = C.c_void_p()
        lib.call("tf_model_create", C.byref(cs),
                 C.cast(self._code, C.c_void_p) if self._code is not None else None,
                 len(code) if code else 0, C.byref(handle))
        self.handle = handle
        self._alt = None
        if alt_code and alt_kernels:
            names = lib.kernel_names()
            mask = 0
            for k in alt_kernels:
                mask |= 1 << names.index(k)
            self._alt = C.create_string_buffer(alt_code, len(alt_code))
            lib.call("tf_model_add_alternate", handle, C.cast(self._alt, C.c_void_p), len(alt_code), mask)

    def close(self):
        if self.handle:
            self.lib.dll.tf_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceSolver:
    """``tf_solver``: resident state + kernels for ``nsys`` systems of ``N`` nodes."""

    def __init__(self, model, N, nsys=1, periodic=False, m1=0, m_upper=0, nstate=0,
                 refine=-1, device=-1, berr_every=0):
        self.model = model
        self.lib = model.lib
        self.N, self.nsys, self.periodic = int(N), int(nsys), bool(periodic)
        self.nvar, self.nh = model.spec["nvar"], model.spec["nh"]
        self.npar, self.nnz = model.spec["npar"], model.spec["nnz"]
        m1 = m1 or int(os.environ.get("TRIFLOW_M1", "0"))
        m_upper = m_upper or int(os.environ.get("TRIFLOW_M_UPPER", "0"))
        opts = SolverOpts(m1, m_upper, nstate, refine, device, berr_every, 0)
        handle = C.c_void_p()
        self.lib.call("tf_solver_create", model.handle, self.N, self.nsys,
                      int(self.periodic), C.byref(opts), C.byref(handle))
        self.handle = handle
        self.nstate = nstate if nstate > 0 else 3
        # constant-coefficient linear model: the factorisation of a step is kept while c and the
        # parameters are unchanged (tf_set_constant_jacobian)
        ju = model.spec.get("j_uniform") or []
        self.constant_jacobian = bool(ju) and all(ju) and os.environ.get("TRIFLOW_REUSE_FACTOR", "1") != "0"
        if self.constant_jacobian:
            self.lib.call("tf_set_constant_jacobian", self.handle, 1)

    def close(self):
        if self.handle:
            self.lib.dll.tf_solver_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def describe(self):
        nl, nbytes = C.c_int32(0), C.c_int64(0)
        chunks = (C.c_int32 * 32)()
        self.lib.call("tf_solver_describe", self.handle, C.byref(nl), chunks, 32, C.byref(nbytes))
        return dict(levels=nl.value, chunks=list(chunks[:nl.value]), device_bytes=nbytes.value)

    # ------------------------------------------------------------------ inputs
    def set_state(self, slot, arrays, first=0):
        """``arrays``: [nvars][nsys][N] (or [nvars][N] when nsys == 1)."""
        a = _f64(arrays).reshape(-1, self.nsys, self.N)
        self.lib.call("tf_set_state", self.handle, slot, first, a.shape[0], _dptr(a))

    def get_state(self, slot, first=0, nvars=None):
        nvars = self.nvar - first if nvars is None else nvars
        out = np.empty((nvars, self.nsys, self.N))
        self.lib.call("tf_get_state", self.handle, slot, first, nvars, _dptr(out))
        return out

    def set_state_flat(self, slot, uflat):
        a = _f64(uflat).reshape(self.nsys, self.N * self.nvar)
        self.lib.call("tf_set_state_flat", self.handle, slot, _dptr(a))

    def get_state_flat(self, slot):
        out = np.empty((self.nsys, self.N * self.nvar))
        self.lib.call("tf_get_state_flat", self.handle, slot, _dptr(out))
        return out

    def copy_state(self, src, dst):
        self.lib.call("tf_copy_state", self.handle, src, dst)

    def set_helpers(self, arrays, first=0):
        a = _f64(arrays).reshape(-1, self.nsys, self.N)
        self.lib.call("tf_set_helpers", self.handle, first, a.shape[0], _dptr(a))

    def set_param(self, k, value):
        """Scalar per system, or [nsys][N] per-node values when the model was
        compiled with parameter ``k`` as a vector."""
        if (self.model.spec["parvec_mask"] >> k) & 1:
            a = _f64(np.broadcast_to(np.asarray(value, dtype=float), (self.nsys, self.N)))
            self.lib.call("tf_set_param_vector", self.handle, k, _dptr(a))
        else:
            a = _f64(np.broadcast_to(np.asarray(value, dtype=float), (self.nsys,)))
            self.lib.call("tf_set_param_scalar", self.handle, k, _dptr(a))

    def set_dx(self, dx):
        a = _f64(np.broadcast_to(np.asarray(dx, dtype=float), (self.nsys,)))
        self.lib.call("tf_set_dx", self.handle, _dptr(a))

    def set_x(self, x):
        a = _f64(np.broadcast_to(np.asarray(x, dtype=float), (self.nsys, self.N)))
        self.lib.call("tf_set_x", self.handle, _dptr(a))

    def set_dirichlet(self, entries):
        """``entries``: iterable of (variable index, node index, value)."""
        entries = list(entries)
        n = len(entries)
        var = np.array([e[0] for e in entries], dtype=np.int32)
        node = np.array([e[1] for e in entries], dtype=np.int64)
        val = np.array([e[2] for e in entries], dtype=np.float64)
        self.lib.call("tf_set_dirichlet", self.handle, n,
                      var.ctypes.data_as(c_int32_p), node.ctypes.data_as(c_int64_p), _dptr(val))

    def poke(self, slot, entries):
        """Point writes ``(variable index, node index, value)`` into a resident slot."""
        entries = list(entries)
        if not entries:
            return
        var = np.array([e[0] for e in entries], dtype=np.int32)
        node = np.array([e[1] for e in entries], dtype=np.int64)
        val = np.array([e[2] for e in entries], dtype=np.float64)
        self.lib.call("tf_poke", self.handle, slot, len(entries),
                      var.ctypes.data_as(c_int32_p), node.ctypes.data_as(c_int64_p), _dptr(val))

    def peek(self, slot, var, node):
        """Value of one node of a resident slot (per system)."""
        v = np.array([var], dtype=np.int32)
        n = np.array([node], dtype=np.int64)
        out = np.empty((1, self.nsys))
        self.lib.call("tf_peek", self.handle, slot, 1, v.ctypes.data_as(c_int32_p),
                      n.ctypes.data_as(c_int64_p), _dptr(out))
        return out[0]

    def set_dirichlet_values(self, before=None, after=None):
        b = _f64(before) if before is not None else None
        a = _f64(after) if after is not None else None
        self.lib.call("tf_set_dirichlet_values", self.handle,
                      _dptr(b) if b is not None else None, _dptr(a) if a is not None else None)

    # ----------------------------------------------------------------- seam #1
    def eval(self, slot=0, with_j=False):
        self.lib.call("tf_eval", self.handle, slot, int(with_j))

    def eval_repeat(self, slot=0, with_j=True, reps=20):
        """Mean duration (ms) of ``reps`` back-to-back sweeps (two HIP events)."""
        ms = C.c_double(0.0)
        self.lib.call("tf_eval_repeat", self.handle, slot, int(with_j), reps, C.byref(ms))
        return ms.value / reps

    def get_F(self):
        out = np.empty((self.nsys, self.N * self.nvar))
        self.lib.call("tf_get_F", self.handle, _dptr(out))
        return out

    def get_J(self):
        out = np.empty((self.nsys, self.N, max(self.nnz, 1)))
        self.lib.call("tf_get_J", self.handle, _dptr(out))
        return out[:, :, :self.nnz]

    def set_csc_map(self, index_list):
        """Value-table index (node * nnz + k) of every entry ``get_J_mapped`` is to return."""
        m = np.ascontiguousarray(index_list, dtype=np.int32)
        self.lib.call("tf_set_csc_map", self.handle, m.ctypes.data_as(c_int32_p), m.size)
        self._csc_n = m.size

    def get_J_mapped(self):
        out = np.empty(self._csc_n)
        self.lib.call("tf_get_J_mapped", self.handle, _dptr(out))
        return out

    # ----------------------------------------------------------------- seam #3
    def factor(self, c):
        self.lib.call("tf_factor", self.handle, float(c))

    def solve(self, rhs_flat):
        rhs = _f64(rhs_flat).reshape(self.nsys, self.N * self.nvar)
        out = np.empty_like(rhs)
        self.lib.call("tf_solve", self.handle, _dptr(rhs), _dptr(out))
        return out

    def matvec(self, v_flat):
        v = _f64(v_flat).reshape(self.nsys, self.N * self.nvar)
        out = np.empty_like(v)
        self.lib.call("tf_matvec", self.handle, _dptr(v), _dptr(out))
        return out

    # ----------------------------------------------------------------- seam #2
    def step_theta(self, src, dst, dt, theta=1.0):
        self.lib.call("tf_step_theta", self.handle, src, dst, float(dt), float(theta))

    def step_row(self, src, dst, dt, alpha, gamma, b, b_pred=None, hook_after=False,
                 want_err=True):
        alpha, gamma = _f64(alpha), _f64(gamma)
        b = _f64(b)
        s = b.size
        bp = _f64(b_pred) if b_pred is not None else None
        err = C.c_double(0.0)
        self.lib.call("tf_step_row", self.handle, src, dst, float(dt), s, _dptr(alpha),
                      _dptr(gamma), _dptr(b), _dptr(bp) if bp is not None else None,
                      int(hook_after), C.byref(err) if want_err else None)
        return np.float64(err.value) if (want_err and bp is not None) else None

    def step_row_queued(self, src, dst, dt, alpha, gamma, b, b_pred, hook_after=False, err_slot=1):
        """The same step; its embedded error estimate stays on the device (reduction slot ``err_slot``)
        until :meth:`read_err` fetches it."""
        alpha, gamma, b, bp = _f64(alpha), _f64(gamma), _f64(b), _f64(b_pred)
        self.lib.call("tf_step_row_queued", self.handle, src, dst, float(dt), b.size, _dptr(alpha),
                      _dptr(gamma), _dptr(b), _dptr(bp), int(hook_after), int(err_slot))

    def read_err(self, err_slot):
        err = C.c_double(0.0)
        self.lib.call("tf_read_err", self.handle, int(err_slot), C.byref(err))
        return np.float64(err.value)

    def step_bdf2(self, src, dst, dt, owner=0, continuing=True):
        """``owner`` 0: the solver's own history (a caller that owns the solver); otherwise the
        history buffer of one scheme instance (``tf_step_bdf2_owned``)."""
        if owner:
            self.lib.call("tf_step_bdf2_owned", self.handle, src, dst, float(dt), int(owner),
                          int(bool(continuing)))
        else:
            self.lib.call("tf_step_bdf2", self.handle, src, dst, float(dt))

    def step_bdf2_from(self, src, dst, prev, dt):
        """BDF-2 step whose history U_{n-1} is in state slot ``prev`` (-1: none, backward-Euler form)."""
        self.lib.call("tf_step_bdf2_from", self.handle, src, dst, int(prev), float(dt))

    def bdf2_release(self, owner):
        if self.handle:
            self.lib.call("tf_bdf2_release", self.handle, int(owner))

    def bdf2_reset(self):
        self.lib.call("tf_bdf2_reset", self.handle)

    def step_doubling(self, src, dst, tmp, coarse, dt, m, desc, ord=2, nfine=10):
        """One step-doubling trial (``tf_step_doubling``); ``desc``: dict(kind="theta", theta=..)
        or dict(kind="row", alpha=.., gamma=.., b=.., hook_after=..).  Returns err per system."""
        keep = []
        if desc["kind"] == "theta":
            sch = Scheme(0, 0, float(desc["theta"]), None, None, None, 0, 0)
        else:
            arrs = [_f64(desc[k]) for k in ("alpha", "gamma", "b")]
            keep.extend(arrs)
            sch = Scheme(1, arrs[2].size, 0.0, _dptr(arrs[0]), _dptr(arrs[1]), _dptr(arrs[2]),
                         int(bool(desc.get("hook_after", True))), 0)
        err = np.empty(self.nsys)
        self.lib.call("tf_step_doubling", self.handle, src, dst, tmp, coarse, float(dt), int(m),
                      int(nfine), C.byref(sch), 0 if ord in (0, np.inf, "inf") else int(ord), _dptr(err))
        return err

    def diff_norms(self, slot_a, slot_b, ord=2):
        """||state[a] - state[b]|| per system and dependent variable, [nsys][nvar]."""
        out = np.empty((self.nsys, self.nvar))
        self.lib.call("tf_diff_norm", self.handle, slot_a, slot_b,
                      0 if ord in (0, np.inf, "inf") else int(ord), _dptr(out))
        return out

    def backward_error(self):
        """(componentwise backward error of the checked solve, refinement active?)"""
        om, flag = C.c_double(0.0), C.c_int32(0)
        self.lib.call("tf_backward_error", self.handle, C.byref(om), C.byref(flag))
        return om.value, bool(flag.value)

    def counters(self):
        """dict(factorisations, checks, replans) since the solver was created."""
        f, c, r = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self.lib.call("tf_solver_counters", self.handle, C.byref(f), C.byref(c), C.byref(r))
        return dict(factorisations=f.value, checks=c.value, replans=r.value)

    def kernel_block(self, name):
        """Workgroup size the kernel ``name`` of the model's code object was built for."""
        b = C.c_int32(0)
        self.lib.call("tf_solver_kernel_block", self.handle, self.lib.kernel_names().index(name), C.byref(b))
        return b.value

    def monitor_error(self):
        """Worst backward error seen by the in-pass monitor of the Rosenbrock steps since the last
        synchronising call."""
        w = C.c_double(0.0)
        self.lib.call("tf_monitor_error", self.handle, C.byref(w))
        return w.value

    def sync(self):
        self.lib.call("tf_sync", self.handle)
        if not getattr(self, "_warned_replan", False) and self.counters()["replans"]:
            import warnings
            self._warned_replan = True
            warnings.warn("banded solver: the default partition lost accuracy on this matrix and the "
                          "factorisations go through the rescue plan (8 x longer chunks, every solve "
                          "checked): several times slower; see DESIGN.md section 4.5", RuntimeWarning,
                          stacklevel=2)

    def debug_stamps(self, levels=8):
        """[levels][64] uint64 stamps of a -DTF_STAMPS kernel build (first call: switch on)."""
        out = np.zeros((levels, 64), dtype=np.uint64)
        self.lib.call("tf_debug_stamps", self.handle, out.ctypes.data_as(C.POINTER(C.c_uint64)), levels)
        return out

    # ------------------------------------------------------------- measurement
    def timing(self, on=True, kernels=None):
        """Time every launch (``on=True``), none, or only the named ``kernels``."""
        if kernels is not None:
            names = self.lib.kernel_names()
            mask = 0
            for k in kernels:
                mask |= 1 << names.index(k)
        else:
            mask = -1 if on else 0
        self.lib.call("tf_timing_enable", self.handle, mask)

    def timing_reset(self):
        self.lib.call("tf_timing_reset", self.handle)

    def timing_report(self):
        """{kernel name: (total ms, launches)} for kernels launched since reset."""
        out = {}
        for k, name in enumerate(self.lib.kernel_names()):
            ms, n = C.c_double(0.0), C.c_int64(0)
            self.lib.call("tf_timing_get", self.handle, k, C.byref(ms), C.byref(n))
            if n.value:
                out[name] = (ms.value, n.value)
        return out
