"""The HIP "compiler" plugin (seam #1 of the reference).

``hip_compiler(model) -> (F_function, J_function)`` plays the role of the
reference's ``numpy_compiler`` / ``theano_compiler``
(``triflow/core/compilers.py:11, 181``): it lowers the model's symbolic stencil
to a gfx950 code object (generated per-node bodies inside the hand-written
kernel skeleton of ``csrc/tf_kernels.h``, built with ``hipcc --genco`` and
cached by source hash) and returns callables obeying the positional
``_ufunc(x, *dep_vars, *help_funcs, *pars, periodic)`` protocol of
``triflow/core/routines.py:37-45, 82-91``: ``F`` comes back as a flat float64
array in node-major order, ``J`` as a ``scipy.sparse.csc_matrix`` with the
reference's (row, col) pattern (``compilers.py:303-331``).

The two callables are the *drop-in* path (every call moves its inputs and
outputs over PCIe).  The schemes of this package do not use them: they drive
the resident state of the same :class:`CompiledModel` through ``device.py``.

There is no CPU implementation behind this module: without ``hipcc`` or the
built ``libtriflow_hip.so`` it raises.
"""

import fcntl
import json
import logging
import os
import re
import shutil
import subprocess
import threading
import warnings

import numpy as np
import scipy.sparse as sps

from . import codegen
from ._capi import DeviceModel, DeviceSolver, Library

log = logging.getLogger(__name__)
log.addHandler(logging.NullHandler())

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "lib", "libtriflow_hip.so")
CACHE_DIR = os.path.join(PKG_DIR, "_cache")
GPU_ARCH = "gfx950"
# TRIFLOW_HIPCC_EXTRA: e.g. "-DTF_USE_JUNIFORM=0 -DTF_BACKSUB_DEPTH=1" (A/B runs of kernel variants)
HIPCC_FLAGS = [*os.environ.get("TRIFLOW_HIPCC_OPT", "-O3").split(), "-std=c++17",
               "-ffp-contract=off", "--offload-arch=" + GPU_ARCH,
               *os.environ.get("TRIFLOW_HIPCC_EXTRA", "").split()]

#: translation units of the host runtime (see the header of csrc/tf_solver.h)
RUNTIME_SOURCES = ("tf_rt_plan.cpp", "tf_rt_io.cpp", "tf_rt_steps.cpp", "tf_rt_diag.cpp",
                   "tf_solver_sweeps.cpp", "tf_solver_linear.cpp")
_SKELETON = ("tf_args.h", "tf_math.h", "tf_kernels.h", "tf_crs.h", "tf_coop_hip.h", "tf_cr2_hip.h", "tf_cr3_hip.h", "tf_entry_hip.h")
_TU_HEAD = ('#include <hip/hip_runtime.h>\n'
            '#define TF_DEVICE __device__ __forceinline__\n'
            '%s'
            '#include "tf_math.h"\n')

_TU_TAIL = '#include "tf_kernels.h"\n#include "tf_entry_hip.h"\n'


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP compiler plugin needs the ROCm toolchain")
    return exe


def _skeleton_stamp():
    parts = []
    for name in _SKELETON:
        with open(os.path.join(CSRC, name), "rb") as f:
            parts.append(f.read())
    return codegen.source_hash(*parts)


def build_runtime_library(force=False):
    """Compile the host runtime (``csrc/tf_rt_*.cpp``, ``tf_solver_*.cpp``: ``RUNTIME_SOURCES``) +
    ``tf_backend_hip.cpp`` into ``lib/libtriflow_hip.so`` (host code only; links libamdhip64)."""
    srcs = [os.path.join(CSRC, n) for n in RUNTIME_SOURCES] + [os.path.join(CSRC, "tf_backend_hip.cpp")]
    deps = srcs + [os.path.join(CSRC, n) for n in ("tf_args.h", "tf_backend.h", "tf_solver.h")] + \
        [os.path.join(os.path.dirname(PKG_DIR), "include", "triflow_hip.h")]
    if not force and os.path.exists(LIB_PATH) and \
            all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    cmd = [_hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=" + GPU_ARCH,
           "-I", CSRC, *srcs, "-lpthread", "-o", LIB_PATH + ".tmp"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libtriflow_hip.so failed:\n" + res.stderr[-4000:])
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


def _parse_resource_usage(stderr):
    """{kernel: {VGPRs, SGPRs, ScratchSize, Occupancy}} from -Rpass-analysis output."""
    usage, name = {}, None
    for line in stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        for key, pat in (("VGPRs", r" VGPRs: (\d+)"), ("SGPRs", r" SGPRs: (\d+)"),
                         ("ScratchSize", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("Occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and name:
                usage[name][key] = int(m.group(1))
    return usage


def resource_usage(hsaco_path):
    """Resource table written next to a cached code object."""
    with open(hsaco_path[:-len(".hsaco")] + ".json") as f:
        return json.load(f)["kernels"]


def _compile_code_object(model, source, tag, hsaco):
    """hipcc on the generated translation unit -> ``hsaco`` (+ source and resource table next to it)."""
    global BUILD_COUNT
    BUILD_COUNT += 1
    hip = os.path.join(CACHE_DIR, "model_%s.hip" % tag)
    # ranks of one node may compile the same uncached model at the same time: every
    # file of the cache appears by rename, never half written
    hip_tmp = hip + ".%d.tmp" % os.getpid()
    with open(hip_tmp, "w") as f:
        f.write(source)
    os.replace(hip_tmp, hip)
    tmp = hsaco + ".%d.tmp" % os.getpid()
    log.info("hipcc: compiling stencil + solver kernels for %s", model._diff_eqs)

    def compile_with(flags, out):
        cmd = [_hipcc(), *flags, "-I", CSRC, "--genco", "--no-gpu-bundle-output",
               "-Rpass-analysis=kernel-resource-usage", "-o", out, hip]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed on the generated kernels (%s):\n%s"
                               % (hip, res.stderr[-4000:]))
        return _parse_resource_usage(res.stderr)

    flags = list(HIPCC_FLAGS)
    usage = compile_with(flags, tmp)
    spilled = sorted(k for k, u in usage.items() if u.get("ScratchSize", 0) > 0)
    alt_flags, alt_usage = None, None
    alt = hsaco[:-len(".hsaco")] + ".alt.hsaco"
    if spilled and "-O1" not in flags and "-O0" not in flags \
            and os.environ.get("TRIFLOW_ALLOW_SCRATCH") != "1":          # (A/B runs)
        # Round 1 saw wrong solves from solver kernels that spill to scratch at -O2/-O3 (wide
        # blocks: error 0.3 at -O3, 6e-15 at -O1); round 2 could not reproduce it (DESIGN.md
        # "compiler notes"), the conservative gate stays -- per kernel since round 3: the kernels
        # that spill are taken from a second build of the same source at -O1 (they spill there
        # too, correctly), every other kernel keeps its -O3 build.  TRIFLOW_SPILL_GATE=object
        # restores the round-2 behaviour (the whole code object at -O1).
        _warn_wide(model, spilled)
        alt_flags = ["-O1"] + [f for f in flags if not f.startswith("-O")]
        if os.environ.get("TRIFLOW_SPILL_GATE") == "object":
            flags, alt_flags = alt_flags, None
            usage = compile_with(flags, tmp)
        else:
            alt_usage = compile_with(alt_flags, alt + ".%d.tmp" % os.getpid())
            os.replace(alt + ".%d.tmp" % os.getpid(), alt)
    meta = os.path.join(CACHE_DIR, "model_%s.json" % tag)
    with open(meta + ".%d.tmp" % os.getpid(), "w") as f:
        json.dump(dict(equations=list(model._diff_eqs), flags=flags, kernels=usage,
                       alt_flags=alt_flags, alt_kernels=spilled if alt_flags else [],
                       alt_usage={k: alt_usage[k] for k in spilled} if alt_usage else {},
                       skeleton=_skeleton_stamp(), hipcc=hipcc_version()), f, indent=1)
    os.replace(meta + ".%d.tmp" % os.getpid(), meta)
    os.replace(tmp, hsaco)
    evict_stale_cache()


_warned_wide = set()


def _warn_wide(model, kernels):
    """Once per model and process: its wide blocks do not fit the registers of a wavefront."""
    key = tuple(str(e) for e in model._diff_eqs)
    if key in _warned_wide:
        return
    _warned_wide.add(key)
    warnings.warn("triflow_amd: %d kernel(s) of the model %s need more registers than a wavefront has "
                  "(%s spill to scratch): those kernels are built at -O1 and run slower than the "
                  "narrow-model path (DESIGN.md, solver limits)"
                  % (len(kernels), list(key), ", ".join(kernels)), RuntimeWarning, stacklevel=4)


def alternate_of(hsaco_path):
    """(path of the second build, kernels taken from it) of a cached code object, or (None, [])."""
    try:
        with open(hsaco_path[:-len(".hsaco")] + ".json") as f:
            meta = json.load(f)
    except (OSError, ValueError):
        return None, []
    alt = hsaco_path[:-len(".hsaco")] + ".alt.hsaco"
    kernels = meta.get("alt_kernels") or []
    return (alt, kernels) if kernels and os.path.exists(alt) else (None, [])


def build_code_object(model, parvec_mask=0, seg=None, sweep_block=None):
    """Model -> (path of the cached gfx950 code object, spec dict)."""
    seg = seg or int(os.environ.get("TRIFLOW_SWEEP_SEG", "8"))
    sweep_block = sweep_block or int(os.environ.get("TRIFLOW_SWEEP_BLOCK", "256"))
    body, spec = codegen.lower_model(model, parvec_mask=parvec_mask, seg=seg,
                                     sweep_block=sweep_block)
    source = _TU_HEAD % "" + body + _TU_TAIL
    tag = codegen.source_hash(source, _skeleton_stamp(), " ".join(HIPCC_FLAGS), hipcc_version(), os.environ.get("TRIFLOW_SPILL_GATE", "kernel"), "elf")
    os.makedirs(CACHE_DIR, exist_ok=True)
    hsaco = os.path.join(CACHE_DIR, "model_%s.hsaco" % tag)
    if not os.path.exists(hsaco):
        # The ranks of one node may ask for the same uncached model at the same time (8 ranks of a
        # sweep on a cold cache): one of them compiles, the others wait on the lock and find the
        # code object.  Every file of the cache still appears by rename, never half written, so a
        # file system without working locks costs duplicate compilations, not a torn file.
        lock_path = hsaco + ".lock"
        try:
            with open(lock_path, "w") as lock:
                try:
                    fcntl.flock(lock, fcntl.LOCK_EX)
                except OSError:
                    pass
                try:
                    if not os.path.exists(hsaco):
                        _compile_code_object(model, source, tag, hsaco)
                finally:
                    try:
                        fcntl.flock(lock, fcntl.LOCK_UN)
                    except OSError:
                        pass
        finally:
            # (also after a failed or interrupted build: nothing of it stays in the cache)
            for leftover in [lock_path] + [os.path.join(CACHE_DIR, n) for n in os.listdir(CACHE_DIR)
                                            if n.endswith(".%d.tmp" % os.getpid())]:
                try:
                    os.remove(leftover)
                except OSError:
                    pass
    return hsaco, spec


_hipcc_version = None
#: hipcc runs on generated kernels in this process (tests: N ranks on a cold cache compile once)
BUILD_COUNT = 0


def hipcc_version():
    """First line of ``hipcc --version`` (part of the cache key: a code object is only
    reused with the compiler that built it)."""
    global _hipcc_version
    if _hipcc_version is None:
        try:
            res = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True)
            lines = [ln.strip() for ln in res.stdout.splitlines() if ln.strip()]
            _hipcc_version = " / ".join(ln for ln in lines if "version" in ln.lower())[:200] or "unknown"
        except OSError:
            _hipcc_version = "unknown"
    return _hipcc_version


def evict_stale_cache(keep=()):
    """Remove cached code objects that were built against another version of the kernel
    skeleton (``csrc/*.h``) or another hipcc: they can never be selected again (the tag
    hashes both), they only make the cache -- which travels with the package -- grow."""
    if not os.path.isdir(CACHE_DIR):
        return 0
    stamp, ver, removed = _skeleton_stamp(), hipcc_version(), 0
    for name in os.listdir(CACHE_DIR):
        if not (name.startswith("model_") and name.endswith(".json")):
            continue
        base = os.path.join(CACHE_DIR, name[:-len(".json")])
        try:
            with open(base + ".json") as f:
                meta = json.load(f)
        except (OSError, ValueError):
            continue
        if meta.get("skeleton") == stamp and meta.get("hipcc") == ver:
            continue
        if os.path.basename(base) in keep:
            continue
        for ext in (".hsaco", ".alt.hsaco", ".hip", ".json"):
            try:
                os.remove(base + ext)
                removed += 1
            except OSError:
                pass
    return removed


class HipBackend:
    """hipcc + libtriflow_hip.so (the product back end)."""

    _lib = None
    _lock = threading.Lock()

    def library(self):
        with HipBackend._lock:
            if HipBackend._lib is None:
                HipBackend._lib = Library(LIB_PATH)
            return HipBackend._lib

    def load(self, model, parvec_mask, device=-1, seg=None):
        hsaco, spec = build_code_object(model, parvec_mask, seg=seg)
        lib = self.library()
        is_dev, ndev = lib.runtime_info()
        if ndev < 1:
            raise RuntimeError("no HIP device visible: the triflow_amd compute path needs "
                               "an MI355X (gfx950); there is no CPU fallback")
        if device >= 0:
            lib.set_device(device)          # the code object belongs to this GPU
        with open(hsaco, "rb") as f:
            code = f.read()
        alt, alt_kernels = alternate_of(hsaco)
        alt_code = None
        if alt_kernels:
            _warn_wide(model, alt_kernels)
        if alt:
            with open(alt, "rb") as f:
                alt_code = f.read()
        return DeviceModel(lib, spec, code, alt_code, alt_kernels)


_default_backend = HipBackend()


# --------------------------------------------------------------------------
# Jacobian pattern (compilers.py:303-331), computed once per (N, periodic)
# --------------------------------------------------------------------------
class CscPattern:
    """Fixed CSC structure of the Jacobian and the map from the raw
    ``[N, nnz]`` value table to its ``data`` array (duplicates from clamped
    ghost columns are summed in table order, like ``csc_matrix((data, (rows,
    cols)))`` does)."""

    def __init__(self, nvar, mp, sparse_indices, N, periodic):
        k = np.asarray(sparse_indices, dtype=np.int64)
        eq, slot = k % nvar, k // nvar
        var, off = slot % nvar, slot // nvar - mp
        node = np.arange(N, dtype=np.int64)[:, None]
        neigh = node + off[None, :]
        neigh = neigh % N if periodic else np.clip(neigh, 0, N - 1)
        rows = (node * nvar + eq[None, :]).ravel()
        cols = (neigh * nvar + var[None, :]).ravel()
        n = N * nvar
        key = cols * n + rows                       # CSC order: column major
        uniq, inverse = np.unique(key, return_inverse=True)
        self.shape = (n, n)
        self.slot = inverse
        self.nslots = uniq.size
        self.indices = (uniq % n).astype(np.int32 if n < 2 ** 31 else np.int64)
        ucols = uniq // n
        self.indptr = np.searchsorted(ucols, np.arange(n + 1)).astype(self.indices.dtype)
        # for the device-side gather: the first table entry of every slot, then the further
        # entries of slots with duplicates (clamped ghost columns: a handful of boundary entries)
        # in table order, which is the order csc_matrix() sums them in
        order = np.argsort(inverse, kind="stable")
        first = np.ones(order.size, dtype=bool)
        first[1:] = inverse[order[1:]] != inverse[order[:-1]]
        self.extra_src = order[~first]
        self.extra_slot = inverse[self.extra_src]
        self.gather = np.concatenate([order[first], self.extra_src]).astype(np.int32) \
            if key.size < 2 ** 31 else None

    def assemble(self, values):
        data = np.bincount(self.slot, weights=np.asarray(values).ravel(),
                           minlength=self.nslots)
        return sps.csc_matrix((data, self.indices, self.indptr), shape=self.shape)

    def from_gathered(self, gathered):
        """``gathered``: the value table in the order of ``self.gather`` (device-side gather)."""
        data = gathered[:self.nslots]
        if self.extra_src.size:
            data = data.copy()
            np.add.at(data, self.extra_slot, gathered[self.nslots:])
        return sps.csc_matrix((data, self.indices, self.indptr), shape=self.shape)


# --------------------------------------------------------------------------
class CompiledModel:
    """Device side of one Model: code objects per parameter layout, resident
    solvers per (N, periodic, nsys), and the host boundary of seam #1."""

    def __init__(self, model, backend=None):
        self.model = model
        self.backend = backend or _default_backend
        self.nvar = model._nvar
        self.fields = list(model._dep_vars) + list(model._help_funcs)
        self.nh = len(model._help_funcs)
        self.pars = list(model._pars)
        self.mp = max((model._window_range - 1) // 2, 1)
        self.real_mp = (model._window_range - 1) // 2
        self._device_models = {}
        self._solvers = {}
        self._patterns = {}
        # fail at compile time, like the reference does, if the expressions
        # cannot be lowered
        codegen.lower_model(model)

    # ---- code objects / solvers ------------------------------------------------
    @staticmethod
    def sweep_segment(total_nodes):
        """Nodes per thread of the stencil sweeps (TF_SEG): short segments when the grid has
        too few chunks to fill the GPU otherwise (N = 1e6: spmv 43 -> 35 us, stage sweep 31 ->
        27 us), the longer ones for big batches (8 members / N = 4e6: 1-2 % the other way)."""
        if "TRIFLOW_SWEEP_SEG" in os.environ:
            return int(os.environ["TRIFLOW_SWEEP_SEG"])
        return 4 if total_nodes <= 2_000_000 else 8

    def device_model(self, parvec_mask=0, device=-1, seg=None):
        key = (parvec_mask, device, seg)
        if key not in self._device_models:
            kw = {}
            if device >= 0:
                kw["device"] = device
            if seg is not None:
                kw["seg"] = seg          # (test back ends without the knob are never passed it)
            self._device_models[key] = self.backend.load(self.model, parvec_mask, **kw)
        return self._device_models[key]

    def solver(self, N, periodic, nsys=1, parvec_mask=0, **opts):
        key = (int(N), bool(periodic), int(nsys), int(parvec_mask), tuple(sorted(opts.items())))
        if key not in self._solvers:
            seg = self.sweep_segment(int(N) * int(nsys)) if isinstance(self.backend, HipBackend) else None
            dm = self.device_model(parvec_mask, int(opts.get("device", -1)), seg)
            self._solvers[key] = DeviceSolver(dm, N, nsys=nsys, periodic=periodic, **opts)
        return self._solvers[key]

    def pattern(self, N, periodic):
        key = (int(N), bool(periodic))
        if key not in self._patterns:
            self._patterns[key] = CscPattern(self.nvar, self.real_mp,
                                             self.model._sparse_indices[0], N, periodic)
        return self._patterns[key]

    def release(self):
        for s in self._solvers.values():
            s.close()
        self._solvers.clear()

    # ---- binding inputs ----------------------------------------------------------
    @staticmethod
    def grid_spacing(x):
        """``dx = (x[-1] - x[0]) / (N - 1)``; a ``dx`` parameter is ignored
        (reference compilers.py:234-237)."""
        x = np.asarray(x)
        return (x[-1] - x[0]) / (x.size - 1)

    def parvec_mask_of(self, par_values):
        mask = 0
        for k, v in enumerate(par_values):
            if np.ndim(v) > 0 and np.size(v) > 1 and np.ptp(v) != 0:
                mask |= 1 << k
        return mask

    def bind_inputs(self, solver, x, par_values, helper_arrays=None):
        spec = solver.model.spec
        dx = self.grid_spacing(x)
        solver.set_dx(dx)
        solver.set_x(x)
        mask = spec["parvec_mask"]
        for k, v in enumerate(par_values):
            if (mask >> k) & 1:
                solver.set_param(k, np.asarray(v, dtype=float))
            else:
                solver.set_param(k, float(np.ravel(v)[0]))
        # node-independent pow()/libm sub-expressions, evaluated with NumPy on the host
        for j, value in enumerate(codegen.eval_host_constants(spec, dx, par_values)):
            solver.set_param(spec["npar_model"] + j, value)
        if self.nh and helper_arrays is not None:
            solver.set_helpers(np.asarray(helper_arrays, dtype=float))

    # ---- seam #1 callables ---------------------------------------------------------
    def _unpack(self, args):
        x = np.asarray(args[0], dtype=float)
        nf = self.nvar + self.nh
        fields = [np.asarray(a, dtype=float) for a in args[1:1 + nf]]
        pars = list(args[1 + nf:1 + nf + len(self.pars)])
        periodic = bool(args[1 + nf + len(self.pars)])
        if len(args) != 2 + nf + len(self.pars):
            raise TypeError("expected %d positional arguments, got %d"
                            % (2 + nf + len(self.pars), len(args)))
        return x, fields, pars, periodic

    def _evaluate(self, args, with_j):
        x, fields, pars, periodic = self._unpack(args)
        mask = self.parvec_mask_of(pars)
        solver = self.solver(x.size, periodic, 1, mask)
        self.bind_inputs(solver, x, pars, fields[self.nvar:] if self.nh else None)
        solver.set_state(0, np.asarray(fields[:self.nvar]))
        solver.eval(0, with_j=with_j)
        return solver, x.size, periodic

    def F_function(self, *args):
        solver, N, periodic = self._evaluate(args, with_j=False)
        return solver.get_F()[0]

    def J_function(self, *args):
        solver, N, periodic = self._evaluate(args, with_j=True)
        pat = self.pattern(N, periodic)
        if pat.gather is None:
            return pat.assemble(solver.get_J()[0])
        # the CSC data array is gathered on the device in the pattern's order (index list
        # uploaded once per solver): the download is the matrix, nothing is assembled on the host
        if getattr(solver, "_csc_pattern", None) is not pat:
            solver.set_csc_map(pat.gather)
            solver._csc_pattern = pat
        return pat.from_gathered(solver.get_J_mapped())


def hip_compiler(model, backend=None):
    """``compiler(model) -> (F_function, J_function)`` for the MI355X."""
    compiled = CompiledModel(model, backend)

    def F_function(*args):
        return compiled.F_function(*args)

    def J_function(*args):
        return compiled.J_function(*args)

    F_function.device_model = compiled
    J_function.device_model = compiled
    return F_function, J_function


def resolve_compiler(name):
    """String spellings accepted by ``Model(compiler=...)``.  The reference maps
    ``"theano"`` / ``"numpy"`` to its two CPU back ends (model.py:152-155); this
    framework has a single back end, so every spelling selects it."""
    if name in ("hip", "theano", "numpy"):
        if name != "hip":
            log.info("compiler=%r requested: using the HIP compiler (only back end)", name)
        return hip_compiler
    raise ValueError("unknown compiler %r" % (name,))
