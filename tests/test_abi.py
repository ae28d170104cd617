"""The C ABI: include/triflow_hip.h, the ctypes table and the built library agree."""
import os
import re
import subprocess

import pytest

from triflow_amd import _capi, compilers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(ROOT, "include", "triflow_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tf_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_capi.SIGNATURES)


def test_library_builds_and_exports_every_symbol():
    """hipcc cross-compiles the host runtime without a GPU; no compute call."""
    lib_path = compilers.build_runtime_library()
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True,
                         text=True, check=True).stdout
    exported = set(re.findall(r" T (tf_[A-Za-z0-9_]+)", out))
    assert set(declared_symbols()) <= exported
    lib = _capi.Library(lib_path)          # loads, binds every entry point
    assert lib.dll.tf_kernel_count() == len(lib.kernel_names())
    is_device, ndev = lib.runtime_info()
    assert is_device


def test_product_has_no_cpu_fallback():
    """Without a device the compute path must raise, not fall back."""
    lib = _capi.Library(compilers.build_runtime_library())
    _, ndev = lib.runtime_info()
    if ndev > 0:
        pytest.skip("a GPU is present")
    from triflow_amd import Model
    m = Model("k * dxxU", "U", "k")
    import numpy as np
    x = np.linspace(0, 1, 20)
    with pytest.raises(RuntimeError):
        m.F(m.fields_template(x=x, U=x), dict(k=1., periodic=True))


def test_vec_op_enums_in_sync():
    with open(os.path.join(compilers.CSRC, "tf_kernels.h")) as f:
        k = dict(re.findall(r"(TF_VEC_[A-Z0-9_]+) = (\d+),", f.read()))
    r = {}
    for name in ("tf_solver.h",) + tuple(compilers.RUNTIME_SOURCES):
        with open(os.path.join(compilers.CSRC, name)) as f:
            r.update(re.findall(r"(TF_VEC_[A-Z0-9_]+) = (\d+)", f.read()))
    assert r and all(k[name] == val for name, val in r.items())


def test_code_object_holds_every_kernel():
    """The generated per-model code object cross-compiles for gfx950 and
    contains every entry point of the runtime's kernel table."""
    from triflow_amd import Model
    m = Model("k * dxxU", "U", "k", hold_compilation=True)
    hsaco, spec = compilers.build_code_object(m)
    syms = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", hsaco],
                          capture_output=True, text=True, check=True).stdout
    lib = _capi.Library(compilers.build_runtime_library())
    for name in lib.kernel_names():
        assert re.search(r"\b%s\b" % name, syms), name
