"""TEST-ONLY: build the host emulation of the per-model kernels.

Compiles the host runtime (triflow_amd/csrc/tf_rt_*.cpp, tf_solver_*.cpp) + tests/emu/tf_backend_emu.cpp +
the generated model header with g++ into tests/emu/_build/emu_<hash>.so.  The
resulting library exposes the same C ABI as libtriflow_hip.so but executes the
kernel bodies on the CPU; it exists so that the CPU test suite can check the
host orchestration and the kernel arithmetic against the oracle.  The
triflow_amd package never loads it.
"""
import os
import subprocess

from triflow_amd import codegen
from triflow_amd.compilers import RUNTIME_SOURCES

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "triflow_amd", "csrc")
BUILD = os.path.join(HERE, "_build")


def _deps_stamp():
    parts = []
    for d, names in ((CSRC, ("tf_args.h", "tf_math.h", "tf_kernels.h", "tf_crs.h", "tf_solver.h", *RUNTIME_SOURCES,
                             "tf_backend.h")),
                     (HERE, ("tf_backend_emu.cpp",)),
                     (os.path.join(ROOT, "include"), ("triflow_hip.h",))):
        for n in names:
            with open(os.path.join(d, n), "rb") as f:
                parts.append(f.read())
    return codegen.source_hash(*parts)


def build(model, parvec_mask=0, opt="-O1"):
    """Returns (path of the emulation library, spec dict)."""
    src, spec = codegen.lower_model(model, parvec_mask=parvec_mask)
    tag = codegen.source_hash(src, _deps_stamp(), opt)
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "emu_%s.so" % tag)
    if not os.path.exists(so):
        # (pytest-xdist workers build the same library at the same time: every file under its own name
        # until it is complete)
        hdr = os.path.join(BUILD, "model_%s.h" % tag)
        with open(hdr + ".%d.tmp" % os.getpid(), "w") as f:
            f.write(src)
        os.replace(hdr + ".%d.tmp" % os.getpid(), hdr)
        tmp = so + ".%d.tmp" % os.getpid()
        # (one translation unit for the whole host runtime: its parts share one big header, and this
        # library is rebuilt per test model)
        unity = os.path.join(BUILD, "runtime_%s.%d.cpp" % (tag, os.getpid()))
        with open(unity, "w") as f:
            f.write("".join('#include "%s"\n' % os.path.join(CSRC, n) for n in RUNTIME_SOURCES))
        cmd = ["g++", "-std=c++17", *opt.split(), "-g0", "-shared", "-fPIC", "-ffp-contract=off",
               "-fno-fast-math", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
               '-DTF_EMU_MODEL_HEADER="%s"' % hdr,
               unity, os.path.join(HERE, "tf_backend_emu.cpp"),
               "-o", tmp]
        res = subprocess.run(cmd, capture_output=True, text=True)
        os.remove(unity)
        if res.returncode != 0:
            raise RuntimeError("emulation build failed:\n" + res.stderr[-4000:])
        os.replace(tmp, so)
    return so, spec


class EmuBackend:
    """Drop-in for triflow_amd.compilers.HipBackend in CPU tests."""

    def __init__(self):
        self._libs = {}

    def load(self, model, parvec_mask):
        from triflow_amd._capi import DeviceModel, Library
        so, spec = build(model, parvec_mask)
        if so not in self._libs:
            self._libs[so] = Library(so)
        return DeviceModel(self._libs[so], spec, b"")
