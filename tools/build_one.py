"""Build the code object of one benchmark model here (hipcc cross-compiles gfx950 without a GPU)
and print the register / scratch / occupancy table of selected kernels.
usage: python tools/build_one.py [model] [seg] [kernel-substring ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import corpus                                  # noqa: E402
from triflow_amd import Model, compilers, workloads       # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "M3_film"
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pats = sys.argv[3:] or ["cr_", "tail"]
args = workloads.BENCH_MODELS[name] if name in workloads.BENCH_MODELS else corpus.model_args(name)
t = time.time()
m = Model(*args, compiler=lambda m: (None, None))
hsaco, spec = compilers.build_code_object(m, 0, seg=seg)
print(hsaco, "%.1fs" % (time.time() - t))
for k, u in compilers.resource_usage(hsaco).items():
    if any(p in k for p in pats):
        print("%-24s %s" % (k, u))
