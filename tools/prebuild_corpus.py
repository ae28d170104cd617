"""Compile the code objects of every corpus model into triflow_amd/_cache so
that a GPU test run does not spend its time in hipcc (the cache travels with
the repository snapshot)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import corpus
from triflow_amd import Model, compilers

for name in sorted(corpus.MODELS_ALL):
    t = time.time()
    m = Model(*corpus.model_args(name), compiler=lambda m: (None, None))
    for mask in (0,):
        compilers.build_code_object(m, mask, seg=4)      # test sizes are small: TF_SEG = 4
    print("%-16s %.1fs" % (name, time.time() - t), flush=True)
