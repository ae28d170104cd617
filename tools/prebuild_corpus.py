"""Compile the code objects of every corpus model into triflow_amd/_cache so
that a GPU test run does not spend its time in hipcc (the cache travels with
the repository snapshot).  usage: python tools/prebuild_corpus.py [workers]"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def build(name):
    from oracle import corpus
    from triflow_amd import Model, compilers
    t = time.time()
    m = Model(*corpus.model_args(name), compiler=lambda m: (None, None))
    compilers.build_code_object(m, 0, seg=4)      # test sizes are small: TF_SEG = 4
    return "%-16s %.1fs" % (name, time.time() - t)


if __name__ == "__main__":
    from oracle import corpus
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    with ProcessPoolExecutor(workers) as pool:
        for line in pool.map(build, sorted(corpus.MODELS_ALL)):
            print(line, flush=True)
