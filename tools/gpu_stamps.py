"""In-kernel phase timing of the level-1 walks and of the reduced-level factorisation (diagnostic build -DTF_STAMPS):
shader-clock stamps of one workgroup per level, printed as cycles per phase."""
import os, sys
os.environ["TRIFLOW_HIPCC_EXTRA"] = (os.environ.get("TRIFLOW_HIPCC_EXTRA", "") + " -DTF_STAMPS").strip()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from triflow_amd import Model, workloads
from triflow_amd.ensemble import Ensemble

members = int(sys.argv[1]) if len(sys.argv) > 1 else 1
table = bench.member_table(members, None)
name, x, fields, pars, dt, scheme = bench.build_problem(3, None, table)
model = Model(*workloads.model_args(name))
ens = Ensemble(model, x, fields, pars, True, scheme=scheme, nstate=2)
s = ens.solver
s.debug_stamps()                     # switch on
for _ in range(5):
    ens.step(dt)
ens.sync()
st = s.debug_stamps().astype(np.int64)
print("levels", s.describe()["chunks"])
r = st[0]
if r[10]:
    print("level 1, factorisation walk (down, band wavefront): first rows %d | %d nodes of the loop %d (one node: %d) | tip pivots, tips, separator row %d cycles"
          % (r[11] - r[10], s.describe()["chunks"][0] and 0 or 0, r[12] - r[11], r[15] - r[14], r[13] - r[12]))
if r[30]:
    print("         right-hand side wavefront, relative to the band wavefront's entry: entry %d, loop from %d to %d, barriers passed %d, tips and separator row done %d | band wavefront: tips done %d, last barrier passed %d, records stored %d"
          % (r[30] - r[10], r[31] - r[10], r[32] - r[10], r[33] - r[10], r[34] - r[10], r[13] - r[10], r[16] - r[10], r[17] - r[10]))
if r[0]:
    print("level 1, re-elimination + back-substitution (down half): first rows %d | loop %d (one node: %d) | barrier %d | to the middle system %d | its solve %d | back-substitution %d cycles"
          % (r[1] - r[0], r[2] - r[1], r[5] - r[4], r[3] - r[2], r[6] - r[3], r[7] - r[6], r[8] - r[7]))
if r[20]:
    print("level 1, first solve walk (down): first rows %d | loop %d (one node: %d) | tips %d cycles"
          % (r[21] - r[20], r[22] - r[21], r[25] - r[24], r[23] - r[22]))
for l in range(1, len(s.describe()["chunks"])):
    r = st[l]
    if r[0] == 0:
        continue
    total, real = r[21] - r[0], (r[31] - r[30]) / 100.0        # s_memrealtime: 100 MHz
    marks = [r[0], r[1]] + [v for v in r[2:20] if v] + [r[20], r[21]]
    d = np.diff(marks)
    print("level %d: %d cycles = %.2f us (clock %.2f GHz)  load %d | rounds (tasks, barrier): %s | share %d | fold/end %d"
          % (l + 1, total, real, total / real / 1e3 if real else 0, d[0],
             " ".join("(%d, %d)" % (d[1 + 2 * i], d[2 + 2 * i]) for i in range((len(d) - 3) // 2)), d[-2], d[-1]))
    if r[44] and r[47]:
        # (tf_cr3_hip.h: one task per node and round; the two marks of a round are "tasks done" and "barrier passed")
        print("         round 2, wavefront 0: task starts %d cycles after the barrier; old values + block inversion %d, "
              "record stores %d, products %d, neighbour stores %d, until the barrier is passed %d"
              % (r[44] - marks[3], r[45] - r[44], r[46] - r[45], r[47] - r[46], r[48] - r[47], marks[5] - r[48]))
    if r[50]:
        print("         tail kernel (this level + the last one): " + " ".join(str(int(v)) for v in np.diff(r[50:57])) + " cycles (requests, forward, park, last level, barrier, backward)")
    if r[57] and r[61]:
        print("         level 1 + this level forwards in one launch (tfk_l1_solve_cr, wavefront 0): walks %d | barrier %d | first chunk %d | second chunk %d cycles"
              % (r[58] - r[57], r[59] - r[58], r[60] - r[59], r[61] - r[60]))
        if r[42]:
            print("             first chunk: records in LDS after %d | rounds %d | end %d" % (r[42] - r[59], r[43] - r[42], r[60] - r[43]))
    if r[62] and r[63]:
        print("         this level + level 1 backwards in one launch (tfk_l1_fwd2_backsub_cr, wavefront 0): its two chunks %d cycles" % (r[63] - r[62]))
    if r[41]:
        print("         block inversions %d, of which with the pivot search %d (all steps so far)" % (r[41], r[40]))
