"""Sweep-kernel tuning matrix (run on the GPU): SEG x BLOCK x NT."""
import sys, os, itertools, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'one':
    import numpy as np
    from triflow_amd.model import Model
    from triflow_amd import workloads as corpus
    cfg = int(os.environ.get('CFG', '3'))
    name, fd, pars, dt, _ = corpus.config_inputs(cfg)
    m = Model(*corpus.model_args(name))
    cm = m._device
    solver = cm.solver(fd['x'].size, pars['periodic'], 1, 0, m1=int(os.environ.get('M1', '32')))
    cm.bind_inputs(solver, fd['x'], [pars[k] for k in cm.pars])
    solver.set_state(0, np.array([fd[k] for k in m._dep_vars]))
    solver.eval_repeat(0, True, 5)
    t = min(solver.eval_repeat(0, True, 50) for _ in range(3))
    tf_ = min(solver.eval_repeat(0, False, 50) for _ in range(3))
    nb = 8*(m._nvar*2 + len(m._J_sparse_array)) * fd['x'].size
    print(json.dumps(dict(seg=os.environ.get('TRIFLOW_SWEEP_SEG'), block=os.environ.get('TRIFLOW_SWEEP_BLOCK'), nt=os.environ.get('TRIFLOW_SWEEP_NT'), waves=os.environ.get('TRIFLOW_SWEEP_WAVES'), m1=os.environ.get('M1'), fj_us=round(t*1e3,2), f_us=round(tf_*1e3,2), fj_TBs=round(nb/t/1e9,3))))
else:
    for seg, block, nt, m1, *rest in json.loads(os.environ['MATRIX']):
        env = dict(os.environ, TRIFLOW_SWEEP_SEG=str(seg), TRIFLOW_SWEEP_BLOCK=str(block), TRIFLOW_SWEEP_NT=str(nt), M1=str(m1), TRIFLOW_SWEEP_WAVES=str(rest[0] if rest else 0))
        r = subprocess.run([sys.executable, __file__, 'one'], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]); sys.stdout.flush()
