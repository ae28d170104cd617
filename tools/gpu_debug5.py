import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import corpus
from tests import parity_cases as pc
for name in ['M1_advdiff', 'M2_diff']:
    g = np.load('tests/golden/fj_%s.npz' % name)
    m = pc.device_model(name, None)
    for periodic in (True, False):
        fd = corpus.synthetic_fields(name, 24, seed=3, periodic=periodic)
        pars = corpus.synthetic_pars(name, 24, periodic, False)
        F = m.F(m.fields_template(**fd), pars)
        tag = "%s_sca" % ("per" if periodic else "clamp")
        d = F - g[tag+'_F']
        print(name, tag, 'nbad', (d != 0).sum(), 'max rel', np.abs(d/g[tag+'_F']).max(), np.nonzero(d)[0][:10], d[np.nonzero(d)[0][:5]])
        dx = (fd['x'][-1]-fd['x'][0])/23
        print('  dx', dx.hex(), 'dx*dx', (dx*dx).hex(), 'dx**2', (dx**2).hex())
