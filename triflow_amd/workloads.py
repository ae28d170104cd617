"""Benchmark workloads: the BASELINE.json configurations.

Model strings and deterministic synthetic inputs of configs 1, 2, 3 and 5; used by
``bench.py``, ``__graft_entry__`` and (through ``oracle/corpus.py``) by the tests.
"""

import numpy as np

#: name -> (equations, dependent variables, parameters, help functions)
BENCH_MODELS = {
    "M1_advdiff": ("k * dxxU - c * dxU", "U", ["k", "c"], None),
    "M2_diff": ("k * dxxU", "U", "k", None),
    "M3_film": (["-dxq",
                 "-upwind(c, q, 2) + (h - q / h**2) / eps + We * h * dxxxh + k * dxxq",
                 "-upwind(c, T, 2) + k * dxxT - q * dxT / h"],
                ["h", "q", "T"], ["c", "eps", "We", "k"], None),
    "M5_stiff": (["Dm*dxxA - k1*A + k3*B*C",
                  "Dm*dxxB + k1*A - k3*B*C - k2*B**2",
                  "Dm*dxxC + k2*B**2 - k4*C*D",
                  "Dm*dxxD - upwind(c, D, 1) - k4*C*D",
                  "Dm*dxxE + k4*C*D"],
                 ["A", "B", "C", "D", "E"],
                 ["Dm", "k1", "k2", "k3", "k4", "c"], None),
}


def model_args(name):
    return BENCH_MODELS[name]


def config_inputs(cfg, N=None):
    """(model name, fields dict, parameter dict, dt, scheme name) of BASELINE
    config 1, 2, 3 or 5 at ``N`` nodes (default: the configured size)."""
    two_pi = 2 * np.pi
    if cfg == 1:
        N = N or 200
        x = np.linspace(0, 1, N)
        return ("M1_advdiff", dict(x=x, U=np.cos(two_pi * x * 5)),
                dict(c=.03, k=.001, periodic=False), 0.5, "Theta")
    if cfg == 2:
        N = N or 10 ** 6
        x = np.linspace(0, 1, N, endpoint=False)
        return ("M2_diff", dict(x=x, U=np.cos(two_pi * 5 * x)),
                dict(k=1e-3, periodic=True), 1e-2, "Theta")
    if cfg == 3:
        N = N or 10 ** 6
        x = np.linspace(0, 100, N, endpoint=False)
        h = 1 + 0.1 * np.cos(two_pi * 4 * x / 100)
        return ("M3_film", dict(x=x, h=h, q=h ** 3, T=np.sin(two_pi * x / 100)),
                dict(c=1., eps=.5, We=.01, k=.05, periodic=True), 1e-3, "ROS2")
    if cfg == 5:
        N = N or 4 * 10 ** 6
        x = np.linspace(0, 1, N)
        zero = np.zeros(N)
        return ("M5_stiff",
                dict(x=x, A=np.ones(N), B=zero.copy(), C=zero.copy(),
                     D=np.exp(-((x - .5) / .1) ** 2), E=zero.copy()),
                dict(Dm=1e-4, k1=.04, k2=3e7, k3=1e4, k4=1., c=.1, periodic=False),
                1e-3, "BDF2")
    raise ValueError(cfg)


