"""per-leapfrog O(N) sums (k_stan_fused, direct): kernel time and time including the result hand-off, at n = 1e6 and 1e7"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import c5_case, make_sampler   # noqa: E402
from stan4bart_amd._lib import load_library   # noqa: E402
hlib = load_library()
for n in ([int(float(v)) for v in sys.argv[1:]] or (1000000, 10000000)):
    from stan4bart_amd import GroupTerm, make_sampler_args
    g = np.random.default_rng(5)
    xb = np.empty((n, 9), order="F")
    for j in range(9):
        xb[:, j] = g.random(n)
    x4 = g.random(n); z = (g.random(n) < 0.2).astype(np.float64)
    g1, g2 = g.integers(1, 6, size=n), g.integers(1, 9, size=n)
    y = 10 * np.sin(np.pi * xb[:, 0] * xb[:, 1]) + 10 * x4 + 5 * z + g.standard_normal(5)[g1 - 1] * (1 + x4) + g.standard_normal(8)[g2 - 1] + g.standard_normal(n)
    args = make_sampler_args(y, xb, X=np.column_stack([x4, z]), groups=[GroupTerm(g1, x4, "g.1"), GroupTerm(g2, None, "g.2")], iter=8, warmup=4,
                             keep_fits=False, bart_args={"n.trees": 20})
    s = make_sampler(hlib, "s4b_", args)
    s.run(2, True, 0)
    lf = s.profile_leapfrog(50)
    print(n, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in lf.items()}, "GB/s", round(lf["algorithmic_bytes"] / lf["kernels_us"] / 1e3, 1), s.get_fused_stats(), flush=True)
    s.free()
