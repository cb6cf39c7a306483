#!/usr/bin/env python3
"""Host cost of one NUTS leapfrog in hmc_mode 0 (sufficient statistics: no device work per leapfrog), measured over the CPU emulation of the
device layer (or the HIP library: `... libs4b.so s4b_`) at a small n: python tools/lf_host_probe.py [library [prefix]]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import friedman_case
from stan4bart_amd import RRng
from stan4bart_amd.abi import Sampler
lib = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests/emul/_build/libs4b_emul.so"))
args, d = friedman_case(n=2000, ranef=True, slopes=True, p=10, T=5, warmup=150, iter=650)
rng = RRng(1); args.seed = int(rng.sample_int(2147483647, 1)[0])
s = Sampler(lib, sys.argv[2] if len(sys.argv) > 2 else "emu_", args, rng.state)
s.run(150, True, 2); s.disengage_adaptation()
best = None
for rep in range(5):
    t0 = time.perf_counter(); out = s.run(100, False, 2); t1 = time.perf_counter()
    lf = out["stan"][4].sum(); us = 1e6 * (t1 - t0) / lf
    best = us if best is None else min(best, us)
print("leapfrogs per iteration %.1f, best of 5: %.3f us per leapfrog (includes the sweeps at n = 2000, T = 5)" % (lf / 100, best))
