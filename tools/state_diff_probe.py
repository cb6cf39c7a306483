#!/usr/bin/env python3
"""Debug helper: one seeded fuzz case (tests/test_gpu_fuzz.py) on the persistent path, oracle and HIP side by side, iteration by iteration,
comparing the whole chain state (s4b_get_state: fits, tree structures, leaf values) — a wrong residual or leaf value shows here long before it
flips a tree move in the trace:   python tools/state_diff_probe.py 3738 [notrace]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_sampler, StateView
from test_gpu_fuzz import random_case
from stan4bart_amd._lib import load_library
seed = int(sys.argv[1])
args, joint, what = random_case(seed)
olib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so")); hlib = load_library()
so, sp = make_sampler(olib, "orc_", args), make_sampler(hlib, "s4b_", args)
tr = "notrace" not in sys.argv; so.set_trace(tr); sp.set_trace(tr); sp.set_tree_path("persistent")
def cmp(tag):
    a, b = StateView(so.get_state()), StateView(sp.get_state())
    tf = np.abs(a.get("total_fits") - b.get("total_fits")).max()
    bad = [k for k, ((na, ma), (nb_, mb)) in enumerate(zip(a.trees, b.trees)) if not np.array_equal(na, nb_) or ma.shape != mb.shape or np.abs(ma - mb).max() > 1e-9]
    print(tag, "max |total_fits diff|", tf, "trees that differ", bad, sp.get_sweep_stats(), flush=True)
cmp("after create")
for it in range(args.iter):
    warm = it < args.warmup
    if it == args.warmup:
        so.disengage_adaptation(); sp.disengage_adaptation()
    so.run(1, warm, 1); sp.run(1, warm, 1)
    a, b = StateView(so.get_state()), StateView(sp.get_state())
    tf = np.abs(a.get("total_fits") - b.get("total_fits")).max()
    mus = max(np.abs(ma - mb).max() if ma.shape == mb.shape else 9e9 for (na, ma), (nb_, mb) in zip(a.trees, b.trees))
    same_struct = all(np.array_equal(na, nb_) for (na, ma), (nb_, mb) in zip(a.trees, b.trees))
    ta, tb = so.get_trace(), sp.get_trace()
    tr_same = np.array_equal(ta, tb)
    print("  trace", ta.tolist()[:6])
    print("iter", it, "max |total_fits diff|", tf, "max |mu diff|", mus, "structures equal", same_struct, "trace equal", tr_same, sp.get_sweep_stats(), flush=True)
    if not tr_same or tf > 1e-6:
        print("  oracle trace", so.get_trace().tolist()[:8]) if False else None
        for k, ((na, ma), (nb_, mb)) in enumerate(zip(a.trees, b.trees)):
            if not np.array_equal(na, nb_) or ma.shape != mb.shape or np.abs(ma - mb).max() > 1e-9:
                if k == 2: print("   oracle nodes", na.tolist(), ma.tolist(), "\n   product nodes", nb_.tolist(), mb.tolist())
                print("  tree", k, "oracle nodes", len(na), "leaves", len(ma), "| product nodes", len(nb_), "leaves", len(mb), "structure equal", np.array_equal(na, nb_))
        break
