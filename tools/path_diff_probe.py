#!/usr/bin/env python3
"""Debug helper: one friedman_case on a chosen tree path, oracle and HIP side by side, iteration by iteration, comparing the whole chain state.
    python tools/path_diff_probe.py stream [n] [T] [results_type]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_sampler, StateView, friedman_case
from stan4bart_amd._lib import load_library
path = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 100; T = int(sys.argv[3]) if len(sys.argv) > 3 else 11
rt = int(sys.argv[4]) if len(sys.argv) > 4 else 1
args, _ = friedman_case(n=n, T=T, warmup=7, iter=13)
olib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so")); hlib = load_library()
so, sp = make_sampler(olib, "orc_", args), make_sampler(hlib, "s4b_", args)
so.set_trace(True); sp.set_trace(True); sp.set_tree_path(path)
print("path", sp.get_tree_path())
for it in range(args.iter):
    warm = it < args.warmup
    if it == args.warmup:
        so.disengage_adaptation(); sp.disengage_adaptation()
    so.run(1, warm, rt); sp.run(1, warm, rt)
    a, b = StateView(so.get_state()), StateView(sp.get_state())
    tf = np.abs(a.get("total_fits") - b.get("total_fits"))
    ta, tb = so.get_trace(), sp.get_trace()
    same_struct = all(np.array_equal(na, nb_) for (na, ma), (nb_, mb) in zip(a.trees, b.trees))
    mus = max(np.abs(ma - mb).max() if ma.shape == mb.shape else 9e9 for (na, ma), (nb_, mb) in zip(a.trees, b.trees))
    first = next((i for i in range(min(len(ta), len(tb))) if not np.array_equal(ta[i], tb[i])), None)
    print("iter", it, "max |total_fits diff|", tf.max(), "at obs", int(tf.argmax()), "#obs off by > 1e-9:", int((tf > 1e-9).sum()), "max |mu diff|", mus, "structures equal", same_struct,
          "first differing tree update", first, flush=True)
    for k in range(T):
        la, lb = so.get_leaf_assignment(k), sp.get_leaf_assignment(k)
        if not np.array_equal(la, lb):
            w = np.nonzero(la != lb)[0]
            print("  leaf assignment of tree", k, "differs at", len(w), "observations, e.g.", w[:8].tolist(), "oracle", la[w[:8]].tolist(), "product", lb[w[:8]].tolist(),
                  "| moves on this tree this iteration: oracle", ta[k].tolist())
    if first is not None or tf.max() > 1e-6:
        print("  observations with a wrong fit:", np.nonzero(tf > 1e-9)[0].tolist()[:40])
        if first is not None:
            print("  oracle ", ta[max(0, first - 2):first + 2].tolist()); print("  product", tb[max(0, first - 2):first + 2].tolist())
        for k, ((na, ma), (nb_, mb)) in enumerate(zip(a.trees, b.trees)):
            if not np.array_equal(na, nb_) or ma.shape != mb.shape or np.abs(ma - mb).max() > 1e-9:
                print("  tree", k, "oracle", na.tolist(), np.round(ma, 5).tolist(), "| product", nb_.tolist(), np.round(mb, 5).tolist())
        break
