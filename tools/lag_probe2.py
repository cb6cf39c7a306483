"""iteration-by-iteration comparison of the lagged path with the oracle (BART block only)"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import friedman_case, make_sampler   # noqa: E402
from stan4bart_amd._lib import load_library   # noqa: E402
hlib = load_library()
olib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
for T in (3, 4, 6):
    args, _ = friedman_case(n=400, T=T, warmup=5, iter=10, ranef=False)
    so, sh = make_sampler(olib, "orc_", args), make_sampler(hlib, "s4b_", args)
    sh.set_tree_path("lagged"); so.set_trace(True); sh.set_trace(True)
    to, th = so.get_trees(), sh.get_trees()
    print("T", T, "after create: n equal", np.array_equal(to["n"], th["n"]), "value maxdiff", float(np.max(np.abs(to["value"] - th["value"]))), sh.get_lag_stats())
    for it in range(6):
        ro, rh = so.run(1, True, 1), sh.run(1, True, 1)
        to, th = so.get_trees(), sh.get_trees()
        tro, trh = so.get_trace(), sh.get_trace()
        same_shape = to["n"].shape == th["n"].shape
        print("  iter", it, "trace equal", np.array_equal(tro, trh), "struct equal", same_shape and np.array_equal(to["var"], th["var"]), "n equal", same_shape and np.array_equal(to["n"], th["n"]),
              "value maxdiff", float(np.max(np.abs(to["value"] - th["value"]))) if same_shape else None,
              "fit maxdiff", float(np.max(np.abs(ro["bart"]["train"] - rh["bart"]["train"]))), sh.get_lag_stats()["launches_per_sweep"], sh.get_lag_stats()["repairs_per_sweep"])
        if not np.array_equal(tro, trh):
            print("   oracle", tro.tolist()); print("   hip   ", trh.tolist())
            if same_shape:
                bad = np.nonzero(np.abs(to["value"] - th["value"]) > 1e-9)[0]
                print("   first value diffs at rows", bad[:10].tolist(), "tree", to["tree"][bad[:10]].tolist())
            break
    so.free(); sh.free()
