"""Development probe for the lagged tree update (k_lag): parity against the oracle on small chains, then timing at n = 1e6.
Usage: python tools/lag_probe.py [parity|time|all]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import friedman_case, run_chain, make_sampler   # noqa: E402
from stan4bart_amd._lib import load_library   # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
hlib = load_library()
olib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))

def first_diff(a, b):
    m = min(len(a), len(b))
    d = np.nonzero((a[:m] != b[:m]).any(axis=1))[0]
    return int(d[0]) if len(d) else (None if len(a) == len(b) else m)

if what in ("parity", "all"):
    cases = [dict(n=2000, T=10, warmup=10, iter=25, ranef=False),
             dict(n=5000, T=50, warmup=10, iter=20, ranef=True),
             dict(n=2000, T=4, warmup=20, iter=40, ranef=False, bart_args={"base": 0.99, "power": 0.45, "k": 0.5}),
             dict(n=300, T=1, warmup=5, iter=10, ranef=False),
             dict(n=300, T=2, warmup=5, iter=10, ranef=False)]
    for kw in cases:
        args, _ = friedman_case(**kw)
        o = run_chain(olib, "orc_", args)
        for path in ("lagged", "fused"):
            try:
                h = run_chain(hlib, "s4b_", args, tree_path=path)
            except Exception as e:   # noqa: BLE001
                print(kw, path, "FAILED:", str(e)[:300]); continue
            fd = first_diff(o["trace"], h["trace"])
            dv = np.max(np.abs(o["sample"]["bart"]["train"] - h["sample"]["bart"]["train"]) / (1e-9 + np.abs(o["sample"]["bart"]["train"])))
            print(kw, path, h["tree_path"], "trace rows", len(o["trace"]), len(h["trace"]), "first diff", fd, "rng equal", bool(np.array_equal(o["rng"], h["rng"])),
                  "max rel fit diff %.2e" % dv, {k: round(v, 2) for k, v in h["lag_stats"].items()}, flush=True)
            if fd is not None:
                print("  oracle", o["trace"][max(0, fd - 2):fd + 3].tolist()); print("  hip   ", h["trace"][max(0, fd - 2):fd + 3].tolist())

if what in ("time", "all"):
    for n, P, T in ((1000000, 50, 200),):
        args, _ = friedman_case(n=n, T=T, p=P + 1, warmup=20, iter=30, ranef=True, slopes=True)
        for path in ("lagged", "fused"):
            s = make_sampler(hlib, "s4b_", args)
            try:
                s.set_tree_path(path)
                s.run(20, True, 1)
                t0 = time.time(); s.run(10, False, 1); dt = time.time() - t0
                prof = s.profile_sweep(3)
                print(f"n={n} T={T} path={s.get_tree_path()} BART-only ms/iter {dt / 10 * 1e3:.3f} profile {prof} lag {s.get_lag_stats()}", flush=True)
            finally:
                s.free()

if what == "big":
    from conftest import c5_case
    for n in (4000000, 10000000):
        args, _ = c5_case(n, P=49, T=200, n_groups=5, warmup=2, iter=4)
        for path in ("lagged", "two-kernel", "fused"):
            s = make_sampler(hlib, "s4b_", args)
            try:
                s.set_tree_path(path)
                s.run(3, True, 1)
                prof = s.profile_sweep(2)
                print(f"n={n} path={s.get_tree_path()} profile {prof} lag {s.get_lag_stats()}", flush=True)
            finally:
                s.free()
