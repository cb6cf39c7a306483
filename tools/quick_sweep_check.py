import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.chdir('/root/repo')
import ctypes
from conftest import friedman_case, run_chain, assert_chain_parity
from stan4bart_amd._lib import load_library
olib = ctypes.CDLL('oracle/_build/liboracle.so'); hlib = load_library()
import time
for kw in [dict(n=5000, T=40, warmup=30, iter=60), dict(n=1003, T=40, warmup=30, iter=60), dict(n=100000, T=50, warmup=5, iter=10, ranef=False)]:
    args, _ = friedman_case(**kw)
    a = run_chain(olib, "orc_", args, results_type=1)
    t0=time.time()
    b = run_chain(hlib, "s4b_", args, results_type=1, tree_path="persistent")
    print(kw, b["tree_path"], 'time', time.time()-t0, flush=True)
    try:
        assert_chain_parity(a, b, stan=False)
        print("  parity OK", flush=True)
    except AssertionError as e:
        print("  PARITY FAIL", str(e)[:600], flush=True)
        tr_a, tr_b = a["trace"], b["trace"]
        m = min(len(tr_a), len(tr_b))
        d = np.nonzero((tr_a[:m] != tr_b[:m]).any(axis=1))[0]
        print("  first trace diff at", d[:5], len(tr_a), len(tr_b))
        if len(d): print(tr_a[d[0]-2:d[0]+3], tr_b[d[0]-2:d[0]+3])
        break
