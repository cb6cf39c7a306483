import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import friedman_case, run_chain, assert_chain_parity
import ctypes
from stan4bart_amd._lib import load_library
hip = load_library()
orc = ctypes.CDLL('oracle/_build/liboracle.so')
t0 = time.time()
# 1. long BART-only parity run (speculation must never change the stream)
args, _ = friedman_case(n=20000, T=50, warmup=150, iter=300)
a = run_chain(orc, "orc_", args, results_type=1)
for path in ("persistent", "fused", "two-kernel"):
    b = run_chain(hip, "s4b_", args, results_type=1, tree_path=path)
    assert b["tree_path"][1] == path
    assert_chain_parity(a, b, stan=False)
    print("long BART parity ok on the", path, "path:", len(a["trace"]), "tree updates, accept rate", (a["trace"][:, 1] == 1).mean().round(3), round(time.time() - t0, 1), "s", flush=True)
# 2. soak: many iterations at n = 1e5, joint chain, gaussian and binary
for binary in (False, True):
    from stan4bart_amd import GroupTerm, generate_friedman_data, make_sampler_args, RRng
    from stan4bart_amd.abi import Sampler
    d = generate_friedman_data(100000, ranef=True, causal=True, binary=binary, p=10)
    x = d["x"]
    args = make_sampler_args(d["y"], x[:, [0,1,2,4,5,6,7,8,9]], X=np.column_stack([x[:, 3], d["z"]]),
                             groups=[GroupTerm(d["g1"], x[:, 3], "g.1"), GroupTerm(d["g2"], None, "g.2")],
                             family="binomial" if binary else "gaussian", iter=1200, warmup=600, keep_fits=False, bart_args={"n.trees": 200})
    rng = RRng(7); args.seed = int(rng.sample_int(2147483647, 1)[0])
    s = Sampler(hip, "s4b_", args, rng.state)
    t1 = time.time()
    s.run(600, True, 0); s.disengage_adaptation(); out = s.run(600, False, 0)
    print("soak binary=%s: 1200 iterations in %.1f s, sigma %.3f, counters %s" % (binary, time.time() - t1, out["bart"]["sigma"][-1], s.get_counters()))
    s.free()
