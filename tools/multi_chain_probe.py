"""How much of one MI355X a single chain leaves idle at n = 1e6: C independent chains of the bench workload on the same GPU,
one host thread + one HIP stream each (measurement only; the metric of bench.py stays one chain per GPU)."""
import sys, threading, time
import numpy as np
sys.path.insert(0, ".")
import bench
from stan4bart_amd import RRng
from stan4bart_amd.abi import Sampler
from stan4bart_amd._lib import load_library
from stan4bart_amd.fit import chain_seeds

lib = load_library()
n, p, trees, warm, steps = 1_000_000, 50, 200, 5, 20
design = bench.friedman_design(n, p, 0, 1, lambda: None)
for C in (1, 2, 3, 4, 6):
    samplers = []
    for c in range(C):
        args = bench.case_from_design(design, p, trees, 0, warm + steps, 2 * (warm + steps))
        rng = RRng(int(chain_seeds(20260101, C)[c]))
        args.seed = int(rng.sample_int(2147483647, 1)[0])
        samplers.append(Sampler(lib, "s4b_", args, rng.state))
        if "--no-hint" not in sys.argv:
            samplers[-1].set_device_sharing(C)
    def work(s, k):
        s.run(k, True, 0)
    th = [threading.Thread(target=work, args=(s, warm)) for s in samplers]
    [t.start() for t in th]; [t.join() for t in th]
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(s, steps)) for s in samplers]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"chains on one GPU: {C}  aggregate {C * steps / dt:.1f} it/s  per chain {steps / dt:.1f} it/s", flush=True)
    [s.free() for s in samplers]
