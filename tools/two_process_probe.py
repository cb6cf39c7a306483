#!/usr/bin/env python3
"""What happens when two PROCESSES run persistent sweeps on one GPU without the sharing hint (each under its own timeout):
    timeout 90 python tools/two_process_probe.py A & timeout 90 python tools/two_process_probe.py B; wait"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import friedman_case, make_sampler
from stan4bart_amd._lib import load_library
tag = sys.argv[1]
import numpy as np
from stan4bart_amd import make_sampler_args
g = np.random.default_rng(5)
n = 1_000_000
xb = g.random((n, 6)); x4 = g.random(n)
y = 10 * np.sin(np.pi * xb[:, 0] * xb[:, 1]) + 5 * xb[:, 3] + 10 * x4 + g.standard_normal(n)
args = make_sampler_args(y, xb, X=x4[:, None], groups=[], iter=8000, warmup=4000, keep_fits=False, bart_args={"n.trees": 20})
s = make_sampler(load_library(), "s4b_", args)
print(tag, "created", flush=True)
s.set_tree_path("persistent")
t0 = time.time()
try:
    for it in range(40):
        s.run(100, True, 1)
    print(tag, "finished 4000 iterations in", round(time.time() - t0, 2), "s; sweeps (run, handed over)", s.get_sweep_stats(),
          "persistent launches that found the device shared (their sweeps ran as k_step launches):", s.get_sweep_busy(), flush=True)
except RuntimeError as e:
    print(tag, "stopped after", round(time.time() - t0, 2), "s with:", str(e)[:300], flush=True)
