#!/usr/bin/env python3
"""Seeded random configurations at the sizes where k_sweep fills the device (n = 5e4 ... 1.04e6: up to 255 pass workgroups exchange their
bin partials), BART block only, against the oracle:  python tools/fuzz_large.py 0 60 [persistent|fused|two-kernel] [sp|w]
(sp: every configuration with cgm(split.probs = ): the persistent sweep as k_sweep_sp / k_sweep_few_sp; w: with observation weights: k_sweep_w)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import assert_chain_parity, run_chain
from large_cases import large_case
from stan4bart_amd._lib import load_library


if __name__ == "__main__":
    lo, hi, path = int(sys.argv[1]), int(sys.argv[2]), (sys.argv[3] if len(sys.argv) > 3 else "persistent")
    sp = len(sys.argv) > 4 and sys.argv[4] == "sp"
    wt = len(sys.argv) > 4 and sys.argv[4] == "w"
    olib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so")); hlib = load_library()
    bad, refused, handed, t0 = [], 0, 0, time.time()
    for seed in range(lo, hi):
        args, what = large_case(seed, split_probs=sp, weights=wt)
        print("seed", seed, what, flush=True)
        a = run_chain(olib, "orc_", args, results_type=1)
        try:
            b = run_chain(hlib, "s4b_", args, results_type=1, tree_path=path)
        except RuntimeError as e:
            if "node capacity exceeded" in str(e) or "outgrew node_capacity" in str(e):
                refused += 1; continue
            raise
        handed += b["sweep_stats"][1]
        try:
            assert_chain_parity(a, b, stan=False)
        except AssertionError as e:
            bad.append(seed); print("  FAILED", str(e)[:300], flush=True)
    print(f"large seeds {lo}..{hi - 1}{' with split.probs' if sp else (' with observation weights' if wt else '')} on the {path} path: {hi - lo - len(bad) - refused} ok, {refused} refused for their node capacity, failed {bad}; {handed} sweeps handed over; {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
