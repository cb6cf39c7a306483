#!/usr/bin/env python3
"""Debug helper: one seeded fuzz case (tests/test_gpu_fuzz.py) on one tree path, against the oracle."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import assert_chain_parity, run_chain
from test_gpu_fuzz import random_case
from stan4bart_amd._lib import load_library
seed, path = int(sys.argv[1]), sys.argv[2]
args, joint, what = random_case(seed)
print(what, "joint", joint, flush=True)
olib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so")); hlib = load_library()
rt = 0 if joint else 1
a = run_chain(olib, "orc_", args, results_type=rt)
print("oracle leaves max", a["trace"][:, 4].max(), "updates", len(a["trace"]), flush=True)
b = run_chain(hlib, "s4b_", args, results_type=rt, tree_path=path)
print(b["tree_path"], b.get("sweep_stats"))
assert_chain_parity(a, b, stan=joint)
print("parity OK")
