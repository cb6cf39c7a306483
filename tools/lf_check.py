import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); os.chdir('/root/repo')
import bench
from stan4bart_amd import RRng
from stan4bart_amd.abi import Sampler
from stan4bart_amd.fit import chain_seeds
from stan4bart_amd._lib import load_library
lib = load_library()
d = bench.friedman_design(1000000, 50, 0, 1, lambda: None)
for variant, burn in (("burn 150", 150), ("burn 600", 600), ("burn 1500", 1500)):
    args = bench.case_from_design(d, 50, 200, 0, burn, burn + 400)
    rng = RRng(int(chain_seeds(20260101, 1)[0])); args.seed = int(rng.sample_int(2147483647, 1)[0])
    s = Sampler(lib, "s4b_", args, rng.state)
    s.run(burn, True, 0); s.disengage_adaptation(); s.run(5, False, 0)
    for k in range(8):
        a = s.get_nuts_stats(); s.run(20, False, 0); b = s.get_nuts_stats()
        print(variant, k, (b["sum_n_leapfrog"] - a["sum_n_leapfrog"]) / 20, flush=True)
    s.free()
