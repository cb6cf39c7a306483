import json, sys
r = json.load(open(sys.argv[1]))
print({k: (v if not isinstance(v, dict) else "...") for k, v in r.items()})
for k in ("config", "roofline", "per_chain_hmc_mode1", "cpu_baseline", "roofline_hmc", "roofline_target_config"):
    print(k, r.get(k))
