#!/bin/bash
# Fuzz / soak campaign of a round on the GPU box (outside the suite: thousands of seeded configurations against the oracle):
#     gpurun --timeout 2400 -- 'bash tools/gpu/campaign.sh r06'
set -u
TAG=${1:-r06}; O=gpurun_out/$TAG; mkdir -p "$O"
R=$O/campaign.txt; : > "$R"
echo "== $TAG campaign on $(hostname), libs4b.so $(sha256sum stan4bart_amd/csrc/libs4b.so | cut -c1-16)" >> "$R"
timeout 1500 python tools/fuzz_large.py 0 160 persistent 2>&1 | tail -n 1 >> "$R"
timeout 900 python tools/fuzz_large.py 0 80 persistent sp 2>&1 | tail -n 1 >> "$R"
timeout 900 python tools/fuzz_large.py 0 80 persistent w 2>&1 | tail -n 1 >> "$R"
timeout 900 python tools/fuzz_range.py 160 1360 persistent 2>&1 | tail -n 1 >> "$R"
timeout 600 python tools/fuzz_range.py 160 760 fused 2>&1 | tail -n 1 >> "$R"
timeout 600 python tools/fuzz_range.py 160 760 two-kernel 2>&1 | tail -n 1 >> "$R"
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_linear.so timeout 900 python tools/fuzz_large.py 0 80 persistent 2>&1 | tail -n 1 | sed 's/^/linear variant: /' >> "$R"
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_linear.so timeout 900 python tools/fuzz_range.py 160 960 persistent 2>&1 | tail -n 1 | sed 's/^/linear variant: /' >> "$R"
timeout 900 python tools/soak.py 2>&1 | tail -n 6 >> "$R"
timeout 600 python tools/multi_chain_probe.py 2>&1 | tail -n 5 >> "$R"
cat "$R"
