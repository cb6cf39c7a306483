#!/bin/bash
# Profiles of one round, run on the GPU box from the repository root:
#     bash tools/profile_round.sh r01
# For n = 1e6 (the metric's workload) and n = 1e7 (north_star's roofline target) it collects
#   1. rocprofv3 --kernel-trace --stats            -> profiles/<tag>_rocprofv3_prof_n<N>.txt   (per-kernel durations)
#   2. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -> profiles/<tag>_rocprofv3_pmc_{fetch,write}_n<N>.txt and
#      profiles/pmc_traffic.json (HBM bytes per launch of the tree-update kernel: k_sweep at n = 1e6 — one launch per sweep —, k_tree<true> at n = 1e7; gfx950 correction of MI355X_MICROARCH.md)
# Counter passes are separate runs without any tracing option, and the program after `--` is python3 itself.
set -u
TAG=${1:-r06}
BURN=${2:-1000}      # burn-in of the n = 1e6 chain (the benchmark's default); n = 1e7 burns in for BURN7
BURN7=${3:-300}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT" "$ROOT/profiles"
cd /tmp && export TMPDIR=/tmp
newest_db() { find "$1" -name '*_results.db' -printf '%T@ %p\n' 2>/dev/null | sort -n | tail -1 | cut -d' ' -f2-; }
TRAFFIC="{\"_note\": \"$TAG: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python3 bench.py --n N --burn-in $BURN (n = 1e6) / $BURN7 (n = 1e7) --steps 2 --warmup 1 --profile-sweeps 1: the chain past its burn-in, like the headline, summarised by tools/pmc_traffic.py; bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md)\""
for N in 1000000 10000000; do
  # (every pass — kernel trace AND counters — runs the chain the benchmark times: past its burn-in.  A cold chain accepts more moves, a stationary one
  # carries more bins per step: until round 5 the counter passes and n = 1e7 used a 30-iteration burn-in)
  TRACEBURN=$BURN7; [ "$N" = "1000000" ] && TRACEBURN=$BURN
  COMMON="--n $N --no-cpu-baseline --no-extra-configs --no-hmc-mode1 --target-n 0 --mode-iters 0 --burn-in $TRACEBURN"
  rm -rf "$OUT/prof_n$N" "$OUT/pmc_fetch_n$N" "$OUT/pmc_write_n$N"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_n$N" -- python3 "$ROOT/bench.py" $COMMON --steps 20 --warmup 5 > "$OUT/bench_prof_n$N.log" 2>&1
  python3 "$ROOT/tools/rocpd_summary.py" "$(newest_db "$OUT/prof_n$N")" "$ROOT/profiles/${TAG}_rocprofv3_prof_n$N.txt"
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch_n$N" -- python3 "$ROOT/bench.py" $COMMON --steps 2 --warmup 1 --profile-sweeps 1 > "$OUT/pmc_fetch_n$N.log" 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write_n$N" -- python3 "$ROOT/bench.py" $COMMON --steps 2 --warmup 1 --profile-sweeps 1 > "$OUT/pmc_write_n$N.log" 2>&1
  F=$(newest_db "$OUT/pmc_fetch_n$N"); W=$(newest_db "$OUT/pmc_write_n$N")
  python3 "$ROOT/tools/rocpd_summary.py" "$F" "$ROOT/profiles/${TAG}_rocprofv3_pmc_fetch_n$N.txt"
  python3 "$ROOT/tools/rocpd_summary.py" "$W" "$ROOT/profiles/${TAG}_rocprofv3_pmc_write_n$N.txt"
  TRAFFIC="$TRAFFIC, \"$N\": $(python3 "$ROOT/tools/pmc_traffic.py" "$F" "$W" $N 200)"
done
echo "$TRAFFIC}" > "$ROOT/profiles/pmc_traffic.json"
# the fused launch per tree (k_step: the automatic choice of rounds 2-3, now the hand-over target of the persistent sweep): per-kernel durations
rm -rf "$OUT/prof_fused"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_fused" -- python3 "$ROOT/bench.py" --n 1000000 --no-cpu-baseline --no-extra-configs --target-n 0 --no-hmc-mode1 --mode-iters 0 --burn-in 150 --steps 20 --warmup 5 --tree-path fused > "$OUT/bench_prof_fused.log" 2>&1
python3 "$ROOT/tools/rocpd_summary.py" "$(newest_db "$OUT/prof_fused")" "$ROOT/profiles/${TAG}_rocprofv3_prof_fused_n1000000.txt"
# the box's repository copy is scratch: hand the summaries back through gpurun_out/
mkdir -p "$OUT/profiles" && cp "$ROOT"/profiles/${TAG}_rocprofv3_* "$ROOT/profiles/pmc_traffic.json" "$OUT/profiles/"
# (the raw rocprofv3 databases stay on the box: gpurun hands back at most 64 MiB)
rm -rf "$OUT"/prof_n* "$OUT"/pmc_fetch_n* "$OUT"/pmc_write_n* "$OUT/prof_fused"
