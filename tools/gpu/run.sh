#!/bin/bash
# ONE parameterised driver for the GPU box (replaces the per-experiment r05_*.sh scripts): run from the repository root through gpurun, e.g.
#     gpurun --timeout 1500 -- 'bash tools/gpu/run.sh r06a tests bench'
#     gpurun --timeout 1200 -- 'bash tools/gpu/run.sh r06b ab libs4b_nolin.so'
# Steps (any subset, in the order given), everything under gpurun_out/<tag>/:
#   tests        the whole GPU suite (per-test timeout: a launch that never returns must not take the box with it)
#   quick        parity + fuzz + configs only
#   bench        python bench.py (the driver's line)               -> bench_default.json
#   benchlite    the headline only (no CPU leg, no extra configs)  -> bench_lite.json
#   probe        tools/step_probe.py --iters 300 at n = 1e6 (stationary sweep)   -> probe.json
#   ab <lib>     A/B/A/B of the product library against stan4bart_amd/csrc/<lib> with step_probe --iters 300
#   wgt          the per-workgroup stamps of the exchange (make wgt build)       -> wgt.txt
#   tl           the instrumented timeline (make sweeptiming build)              -> timeline.txt
#   profile      tools/profile_round.sh <tag>
#   large A B    tools/fuzz_large.py A B
#   libs L1 L2 .. --   step_probe --iters 300 once per variant library
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p "$O"
PT="--timeout 300 --timeout-method thread"
while [ $# -gt 0 ]; do
  step=$1; shift
  case $step in
    tests) timeout 2400 python -m pytest tests -m gpu -x -q $PT > "$O/pytest.log" 2>&1; echo "tests rc $?"; tail -n 3 "$O/pytest.log" ;;
    quick) timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_large.py tests/test_gpu_busy.py -m gpu -x -q $PT > "$O/pytest_quick.log" 2>&1; echo "quick rc $?"; tail -n 3 "$O/pytest_quick.log" ;;
    bench) timeout 1500 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "bench rc $?"; cut -c1-300 "$O/bench_default.json"; tail -n 3 "$O/bench_default.err" ;;
    benchlite) timeout 900 python bench.py --no-cpu-baseline --no-extra-configs --target-n 0 --no-hmc-mode1 --mode-iters 0 > "$O/bench_lite.json" 2> "$O/bench_lite.err"; echo "benchlite rc $?"; cut -c1-300 "$O/bench_lite.json" ;;
    probe) timeout 600 python tools/step_probe.py --iters 300 --sweeps 5 > "$O/probe.json" 2> "$O/probe.err"; cat "$O/probe.json" ;;
    ab) lib=$1; shift
        for r in 1 2; do
          timeout 600 python tools/step_probe.py --iters 300 --sweeps 5 > "$O/ab_product_$r.json" 2> "$O/ab_product_$r.err"
          S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/$lib timeout 600 python tools/step_probe.py --iters 300 --sweeps 5 > "$O/ab_other_$r.json" 2> "$O/ab_other_$r.err"
        done
        for f in "$O"/ab_*.json; do echo "$f $(cut -c1-400 "$f")"; done ;;
    libs) # probe a list of variant libraries once each:  libs libs4b_p1.so libs4b_p2.so --
        while [ $# -gt 0 ] && [ "$1" != "--" ]; do lib=$1; shift
          S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/$lib timeout 600 python tools/step_probe.py --iters 300 --sweeps 5 > "$O/probe_$lib.json" 2> "$O/probe_$lib.err"
          echo "$lib $(cut -c1-60 "$O/probe_$lib.json") $(grep -o '"per_tree_wall_us": [0-9.]*' "$O/probe_$lib.json")"; done
        [ $# -gt 0 ] && shift ;;
    wgt) S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_wgt.so timeout 600 python tools/step_probe.py --iters 300 --sweeps 5 > "$O/wgt.json" 2> "$O/wgt.txt"; cat "$O/wgt.txt" ;;
    tl) S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_sweeptiming.so timeout 600 python tools/step_probe.py --iters 300 --sweeps 5 > "$O/timeline.json" 2> "$O/timeline.txt"; cat "$O/timeline.txt" ;;
    profile) bash tools/profile_round.sh "$TAG" ;;
    large) lo=$1; hi=$2; shift 2; timeout 2400 python tools/fuzz_large.py "$lo" "$hi" > "$O/fuzz_large_${lo}_${hi}.log" 2>&1; tail -n 2 "$O/fuzz_large_${lo}_${hi}.log" ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
