set -u
O=gpurun_out/r05final4; mkdir -p $O
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; echo "profile rc $?"
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; cut -c1-300 $O/bench_default.json
timeout 600 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --burn-in 150 --steps 20 --warmup 5 > $O/bench_burn150.json 2> $O/bench_burn150.err; python -c "
import json; d=json.load(open('$O/bench_burn150.json')); print('burn-in 150 (round 4 window):', d['value'], d['roofline']['avg_launch_us'], d['config']['n_leapfrog_per_step'])"
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_sweeptiming.so timeout 600 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 50 --warmup 5 --profile-sweeps 3 > $O/timeline.json 2> $O/timeline.err
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_wgt.so timeout 600 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 50 --warmup 5 --profile-sweeps 3 > $O/wgt.json 2> $O/wgt.err; grep -c SWEEP $O/timeline.err $O/wgt.err
