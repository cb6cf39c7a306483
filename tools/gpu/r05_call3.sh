set -u
O=gpurun_out/r05c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -n 4 $O/pytest_gpu.log
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_sweeptiming.so timeout 600 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 50 --warmup 5 --profile-sweeps 3 > $O/bench_sweeptiming.json 2> $O/bench_sweeptiming.err
grep SWEEP $O/bench_sweeptiming.err | cut -c1-900
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_sweeptiming.so timeout 600 python bench.py --burn-in 150 --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 20 --warmup 5 --profile-sweeps 3 > $O/bench_sweeptiming_b150.json 2> $O/bench_sweeptiming_b150.err
grep SWEEP $O/bench_sweeptiming_b150.err | cut -c1-900
