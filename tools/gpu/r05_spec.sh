set -u
O=gpurun_out/r05spec; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_teacher_forced.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 4 $O/pytest.log
S4B_HOST_TIMING=1 timeout 600 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err; cut -c1-700 $O/bench.json; grep -i "host timing\|sweep" $O/bench.err | head -5 | cut -c1-400
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_sweeptiming.so timeout 600 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 50 --warmup 5 --profile-sweeps 3 > $O/bench_sweeptiming.json 2> $O/bench_sweeptiming.err
grep "SWEEP decide\|SWEEP timeline\|SWEEP" $O/bench_sweeptiming.err | cut -c1-1200
