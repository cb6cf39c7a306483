O=gpurun_out/r05h; mkdir -p $O
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_sweeptiming.so timeout 600 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 50 --warmup 5 --profile-sweeps 3 > $O/bench_sweeptiming.json 2> $O/bench_sweeptiming.err
grep "SWEEP decide\|SWEEP timeline" $O/bench_sweeptiming.err | cut -c1-900
