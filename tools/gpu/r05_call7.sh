set -u
O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_stream.py tests/test_gpu_teacher_forced.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -n 5 $O/pytest_gpu.log
python tools/step_probe.py --iters 30 --sweeps 3 > $O/probe_product_1.json 2> $O/probe_product_1.err; cut -c1-300 $O/probe_product_1.json
python tools/step_probe.py --iters 30 --sweeps 3 > $O/probe_product_2.json 2> $O/probe_product_2.err; cut -c1-300 $O/probe_product_2.json
S4B_HOST_TIMING=1 timeout 600 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_stationary.json 2> $O/bench_stationary.err
grep "S4B host" $O/bench_stationary.err | sed -n 3p
python - <<'PY'
import json
r=json.load(open('gpurun_out/r05g/bench_stationary.json')); c=r['config']
print(round(r['value'],1), 'ms', round(r['ms_per_step'],3), 'lf', c['n_leapfrog_per_step'], c.get('stationarity',{}).get('stationary'), 'sweep_wall', r['roofline']['sweep_wall_us'], r['roofline']['avg_launch_us'])
PY
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_sweeptiming.so timeout 600 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 50 --warmup 5 --profile-sweeps 3 > $O/bench_sweeptiming.json 2> $O/bench_sweeptiming.err
grep SWEEP $O/bench_sweeptiming.err | cut -c1-900
python tools/step_probe.py --n 747 --p 26 --trees 75 --iters 200 --sweeps 20 > $O/probe_solo.json 2> $O/probe_solo.err; cut -c1-300 $O/probe_solo.json
