set -u
O=gpurun_out/r05spec; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_teacher_forced.py tests/test_gpu_stream.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 3 $O/pytest.log
timeout 600 python bench.py --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 200 --warmup 20 > $O/bench_x.json 2> $O/bench_x.err
python -c "
import json; d=json.load(open('$O/bench_x.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d.get('warmup_phase_iters_per_sec'))
for k,v in d['extra_configs'].items(): print(k, v.get('gpu_iters_per_sec'))"
