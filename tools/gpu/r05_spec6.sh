set -u
O=gpurun_out/r05spec; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_teacher_forced.py tests/test_gpu_stream.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 3 $O/pytest.log
timeout 600 python bench.py --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 200 --warmup 20 > $O/bench_x.json 2> $O/bench_x.err

for v in base base; do
L=$PWD/stan4bart_amd/csrc/libs4b_$v.so; [ "$v" = "base" ] && L=$PWD/stan4bart_amd/csrc/libs4b.so
S4B_LIB_PATH=$L timeout 600 python bench.py --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 200 --warmup 20 > $O/bench_x$v.json 2> $O/bench_x$v.err
python -c "
import json; d=json.load(open('$O/bench_x$v.json')); print('$v', d['value'], d['roofline']['avg_launch_us'], d.get('warmup_phase_iters_per_sec'), [round(v.get('gpu_iters_per_sec') or 0,1) for k,v in d['extra_configs'].items()])"
done
