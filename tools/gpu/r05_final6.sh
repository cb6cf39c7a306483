set -u
O=gpurun_out/r05final6; mkdir -p $O
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; cut -c1-200 $O/bench_default.json
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1; tail -n 2 $O/pytest.log
