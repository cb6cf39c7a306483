set -u
O=gpurun_out/r05final5; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -n 3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; cut -c1-300 $O/bench_default.json
