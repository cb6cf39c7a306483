set -u
O=gpurun_out/r05final3; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -n 3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
timeout 900 python tools/multi_chain_probe.py > $O/multi_hint.log 2>&1; tail -5 $O/multi_hint.log
timeout 120 python tools/two_process_probe.py A > $O/two_A.log 2>&1 & 
timeout 120 python tools/two_process_probe.py B > $O/two_B.log 2>&1; wait; tail -n 1 $O/two_A.log $O/two_B.log
