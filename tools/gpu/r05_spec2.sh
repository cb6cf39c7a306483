set -u
O=gpurun_out/r05spec; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_teacher_forced.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 4 $O/pytest.log
VARIANTS="${BV:-pA pB pC pE}" bash tools/gpu/r05_prio.sh
VARIANTS="${TV:-stB}" bash tools/gpu/r05_tl.sh
