set -u
O=gpurun_out/r05f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -n 5 $O/pytest_gpu.log
timeout 600 python tools/step_probe.py --n 10000000 --iters 4 --sweeps 2 --path stream > $O/probe_n1e7_stream.json 2> $O/probe_n1e7_stream.err; cut -c1-330 $O/probe_n1e7_stream.json
timeout 600 python tools/step_probe.py --n 2000000 --iters 6 --sweeps 3 --path stream > $O/probe_n2e6_stream.json 2> $O/probe_n2e6_stream.err; cut -c1-330 $O/probe_n2e6_stream.json
