set -u
O=gpurun_out/r05d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_stream.py tests/test_shim_exec.py -m gpu -x -q > $O/pytest_stream.log 2>&1; echo "pytest rc $?" >> $O/pytest_stream.log
tail -n 6 $O/pytest_stream.log
for P in stream two-kernel; do
  timeout 600 python tools/step_probe.py --n 10000000 --iters 4 --sweeps 2 --path $P > $O/probe_n1e7_$P.json 2> $O/probe_n1e7_$P.err; cat $O/probe_n1e7_$P.json | cut -c1-400
done
for P in stream fused; do
  timeout 600 python tools/step_probe.py --n 2000000 --iters 6 --sweeps 3 --path $P > $O/probe_n2e6_$P.json 2> $O/probe_n2e6_$P.err; cat $O/probe_n2e6_$P.json | cut -c1-400
done
timeout 600 python tools/step_probe.py --n 1000000 --iters 30 --sweeps 3 --path stream > $O/probe_n1e6_stream.json 2> $O/probe_n1e6_stream.err; cat $O/probe_n1e6_stream.json | cut -c1-400
