set -u
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_shim_exec.py -m gpu -x -q > $O/pytest_stream.log 2>&1; echo "pytest rc $?" >> $O/pytest_stream.log
tail -n 4 $O/pytest_stream.log
for V in ""; do
  L=$PWD/stan4bart_amd/csrc/libs4b$V.so
  S4B_LIB_PATH=$L timeout 600 python tools/step_probe.py --n 10000000 --iters 4 --sweeps 2 --path stream > $O/probe_n1e7_stream$V.json 2> $O/probe_n1e7_stream$V.err; echo "n1e7 stream$V"; cut -c1-330 $O/probe_n1e7_stream$V.json
  S4B_LIB_PATH=$L timeout 600 python tools/step_probe.py --n 2000000 --iters 6 --sweeps 3 --path stream > $O/probe_n2e6_stream$V.json 2> $O/probe_n2e6_stream$V.err; echo "n2e6 stream$V"; cut -c1-330 $O/probe_n2e6_stream$V.json
done
