set -u
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
(timeout 120 python tools/two_process_probe.py A > $O/two_proc_A.log 2>&1 & timeout 120 python tools/two_process_probe.py B > $O/two_proc_B.log 2>&1; wait)
tail -2 $O/two_proc_A.log $O/two_proc_B.log
for KA in 0 8 64 0 8; do
  S4B_KEEPALIVE=$KA S4B_HOST_TIMING=1 timeout 600 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_ka${KA}_$RANDOM.json 2> $O/bench_ka${KA}.err
  grep "S4B host" $O/bench_ka${KA}.err | sed -n 3p
done
python tools/lf_host_probe.py stan4bart_amd/csrc/libs4b.so s4b_ > $O/lf_host_probe.txt 2>&1; cat $O/lf_host_probe.txt
for f in $O/bench_ka*.json; do python - "$f" <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); c=r['config']
print(sys.argv[1], round(r['value'],1), 'ms', round(r['ms_per_step'],3), 'lf', c['n_leapfrog_per_step'], c.get('stationarity',{}).get('stationary'), 'sweep_wall', r['roofline']['sweep_wall_us'], r['roofline']['avg_launch_us'])
PY
done
