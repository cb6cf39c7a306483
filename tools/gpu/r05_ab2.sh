set -u
O=gpurun_out/r05ab; mkdir -p $O
for i in 1 2 3; do
  timeout 300 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 400 --warmup 20 > $O/bench_$i.json 2> $O/bench_$i.err
  python -c "import json,sys; d=json.load(open('$O/bench_$i.json')); print('base', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), round(d.get('warmup_phase_iters_per_sec'),1))"
done
VARIANTS="wgt" bash tools/gpu/r05_tl.sh
