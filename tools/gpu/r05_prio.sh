set -u
O=gpurun_out/r05prio; mkdir -p $O
for v in ${VARIANTS:-base a1 a2 a3 base a1}; do
  L=$PWD/stan4bart_amd/csrc/libs4b_$v.so; [ "$v" = "base" ] && L=$PWD/stan4bart_amd/csrc/libs4b.so
  S4B_LIB_PATH=$L timeout 300 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 400 --warmup 20 > $O/bench_$v.json 2> $O/bench_$v.err
  echo "$v $(python -c "import json,sys; d=json.load(open('$O/bench_$v.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d.get('warmup_phase_iters_per_sec'))")"
done
