set -u
O=gpurun_out/r05ab; mkdir -p $O
for v in ${VARIANTS:-base cs4 cs12 base cs4 cs12}; do
  L=$PWD/stan4bart_amd/csrc/libs4b_$v.so; [ "$v" = "base" ] && L=$PWD/stan4bart_amd/csrc/libs4b.so
  S4B_LIB_PATH=$L timeout 300 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 400 --warmup 20 > $O/bench_$v.json 2> $O/bench_$v.err
  echo "$v $(python -c "import json,sys; d=json.load(open('$O/bench_$v.json')); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), round(d.get('warmup_phase_iters_per_sec'),1))")"
done
