set -u
O=gpurun_out/r05prio; mkdir -p $O
for v in ${VARIANTS:-pB pC pBl pB}; do
  S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_$v.so timeout 300 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 200 --warmup 20 > $O/bench_$v.json 2> $O/bench_$v.err
  echo "$v $(python -c "import json,sys; d=json.load(open('$O/bench_$v.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'] if 'roofline' in d else '')")"
done
