O=gpurun_out/r05solo; mkdir -p $O
for v in base ss0 ss4 base ss0 ss4; do
L=$PWD/stan4bart_amd/csrc/libs4b_$v.so; [ "$v" = "base" ] && L=$PWD/stan4bart_amd/csrc/libs4b.so
a=$(S4B_LIB_PATH=$L timeout 300 python tools/step_probe.py --n 747 --p 25 --trees 75 --sweeps 40 --iters 400 --path persistent 2>&1 | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['per_tree_wall_us'],3))")
S4B_LIB_PATH=$L timeout 600 python bench.py --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --burn-in 30 --steps 5 --warmup 2 > $O/b_$v.json 2> $O/b_$v.err
echo "$v solo step us $a  $(python -c "
import json; d=json.load(open('$O/b_$v.json')); print({k:round(v.get('gpu_iters_per_sec') or 0,1) for k,v in d['extra_configs'].items()})")"
done
