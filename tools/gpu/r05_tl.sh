set -u
O=gpurun_out/r05tl; mkdir -p $O
for v in ${VARIANTS:-wgt}; do
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_$v.so timeout 600 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 50 --warmup 5 --profile-sweeps 3 > $O/bench_$v.json 2> $O/bench_$v.err
echo "== $v"; grep "SWEEP critical\|SWEEP exchange\|SWEEP profile\|SWEEP per workgroup\|SWEEP workgroup\|SWEEP timeline\|SWEEP speculation\|SWEEP decide()\|SWEEP statistics\|SWEEP steps" $O/bench_$v.err | cut -c1-1200
done
