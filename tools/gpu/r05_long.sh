O=gpurun_out/r05long; mkdir -p $O
timeout 900 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --burn-in 3000 --steps 6000 --warmup 20 > $O/long_1e6.json 2> $O/long_1e6.err; echo "rc $?"; python -c "
import json; d=json.load(open('$O/long_1e6.json')); r=d['roofline']; print('n=1e6: 9000 iterations', d['value'], r['avg_launch_us'], r['persistent_sweeps'], r['sweeps_handed_over_to_k_step'])"
timeout 900 python bench.py --n 100000 --p 10 --trees 200 --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --burn-in 5000 --steps 20000 --warmup 20 > $O/long_1e5.json 2> $O/long_1e5.err; echo "rc $?"; python -c "
import json; d=json.load(open('$O/long_1e5.json')); r=d['roofline']; print('n=1e5: 25000 iterations', d['value'], r['avg_launch_us'], r['persistent_sweeps'], r['sweeps_handed_over_to_k_step'])"
