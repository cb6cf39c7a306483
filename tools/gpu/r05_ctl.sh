S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_timing.so timeout 600 python tools/step_probe.py --n 10000000 --p 50 --trees 200 --sweeps 3 --iters 40 --path two-kernel 2>&1 | tail -25 | cut -c1-700
