set -u
bash tools/profile_round.sh r05 > gpurun_out/profile_round.log 2>&1; echo "profile rc $?"; ls gpurun_out/profiles/ | head -20
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_sweeptiming.so timeout 600 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 50 --warmup 5 --profile-sweeps 3 > gpurun_out/r05_timeline.json 2> gpurun_out/r05_timeline.err; grep -c SWEEP gpurun_out/r05_timeline.err
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_wgt.so timeout 600 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 50 --warmup 5 --profile-sweeps 3 > gpurun_out/r05_wgt.json 2> gpurun_out/r05_wgt.err; grep SWEEP gpurun_out/r05_wgt.err | cut -c1-300
