set -u
mkdir -p gpurun_out/r05a
O=gpurun_out/r05a
python tools/step_probe.py --iters 30 --sweeps 3 > $O/probe_product_1.json 2> $O/probe_product_1.err
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_mockstats.so python tools/step_probe.py --iters 30 --sweeps 3 > $O/probe_mock_1.json 2> $O/probe_mock_1.err
python tools/step_probe.py --iters 30 --sweeps 3 > $O/probe_product_2.json 2> $O/probe_product_2.err
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_mockstats.so python tools/step_probe.py --iters 30 --sweeps 3 > $O/probe_mock_2.json 2> $O/probe_mock_2.err
S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_sweeptiming.so python tools/step_probe.py --iters 30 --sweeps 3 > $O/probe_sweeptiming.json 2> $O/probe_sweeptiming.err
S4B_HOST_TIMING=1 timeout 900 python bench.py --burn-in 1000 --no-extra-configs --target-n 0 --no-cpu-baseline > $O/bench_burn1000.json 2> $O/bench_burn1000.err
tail -3 $O/*.json | cut -c1-1500
