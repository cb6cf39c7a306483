for r in 1 2 3 4 5 6 7 8; do
  timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config3_full_size" --timeout 300 2>&1 | tail -1 | sed "s/^/run $r: /"
done
for r in 1 2 3; do
  S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b_nolin.so timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config3_full_size" --timeout 300 2>&1 | tail -1 | sed "s/^/nolin run $r: /"
done
