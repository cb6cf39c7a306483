O=gpurun_out/r05ab; mkdir -p $O
for V in "" _poll2 "" _poll2; do
  S4B_HOST_TIMING=1 S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/libs4b$V.so timeout 600 python bench.py --no-extra-configs --target-n 0 --no-cpu-baseline --no-hmc-mode1 --mode-iters 0 --steps 200 --warmup 20 > $O/bench$V.json 2> $O/bench$V.err
  echo "variant '$V'"; grep "S4B host" $O/bench$V.err | sed -n 3p
  python - "$O/bench$V.json" <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); print(round(r['value'],1), 'it/s', round(r['ms_per_step'],3), 'ms; sweep wall', r['roofline']['sweep_wall_us'])
PY
done
