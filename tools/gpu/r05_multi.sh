O=gpurun_out/r05multi; mkdir -p $O
timeout 900 python tools/multi_chain_probe.py > $O/hint.log 2>&1; cat $O/hint.log | tail -6
timeout 900 python tools/multi_chain_probe.py --no-hint > $O/nohint.log 2>&1; cat $O/nohint.log | tail -6
