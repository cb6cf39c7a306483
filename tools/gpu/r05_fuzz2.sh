O=gpurun_out/r05fuzz2; mkdir -p $O
timeout 1200 python tools/fuzz_large.py 0 400 persistent > $O/fuzz_large_persistent.log 2>&1; echo "rc $?" >> $O/fuzz_large_persistent.log; tail -n 2 $O/fuzz_large_persistent.log | cut -c1-600
timeout 700 python tools/fuzz_range.py 160 8160 persistent > $O/fuzz_persistent.log 2>&1; echo "rc $?" >> $O/fuzz_persistent.log; tail -n 2 $O/fuzz_persistent.log | cut -c1-600
timeout 300 python tools/fuzz_range.py 160 2160 stream > $O/fuzz_stream.log 2>&1; echo "rc $?" >> $O/fuzz_stream.log; tail -n 2 $O/fuzz_stream.log | cut -c1-600
timeout 600 python tools/soak.py > $O/soak.log 2>&1; echo "rc $?" >> $O/soak.log; tail -n 3 $O/soak.log | cut -c1-400
