O=gpurun_out/r05fuzz; mkdir -p $O
timeout 700 python tools/fuzz_range.py 160 8160 persistent > $O/fuzz_persistent.log 2>&1; echo "rc $?" >> $O/fuzz_persistent.log; tail -n 2 $O/fuzz_persistent.log | cut -c1-600
timeout 400 python tools/fuzz_range.py 160 3160 stream > $O/fuzz_stream.log 2>&1; echo "rc $?" >> $O/fuzz_stream.log; tail -n 2 $O/fuzz_stream.log | cut -c1-600
timeout 300 python tools/fuzz_range.py 160 2160 fused > $O/fuzz_fused.log 2>&1; echo "rc $?" >> $O/fuzz_fused.log; tail -n 2 $O/fuzz_fused.log | cut -c1-600
timeout 600 python tools/fuzz_large.py 0 250 persistent > $O/fuzz_large_persistent.log 2>&1; echo "rc $?" >> $O/fuzz_large_persistent.log; tail -n 2 $O/fuzz_large_persistent.log | cut -c1-600
