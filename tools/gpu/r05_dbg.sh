python tools/path_diff_probe.py stream 100 11 1 2>&1 | grep -v amdgpu | tail -20
python tools/path_diff_probe.py stream 1003 3 1 2>&1 | grep -v amdgpu | tail -12
