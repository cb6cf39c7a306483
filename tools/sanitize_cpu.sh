#!/bin/bash
# The CPU test suite with the product's host logic (over the emulated device layer) and the oracle built with AddressSanitizer +
# UndefinedBehaviorSanitizer.  (GPU ASan is not available on the pool; the device code is covered by the parity suite instead.)
#   bash tools/sanitize_cpu.sh            -> prints the pytest summary and every sanitizer report (none expected)
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
FLAGS="-O1 -g -std=c++17 -fPIC -Wall -Wno-unused-function -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -shared"
make -s -C tests/emul; make -s -C oracle
cp tests/emul/_build/libs4b_emul.so "$tmp/emul.orig"; cp oracle/_build/liboracle.so "$tmp/oracle.orig"
restore() { cp "$tmp/emul.orig" tests/emul/_build/libs4b_emul.so; cp "$tmp/oracle.orig" oracle/_build/liboracle.so; rm -rf "$tmp"; }
trap restore EXIT
(cd tests/emul && g++ $FLAGS -o _build/libs4b_emul.so emul_api.cpp)
(cd oracle && g++ $FLAGS -o _build/liboracle.so gibbs_ref.cpp)
asan=$(gcc -print-file-name=libasan.so); stdcpp=$(gcc -print-file-name=libstdc++.so)
# (libstdc++ is preloaded too: the interpreter loads it late, and ASan's __cxa_throw interceptor needs the real one at start-up)
LD_PRELOAD="$asan $stdcpp" ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest tests/ -q -m "not gpu" -s --ignore=tests/test_distributed.py > "$tmp/out.txt" 2>&1 || true
grep -E "runtime error|ERROR: AddressSanitizer|SUMMARY|passed|failed" "$tmp/out.txt" | sort | uniq -c | sort -rn | head -40
