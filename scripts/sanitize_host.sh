#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer build of the HOST-side native code -- the SVMrank
# parser (csrc/svmrank_parser.cpp, include/ltr_io.h) and the C oracle (oracle/ltr_oracle.c) -- and the
# CPU tests that drive them, run against those builds.  No GPU needed:
#     scripts/sanitize_host.sh [extra pytest args]
# (The device code has no counterpart here: its checks are the parity tests and the NaN / garbage-in-
# padding / timeout tests of the GPU tier.)
set -euo pipefail
cd "$(dirname "$0")/.."
out=build/sanitize
mkdir -p "$out"
flags="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined"
g++ $flags -std=c++17 -fPIC -shared -pthread -Wall -I include -o "$out/libltr_io.so" pytorchltr_amd/csrc/svmrank_parser.cpp
gcc $flags -std=c99 -fPIC -shared -o "$out/libltr_oracle.so" oracle/ltr_oracle.c -lm
asan="$(gcc -print-file-name=libasan.so)"
ubsan="$(gcc -print-file-name=libubsan.so)"
# the interpreter itself is not instrumented: preload the runtimes, leaks of CPython are not ours
LD_PRELOAD="$asan:$ubsan" ASAN_OPTIONS="detect_leaks=0:abort_on_error=1" UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1" \
LTR_IO_LIB="$PWD/$out/libltr_io.so" LTR_ORACLE_LIB="$PWD/$out/libltr_oracle.so" \
python -m pytest tests/test_svmrank_parser.py tests/test_oracle_golden.py -q -m "not gpu" "$@"
