#!/bin/bash
# The plain-C oracle under AddressSanitizer + UndefinedBehaviorSanitizer against the golden vectors (CPU only: GPU sanitizer
# runs are not available on this pool).  Builds into /tmp, swaps the library in for the run, puts the regular one back.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/san
gcc -O1 -g -fPIC -std=gnu11 -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o /tmp/san/liboracle.so "$ROOT/oracle/chain.c" -lm
make -s -C "$ROOT/oracle"
cp "$ROOT/oracle/_build/liboracle.so" /tmp/san/liboracle.orig.so
trap 'cp /tmp/san/liboracle.orig.so "$ROOT/oracle/_build/liboracle.so"' EXIT
cp /tmp/san/liboracle.so "$ROOT/oracle/_build/liboracle.so"
cd "$ROOT" && LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_oracle_golden.py -q -x
# ... and the product's host-side table builders (operators.cpp, plain C++): built with the sanitizers by g++, linked with the
# library's other objects, driven by the CPU tests of the tables and by the numpy emulation of the wide kernel's arithmetic
( cd "$ROOT/sdrterm_amd/csrc" && g++ -O1 -g -fPIC -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -c operators.cpp -o /tmp/san/operators.o \
  && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/san/libsdrterm_hip_san.so /tmp/san/operators.o $(ls _obj/*.o | grep -v operators.o) -lm -ldl -fsanitize=address,undefined )
export SDRTERM_HIP_LIB=/tmp/san/libsdrterm_hip_san.so LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0
python -m pytest tests/test_post8_tables.py tests/test_host_logic.py -q -x
python tools/emulate_wide.py | tail -1
