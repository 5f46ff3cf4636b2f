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
