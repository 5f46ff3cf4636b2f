#!/bin/bash
# compile kernels_i8w.hip alone with temporaries kept in /tmp/asm, print register/spill summary.
#   tools/cc_i8w.sh [name filter]            every instantiation (minutes)
#   PROBE=1 tools/cc_i8w.sh                  the headline instantiation alone (PROBE_ARGS="SDRT_S16, 8, 8, SDRT_I8_DIGITS, 0, false" picks another)
mkdir -p /tmp/asm; cd "$(dirname "$0")/../sdrterm_amd/csrc" || exit 1
extra=""
[ -n "$PROBE" ] && extra="-DSDRT_I8W_PROBE"
[ -n "$PROBE_ARGS" ] && extra="$extra -DSDRT_I8W_PROBE_ARGS=$PROBE_ARGS"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-const-variable $extra $EXTRA -c kernels_i8w.hip -o /tmp/asm/kernels_i8w.o -save-temps=obj 2>&1 | grep -v "warning: argument unused" | head
f=/tmp/asm/kernels_i8w-hip-amdgcn-amd-amdhsa-gfx950.s
grep -n "\.name:\|vgpr_spill\|\.vgpr_count" $f | paste - - - | awk '{print $3, $5, $6, $8, $9}' | grep "${1:-chain}"
