#!/bin/bash
# compile kernels_i8w.hip alone with temporaries kept, print register/spill summary
mkdir -p /tmp/asm; cd /root/repo/sdrterm_amd/csrc || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-const-variable -c kernels_i8w.hip -o /tmp/asm/kernels_i8w.o -save-temps=obj 2>&1 | grep -v "warning: argument unused" | head
f=/tmp/asm/kernels_i8w-hip-amdgcn-amd-amdhsa-gfx950.s
grep -n "\.name:\|vgpr_spill\|\.vgpr_count" $f | paste - - - | awk '{print $3, $5, $6, $8, $9}' | grep "${1:-chain}"
