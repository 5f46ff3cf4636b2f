#!/bin/bash
# scratch (spill) instructions and vmcnt waits by region of one chain_i8w_kernel instance (regions split at s_barrier)
f=/tmp/asm/kernels_i8w-hip-amdgcn-amd-amdhsa-gfx950.s
awk -v k="$1" '$0 ~ "^"k":" {p=1} p{print} /^\.Lfunc_end/{if(p){exit}}' $f > /tmp/asm/w.s
awk '/s_barrier/{r++} /scratch_/{s[r]++} /s_waitcnt vmcnt\(0\)/{v[r]++} /v_mfma/{m[r]++} END{for(i=0;i<=r;i++) printf "region %d: mfma %d scratch %d vmcnt0 %d\n", i, m[i], s[i], v[i]}' /tmp/asm/w.s
