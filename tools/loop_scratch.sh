#!/bin/bash
# scratch (spill) instructions, full vmcnt(0) waits and instruction counts by region of one chain_i8w_kernel instance (regions
# split at s_barrier; the steady-state iteration is the largest one).  After tools/cc_i8w.sh.
f=/tmp/asm/kernels_i8w-hip-amdgcn-amd-amdhsa-gfx950.s
k=${1:-_Z16chain_i8w_kernelILi2ELi4ELi8ELi5ELi0ELb0EEv9DecimArgs}
awk -v k="$k" '$0 ~ "^"k":" {p=1} p{print} /^\.Lfunc_end/{if(p){exit}}' $f > /tmp/asm/w.s
awk '/s_barrier/{r++} /^\t[a-z]/{n[r]++} /^\ts_/{sc[r]++} /scratch_/{s[r]++} /s_waitcnt vmcnt\(0\)/{v[r]++} /v_mfma/{m[r]++} END{for(i=0;i<=r;i++) printf "region %d: instructions %d (scalar %d) mfma %d scratch %d vmcnt0 %d\n", i, n[i], sc[i], m[i], s[i], v[i]}' /tmp/asm/w.s
