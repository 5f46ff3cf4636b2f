#!/bin/bash
# register / spill summary of every kernel in a -save-temps assembly file:  tools/regs.sh /tmp/asm/kernels_sq-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/\.agpr_count:/{ag=$2} /\.name:/{nm=$2} /\.sgpr_spill_count:/{ss=$2} /\.vgpr_count:/{vc=$2} /\.vgpr_spill_count:/{print nm, "vgpr", vc, "agpr", ag, "vgpr_spill", $2, "sgpr_spill", ss}' "$1"
