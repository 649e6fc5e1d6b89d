#!/bin/bash
# VGPR / SGPR / scratch of every render kernel variant and static instruction counts of the headline one (dev tool, no GPU needed)
# usage: tools/kernel_resources.sh [extra -D flags]
set -e
D=/tmp/terra_isa; rm -rf $D; mkdir -p $D; cd $D
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -fno-slp-vectorize -I$OLDPWD/include "$@" --save-temps -c $OLDPWD/terra_amd/csrc/render_kernels.hip -o rk.o 2>/dev/null
S=render_kernels-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.(name|vgpr_count|sgpr_count|private_segment_fixed_size):" $S | paste - - - - | sed -E 's/\s+/ /g' | grep terra_render_kernel | awk '{print $2, "priv", $4, "sgpr", $6, "vgpr", $8}' | sed 's/_Z19terra_render_kernelILi//; s/EEv15DevRenderParams//; s/ELi/,/g' | sort | awk '{printf "<I,COUNT,MODE,KINDS>=%-12s %s %s %s %s %s %s\n",$1,$2,$3,$4,$5,$6,$7}' | grep -E "=0,0,1,1 |=1,0,1,1 |=0,0,0,1 |=0,0,2,1 |=1,0,2,1 |=0,0,2,63 |=0,0,1,3 |=0,2,1,1 " 
