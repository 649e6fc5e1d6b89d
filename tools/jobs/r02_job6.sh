#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 250 tools/ubench/valu_rates > gpurun_out/r02_valu_rates2.log 2>&1; grep "waves/SIMD 4" gpurun_out/r02_valu_rates2.log | grep "pk_\|v_mov\|v_sub\|v_max\|v_add_f32\|v_fma_f32"
bash tools/ab_variants.sh "--steps 3 --warmup 1 --no-workloads" base noslp
bash tools/ab_variants.sh "--steps 2 --warmup 1 --no-workloads --integrator direct" base noslp
