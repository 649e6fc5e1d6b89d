#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 250 tools/ubench/valu_rates > gpurun_out/r02_valu_rates3.log 2>&1; grep "waves/SIMD 4" gpurun_out/r02_valu_rates3.log | awk '{print $1, $2, $9, $13}' | column -t
bash tools/ab_variants.sh "--steps 3 --warmup 1 --no-workloads" base o2
python -m pytest tests -x -q -m gpu > gpurun_out/r02_gputests.log 2>&1; echo "gpu tests rc $?"; tail -2 gpurun_out/r02_gputests.log
