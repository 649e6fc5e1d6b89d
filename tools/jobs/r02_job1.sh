#!/bin/bash
# round 2, GPU job 1: new full-size config tests + phase statistics of the headline kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_configs.py -x -q -m gpu --durations=10 > gpurun_out/r02_configs.log 2>&1; echo "configs rc $?" 
TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_ps.so python tools/phase_stats.py --spp 512 --split 8 > gpurun_out/r02_phase_split8.log 2>&1; echo "ps rc $?"
TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_ps.so python tools/phase_stats.py --spp 512 --split 1 > gpurun_out/r02_phase_split1.log 2>&1
tail -5 gpurun_out/r02_configs.log
