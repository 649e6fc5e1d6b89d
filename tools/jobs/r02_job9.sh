#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r02_gputests.log 2>&1; echo "gpu tests rc $?"; tail -2 gpurun_out/r02_gputests.log
export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_ps.so
python tools/phase_stats.py --spp 512 --split 8 > gpurun_out/r02_phase_final_cornell.log 2>&1; tail -7 gpurun_out/r02_phase_final_cornell.log
python tools/phase_stats.py --scene hall --spp 16 --split 1 --tree 0 > gpurun_out/r02_phase_hall_ref.log 2>&1; tail -2 gpurun_out/r02_phase_hall_ref.log
python tools/phase_stats.py --scene hall --spp 64 --split 1 --tree 1 > gpurun_out/r02_phase_hall_fast.log 2>&1; tail -2 gpurun_out/r02_phase_hall_fast.log
unset TERRA_AMD_LIB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/profile_r02.py > gpurun_out/r02_profile_driver.log 2>&1; tail -5 gpurun_out/r02_profile_driver.log
