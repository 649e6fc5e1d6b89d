#!/bin/bash
cd $GRAFT_REPO_ROOT
export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_ps.so
python tools/phase_stats.py --scene hall --spp 32 --split 1 --tree 2 2>&1 | tail -6
