#!/bin/bash
cd $GRAFT_REPO_ROOT
TERRA_AMD_TIMING=1 timeout -k 10 600 python tools/scale_triangles.py > gpurun_out/r02_scale_triangles.log 2>&1; tail -30 gpurun_out/r02_scale_triangles.log
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu --durations=6 2>&1 | tail -12
