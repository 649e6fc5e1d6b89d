#!/bin/bash
cd $GRAFT_REPO_ROOT
TERRA_AMD_TIMING=1 timeout -k 10 600 python tools/scale_triangles.py > gpurun_out/r02_scale_triangles.log 2>&1; grep -v "thread" gpurun_out/r02_scale_triangles.log | tail -32
TERRA_AMD_BUILD_THREADS=1 TERRA_AMD_TIMING=1 timeout -k 10 600 python tools/scale_triangles.py 2>&1 | grep "pass\|host)"
