#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --workload hall_1080p_256spp --spp 64 --sample-split 1 --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>gpurun_out/j20.err | tail -1 > gpurun_out/j20.json
TERRA_AMD_TIMING=1 python bench.py --workload hall_1080p_256spp --spp 16 --sample-split 1 --steps 2 --warmup 1 --no-cpu-baseline --no-workloads 2>&1 | grep -v "^{" | tail -20
