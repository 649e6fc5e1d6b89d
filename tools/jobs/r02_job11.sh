#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r02_gputests.log 2>&1; echo "gpu tests rc $?"; tail -2 gpurun_out/r02_gputests.log
echo "== hall auto: 64-node LDS prefix (base) vs none (p0: original creation order, nothing staged)"; bash tools/ab_variants.sh "--workload hall_1080p_256spp --sample-split 1 --steps 2 --warmup 1 --no-workloads" base p0
echo "== spheres"; bash tools/ab_variants.sh "--workload spheres_1080p_1024spp --steps 1 --warmup 1 --no-workloads" base p0
