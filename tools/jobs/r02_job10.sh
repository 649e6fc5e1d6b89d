#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== fast-tree LDS prefix (hall 1080p 256 spp, auto)"; bash tools/ab_variants.sh "--workload hall_1080p_256spp --sample-split 1 --steps 2 --warmup 1 --no-workloads" base p64 p256 p1024
echo "== spheres (fast tree, generic kinds)"; bash tools/ab_variants.sh "--workload spheres_1080p_1024spp --steps 1 --warmup 1 --no-workloads" base p64 p256 p1024
echo "== headline occupancy"; bash tools/ab_variants.sh "--steps 3 --warmup 1 --no-workloads" base w4 w6
echo "== direct occupancy"; bash tools/ab_variants.sh "--steps 2 --warmup 1 --no-workloads --integrator direct" base l3 l5
