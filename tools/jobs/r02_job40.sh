#!/bin/bash
cd $GRAFT_REPO_ROOT
for wl in "hall_1080p_256spp --spp 4 --sample-split 1" "hall_1080p_256spp --spp 64 --sample-split 1" "spheres_1080p_1024spp --spp 128 --sample-split 8"; do
    TERRA_AMD_TIMING=1 timeout -k 5 60 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>gpurun_out/j40.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['counters_per_launch']; print('$wl', 'kernel_ms', d['roofline']['kernel_ms'], 'Msamples/s', d['value'], 'nodes/ray %.2f tris/ray %.2f' % (c['nodes']/c['rays'], c['tri_tests']/c['rays']))" || { echo "FAILED $wl"; grep -i "wide\|error" gpurun_out/j40.err | tail -3; break; }
    grep "4-wide" gpurun_out/j40.err | tail -1
done
