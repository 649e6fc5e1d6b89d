#!/bin/bash
cd $GRAFT_REPO_ROOT
for wl in "hall_1080p_256spp --sample-split 1" "hall_1080p_256spp --sample-split 2" "hall_1080p_256spp --sample-split 4" "spheres_1080p_1024spp --sample-split 8" "spheres_1080p_1024spp --sample-split 4" "spheres_1080p_1024spp --sample-split 2"; do
    timeout -k 10 200 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', 'ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'Msamples/s', d['value'])"
done
