#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r02_j22_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r02_j22_tests.log
for v in base; do
  for wl in "hall_1080p_256spp --spp 64 --sample-split 1" "spheres_1080p_1024spp --spp 128 --sample-split 8" "hall_1080p_256spp --spp 32 --sample-split 1 --integrator direct" "hall_1080p_256spp --spp 32 --sample-split 1 --integrator mis"; do
    timeout -k 10 200 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $wl', 'kernel_ms', d['roofline']['kernel_ms'], 'Msamples/s', d['value'])"
  done
done
