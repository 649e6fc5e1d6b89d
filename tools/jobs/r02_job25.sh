#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in sah1 sah15; do
  if [ "$v" = base ]; then unset TERRA_AMD_LIB; else export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_$v.so; fi
  for wl in "hall_1080p_256spp --spp 64 --sample-split 1" "spheres_1080p_1024spp --spp 128 --sample-split 8"; do
    timeout -k 10 200 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['counters_per_launch']; print('$v $wl', 'kernel_ms', d['roofline']['kernel_ms'], 'Msamples/s', d['value'], 'nodes/ray %.2f tris/ray %.2f' % (c['nodes']/c['rays'], c['tri_tests']/c['rays']))"
  done
done
