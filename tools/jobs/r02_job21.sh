#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in mm base; do
  if [ "$v" = base ]; then unset TERRA_AMD_LIB; else export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_$v.so; fi
  for wl in "--spp 4 --sample-split 1" "--spp 16 --sample-split 1" "--spp 64 --sample-split 8"; do
    timeout -k 10 200 python bench.py --workload hall_1080p_256spp $wl --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $wl', 'kernel_ms', d['roofline']['kernel_ms'], 'Msamples/s', d['value'], d['counters_per_launch']['nodes'], d['counters_per_launch']['tri_tests'])"
  done
done
