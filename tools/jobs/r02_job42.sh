#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in base ilp memc; do
  if [ "$v" = base ]; then unset TERRA_AMD_LIB; else export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_$v.so; fi
  for wl in "cornell_1080p_512spp" "cornell_1080p_512spp --integrator direct --spp 128" "hall_1080p_256spp --spp 64 --sample-split 1"; do
    timeout -k 10 200 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $wl', 'kernel_ms', d['roofline']['kernel_ms'], 'Msamples/s', d['value'])"
  done
done
