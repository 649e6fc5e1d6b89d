#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in base g4 base g4; do
  if [ "$v" = base ]; then unset TERRA_AMD_LIB; else export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_$v.so; fi
  for wl in "cornell_phong_1080p_512spp --spp 256"; do
    timeout -k 10 200 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $wl', 'kernel_ms', d['roofline']['kernel_ms'], 'Msamples/s', d['value'])"
  done
done
