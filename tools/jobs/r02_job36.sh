#!/bin/bash
cd $GRAFT_REPO_ROOT
export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_sd.so
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for v in base sd base sd; do
  if [ "$v" = base ]; then unset TERRA_AMD_LIB; else export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_$v.so; fi
  for wl in "cornell_1080p_512spp" "cornell_1080p_512spp --integrator direct --spp 128" "cornell_1080p_512spp --integrator mis --spp 64" "cornell_phong_1080p_512spp --spp 128"; do
    timeout -k 10 200 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $wl', 'kernel_ms', d['roofline']['kernel_ms'], 'Msamples/s', d['value'])"
  done
done
