#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in base ld16 ld12 ld16w5; do
  if [ "$v" = base ]; then unset TERRA_AMD_LIB; else export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_$v.so; fi
  for wl in "cornell_1080p_512spp --spp 128 --sample-split 8 --integrator direct"; do
    timeout -k 10 200 python bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $wl', 'kernel_ms', d['roofline']['kernel_ms'], 'Msamples/s', d['value'])"
  done
  timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "cornell or golden or integrator or trace" 2>&1 | tail -1
done
