#!/bin/bash
# A/B of experiment builds on the GPU box: tools/ab_variants.sh "<bench args>" v1 v2 ...   ("base" = libterra_amd.so)
ARGS="$1"; shift
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset TERRA_AMD_LIB; else export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_$v.so; fi
  python bench.py $ARGS --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'round $round', 'kernel_ms', d['roofline']['kernel_ms'], 'Msamples/s', d['value'])"
done; done
