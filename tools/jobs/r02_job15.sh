#!/bin/bash
# decoupled loop for the fast tree: A/B of exit fractions against the coupled loop (f0), identical-image check by the gpu tests
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "hall or fast or tree or config3 or spheres or fuzz" > gpurun_out/r02_j15_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r02_j15_tests.log
for v in f0 base e4 e12 e15; do
  if [ "$v" = base ]; then unset TERRA_AMD_LIB; else export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_$v.so; fi
  for wl in "hall_1080p_256spp --spp 64 --sample-split 1" "spheres_1080p_1024spp --spp 128 --sample-split 8"; do
    timeout -k 10 200 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['workload'], 'kernel_ms', d['roofline']['kernel_ms'], 'Msamples/s', d['value'])"
  done
done
unset TERRA_AMD_LIB
export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_ps0.so
python tools/phase_stats.py --scene hall --spp 32 --split 1 --tree 2 2>&1 | tail -8
export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_ps.so
python tools/phase_stats.py --scene hall --spp 32 --split 1 --tree 2 2>&1 | tail -8
