#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r02_j51_tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r02_j51_tests.log
timeout -k 10 300 python tools/scaled_hall.py --scale 100 --spp 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_scaled_hall.log
timeout -k 10 300 python tools/scaled_hall.py --scale 100 --spp 8 --integrator 1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02_scaled_hall.log
export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_sc.so
echo "self-checking build:"
for hook in "" 0.0005 0.03 0.2 3.0; do
  if [ -z "$hook" ]; then unset TERRA_AMD_TEST_SHRINK_REFERENCE_BOXES; else export TERRA_AMD_TEST_SHRINK_REFERENCE_BOXES=$hook; fi
  echo "hook '$hook'"; timeout -k 10 300 python tools/scaled_hall.py --scale 100 --spp 4 --width 960 --height 540 2>&1 | grep "tree mode 2\|bit for bit"
done
unset TERRA_AMD_TEST_SHRINK_REFERENCE_BOXES
for sc in 100 1000 100000; do FUZZ_SCALE=$sc timeout -k 10 300 python tools/fuzz_vs_oracle.py 300 $((500+sc)) 2>&1 | tail -1; done
