#!/bin/bash
# third set of seeds (round 4's third session: the fast tree's builder changed -- 32 bins, exact sweep on ranges of <= 1024 triangles, which is every range of these
# small soups); run on the GPU box under a time limit, progress lines keep the run visibly alive
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_final; L=gpurun_out/r04_final/fuzz_e.log; : > $L
for spec in "1 4000 801" "2 1000 802" "4 400 803" "100 1000 804" "1000 500 807" "100000 500 808"; do
  set -- $spec
  FUZZ_SCALE=$1 python tools/fuzz_vs_oracle.py $2 $3 2>&1 | grep -v amdgpu.ids | tee -a $L
done
FUZZ_EXT=1 python tools/fuzz_vs_oracle.py 1500 809 2>&1 | grep -v amdgpu.ids | tee -a $L
python tools/fuzz_split_shard.py 1000 811 2>&1 | grep -v amdgpu.ids | tee -a $L
