#!/bin/bash
# a second set of seeds for tools/fuzz_long.sh's runs, plus the unpinned extensions (run on the GPU box; progress lines every 1000 cases keep the run visibly alive)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_final; L=gpurun_out/r04_final/fuzz_c.log; : > $L
for spec in "1 8000 701" "2 2000 702" "4 800 703" "100 2000 704" "1000 1000 707" "100000 1000 708"; do
  set -- $spec
  FUZZ_SCALE=$1 python tools/fuzz_vs_oracle.py $2 $3 2>&1 | grep -v amdgpu.ids | tee -a $L
done
FUZZ_EXT=1 python tools/fuzz_vs_oracle.py 3000 709 2>&1 | grep -v amdgpu.ids | tee -a $L
FUZZ_EXT=1 FUZZ_SCALE=100 python tools/fuzz_vs_oracle.py 1000 710 2>&1 | grep -v amdgpu.ids | tee -a $L
python tools/fuzz_split_shard.py 3000 711 2>&1 | grep -v amdgpu.ids | tee -a $L
