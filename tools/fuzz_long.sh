#!/bin/bash
# long randomised parity runs (run on the GPU box; totals per round in CHANGELOG.md): device (three traversal modes, both tree builders where applicable) vs oracle
cd $GRAFT_REPO_ROOT
for spec in "1 6000 201" "2 1500 202" "4 600 203" "100 1500 204" "1000 800 207" "100000 800 208" "0.001 300 205"; do
  set -- $spec
  FUZZ_SCALE=$1 python tools/fuzz_vs_oracle.py $2 $3 2>&1 | tail -2
done
python tools/fuzz_split_shard.py 2000 206 2>&1 | tail -1
