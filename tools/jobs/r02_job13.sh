#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 1 2 4 8 16; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-workloads --sample-split $s 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split $s', d['ms_per_step'], d['value'])"; done
for s in 4 8 16; do python tools/shard_balance.py --spp 512 --split $s 2>/dev/null | grep "whole\|tile   64 world 8"; done
