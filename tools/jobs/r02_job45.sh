#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/scaled_hall.py --scale 100 --spp 4 --width 640 --height 360 2>&1 | grep -v amdgpu.ids
for sc in 100 1000 100000; do FUZZ_SCALE=$sc timeout -k 10 300 python tools/fuzz_vs_oracle.py 400 $((900+sc)) 2>&1 | tail -1; done
