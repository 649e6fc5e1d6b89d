#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r02_j50_tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r02_j50_tests.log
timeout -k 10 300 python tools/scaled_hall.py --scale 100 --spp 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_scaled_hall.log
timeout -k 10 300 python tools/scaled_hall.py --scale 100 --spp 8 --integrator 1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02_scaled_hall.log
timeout -k 10 300 python tools/scaled_hall.py --scale 100 --spp 8 --integrator 2 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02_scaled_hall.log
for sc in 100 1000 100000; do FUZZ_SCALE=$sc timeout -k 10 300 python tools/fuzz_vs_oracle.py 500 $((300+sc)) 2>&1 | tail -1; done
timeout -k 10 200 python bench.py --workload hall_1080p_256spp --spp 64 --sample-split 1 --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in-range hall kernel_ms', d['roofline']['kernel_ms'], 'Msamples/s', d['value'])"
