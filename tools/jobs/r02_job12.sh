#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/bounds_check_run.sh > gpurun_out/r02_bounds_check.log 2>&1; tail -12 gpurun_out/r02_bounds_check.log
unset TERRA_AMD_LIB
python tools/host_api_rate.py > gpurun_out/r02_host_api_rate.json 2> gpurun_out/r02_host_api_rate.err; cat gpurun_out/r02_host_api_rate.json | head -40
python tools/shard_balance.py --spp 512 --split 8 > gpurun_out/r02_shard_balance.log 2>&1; cat gpurun_out/r02_shard_balance.log
for integ in simple direct mis; do python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-workloads --integrator $integ 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cornell $integ', d['ms_per_step'], d['value'], d['mrays_per_s'])"; done | tee gpurun_out/r02_cornell_integrators.log
for integ in simple direct mis; do python bench.py --workload hall_1080p_256spp --spp 32 --sample-split 1 --steps 2 --warmup 1 --no-cpu-baseline --no-workloads --integrator $integ 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hall auto 32spp $integ', d['ms_per_step'], d['value'], d['mrays_per_s'])"; done | tee gpurun_out/r02_hall_integrators.log
for integ in direct mis; do python bench.py --workload hall_1080p_256spp --spp 8 --sample-split 1 --tree reference --steps 1 --warmup 1 --no-cpu-baseline --no-workloads --integrator $integ 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hall reference 8spp $integ', d['ms_per_step'], d['value'], d['mrays_per_s'])"; done | tee -a gpurun_out/r02_hall_integrators.log
