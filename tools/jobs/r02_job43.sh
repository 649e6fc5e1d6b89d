#!/bin/bash
# full-size measurements on the final kernels: config 5's frame on one GPU, shard balance of the hall, a longer fuzz
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --workload hall_2160p_4096spp --sample-split 1 --steps 1 --warmup 0 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hall_2160p_4096spp one GPU: ms', d['ms_per_step'], 'Msamples/s', d['value'], 'Mrays/s', d['mrays_per_s'])" | tee gpurun_out/r02_config5_one_gpu.log
echo progress 1
timeout -k 10 300 python tools/shard_balance.py --workload hall --spp 64 --world 8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_shard_balance_hall.log | tail -10
echo progress 2
FUZZ_SCALE=1 timeout -k 10 900 python tools/fuzz_vs_oracle.py 12000 777 2>&1 | tail -1 | tee gpurun_out/r02_fuzz_12k.log
