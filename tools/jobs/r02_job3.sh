#!/bin/bash
# round 2, GPU job 3: leaf-box cull / automatic traversal: full gpu suite, then A/B of the headline and the hall
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r02_gputests.log 2>&1; echo "gpu tests rc $?"; tail -4 gpurun_out/r02_gputests.log
for t in reference auto; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --tree $t 2>/dev/null | tail -1 > gpurun_out/r02_bench_cornell_$t.json
  python -c "import json;d=json.load(open('gpurun_out/r02_bench_cornell_$t.json'));print('cornell',d['config']['tree'],d['ms_per_step'],d['value'],d['counters_per_launch']['tri_tests'])"
done
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --tree auto --integrator direct 2>/dev/null | tail -1 > gpurun_out/r02_bench_cornell_direct_auto.json
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --tree reference --integrator direct 2>/dev/null | tail -1 > gpurun_out/r02_bench_cornell_direct_ref.json
python -c "
import json
for t in ('auto','ref'):
    d=json.load(open('gpurun_out/r02_bench_cornell_direct_%s.json'%t)); print('direct',t,d['ms_per_step'],d['value'])"
python bench.py --workload hall_1080p_256spp --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --tree reference --sample-split 1 2>/dev/null | tail -1 > gpurun_out/r02_bench_hall_ref.json
python bench.py --workload hall_1080p_256spp --steps 2 --warmup 1 --no-cpu-baseline --tree auto --sample-split 1 2>/dev/null | tail -1 > gpurun_out/r02_bench_hall_auto.json
python -c "
import json
for t in ('ref','auto'):
    d=json.load(open('gpurun_out/r02_bench_hall_%s.json'%t)); print('hall',t,d['config']['spp'],d['ms_per_step'],d['value'])"
TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_ps.so python tools/phase_stats.py --spp 512 --split 8 > gpurun_out/r02_phase_cull.log 2>&1; tail -7 gpurun_out/r02_phase_cull.log
