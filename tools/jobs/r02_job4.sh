#!/bin/bash
# round 2, GPU job 4: the new bench.py (default run with workloads; self-launched 2-rank gloo run), launcher gpu test
cd $GRAFT_REPO_ROOT
( time python bench.py ) > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; echo "bench rc $?"; tail -4 gpurun_out/r02_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_bench_default.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['roofline']['bound'], d['roofline']['frac'], d['config']['traversal'], 'cpu', d.get('cpu_baseline',{}).get('value'))
for w in d.get('workloads',[]):
    print(w['config']['workload'], w['config']['tree'], w['config']['integrator'], w['value'], w['ms_per_step'], w['roofline']['bound'], (w.get('cpu_baseline') or {}).get('value'), (w.get('cpu_baseline') or {}).get('sample','')[:90])
PY
( time python bench.py --gpus 2 --dist-backend gloo --spp 16 --check --no-cpu-baseline ) > gpurun_out/r02_bench_gloo2.json 2> gpurun_out/r02_bench_gloo2.err; echo "gloo2 rc $?"; tail -3 gpurun_out/r02_bench_gloo2.err; cut -c1-400 gpurun_out/r02_bench_gloo2.json
TERRA_BENCH_DIST1=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-workloads --check > gpurun_out/r02_bench_dist1.json 2> gpurun_out/r02_bench_dist1.err; echo "dist1 rc $?"; cut -c1-300 gpurun_out/r02_bench_dist1.json
python -m pytest tests/test_bench_launcher.py -q -m gpu 2>&1 | tail -2
