#!/bin/bash
cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r02_prof
python3 tools/profile_r02.py > gpurun_out/r02_profile_driver.log 2>&1; tail -3 gpurun_out/r02_profile_driver.log
cp gpurun_out/r02_prof/r02_pmc.json profiles/r02_pmc.json
( time python bench.py ) > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; echo "bench rc $?"; tail -4 gpurun_out/r02_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']; print('headline', d['value'], d['ms_per_step'], r['bound'], r['achieved'], r['frac'], r.get('lane_util'), r.get('lds_bank_conflict_frac'), 'stale', r.get('pmc_stale'))
for w in d.get('workloads',[]):
    r=w['roofline']; print(w['config']['workload'], w['config']['tree'], w['value'], w['ms_per_step'], r['bound'], r['achieved'], r['frac'], 'stale', r.get('pmc_stale'), (w.get('cpu_baseline') or {}).get('value'))
PY
