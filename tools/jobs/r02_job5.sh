#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r02_gputests.log 2>&1; echo "gpu tests rc $?"; tail -2 gpurun_out/r02_gputests.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['ms_per_step'], d['value'])"
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-workloads --integrator direct 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('direct', d['ms_per_step'], d['value'])"
python bench.py --workload hall_1080p_256spp --sample-split 1 --steps 2 --warmup 1 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hall auto', d['ms_per_step'], d['value'])"
