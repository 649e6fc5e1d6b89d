#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r02_j31_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r02_j31_tests.log
for wl in "cornell_1080p_512spp" "cornell_1080p_512spp --integrator direct" "cornell_1080p_512spp --integrator mis --spp 256" "cornell_1080p_512spp --tree reference" "hall_1080p_256spp --spp 8 --sample-split 1 --tree reference"; do
    timeout -k 10 200 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', 'ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'Msamples/s', d['value'])"
done
