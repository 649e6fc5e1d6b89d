#!/bin/bash
# integrators x scenes on the round's final kernels (ms per step, Msamples/s, Mrays/s)
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --workload $2 --steps $3 --warmup 1 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['mrays_per_s'])"; }
run "cornell simple" "cornell_1080p_512spp" 3
run "cornell direct" "cornell_1080p_512spp --integrator direct" 3
run "cornell mis" "cornell_1080p_512spp --integrator mis" 3
run "cornell replica simple" "cornell_1080p_512spp --tree reference" 3
run "cornell-phong simple" "cornell_phong_1080p_512spp" 3
run "hall auto 256spp simple" "hall_1080p_256spp --sample-split 1" 2
run "hall auto 32spp direct" "hall_1080p_256spp --spp 32 --sample-split 1 --integrator direct" 3
run "hall auto 32spp mis" "hall_1080p_256spp --spp 32 --sample-split 1 --integrator mis" 3
run "hall reference 8spp simple" "hall_1080p_256spp --spp 8 --sample-split 1 --tree reference" 2
run "hall reference 8spp direct" "hall_1080p_256spp --spp 8 --sample-split 1 --tree reference --integrator direct" 2
run "hall reference 8spp mis" "hall_1080p_256spp --spp 8 --sample-split 1 --tree reference --integrator mis" 2
run "spheres auto 1024spp simple" "spheres_1080p_1024spp --sample-split 8" 2
run "spheres auto 64spp direct" "spheres_1080p_1024spp --spp 64 --sample-split 8 --integrator direct" 3
run "spheres auto 64spp mis" "spheres_1080p_1024spp --spp 64 --sample-split 8 --integrator mis" 3
