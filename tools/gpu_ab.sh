# same-box A/B of experiment builds (dev tool, runs on the GPU box): tools/gpu_ab.sh "<variant> <variant> ..." ["<workload> <split> <steps> [bench.py args]" ...]
# "product" = terra_amd/libterra_amd.so; other variants: python -m terra_amd.build --variant NAME -DFOO=1 -> terra_amd/libterra_amd_NAME.so
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04_ab
variants="$1"; shift
if [ $# -eq 0 ]; then set -- "hall_1080p_256spp 32 3" "hall_1080p_64spp_direct 8 3" "spheres_1080p_1024spp 32 2" "hall_x100_1080p_64spp 4 3"; fi
for round in 1 2; do for v in $variants; do
  if [ "$v" = "product" ]; then unset TERRA_AMD_LIB; else export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_$v.so; fi
  line="$v:"
  for w in "$@"; do
    read -r wl sp st extra <<< "$w"; tag=$(echo "$wl $extra" | tr -c 'A-Za-z0-9_\n' '_')
    python bench.py --workload $wl --sample-split $sp --steps $st $extra --no-workloads --no-host-api --no-cpu-baseline > gpurun_out/r04_ab/bench_${v}_$tag.json 2> gpurun_out/r04_ab/bench_${v}_$tag.err || { tail -5 gpurun_out/r04_ab/bench_${v}_$tag.err; exit 1; }
    line="$line $(python -c "import json; d=json.load(open('gpurun_out/r04_ab/bench_${v}_$tag.json')); print('$tag', d['ms_per_step'])")"
  done
  echo "$line"
done; done
