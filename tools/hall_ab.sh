cd $GRAFT_REPO_ROOT
for cfg in "bfs 1" "dfs 1" "bfs 4" "dfs 4"; do set -- $cfg
  if [ $1 = dfs ]; then export TERRA_AMD_NODE_ORDER=dfs; else unset TERRA_AMD_NODE_ORDER; fi
  python bench.py --workload hall_1080p_256spp --spp 8 --steps 2 --warmup 1 --no-cpu-baseline --sample-split $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 split $2', 'kernel_ms', d['roofline']['kernel_ms'], 'Msamples/s', d['value'], 'frac', d['roofline']['frac'])"
done
