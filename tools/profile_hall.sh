#!/bin/bash
# HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, separate passes) of the config-3 workload, reference tree and fast tree.
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_hall; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for tree in reference fast; do
  ARGS="bench.py --workload hall_1080p_256spp --tree $tree --steps 1 --warmup 0 --no-cpu-baseline"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${tree}_fetch -- python3 $ARGS > $OUT/${tree}_fetch.json 2> $OUT/${tree}_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${tree}_write -- python3 $ARGS > $OUT/${tree}_write.json 2> $OUT/${tree}_write.err
done
python3 - <<PY
import csv, glob, json
for tree in ("reference", "fast"):
    tot = {}
    for c, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        for f in glob.glob("$OUT/%s_%s/*/*_counter_collection.csv" % (tree, sub)):
            for r in csv.DictReader(open(f)):
                if "terra_render_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c: tot.setdefault(c, []).append(float(r["Counter_Value"]))
    b = json.loads(open("$OUT/%s_fetch.json" % tree).read().strip().splitlines()[-1])
    fk, wk = sum(tot["FETCH_SIZE"]) / len(tot["FETCH_SIZE"]), sum(tot["WRITE_SIZE"]) / len(tot["WRITE_SIZE"])
    print(json.dumps({"tree": tree, "fetch_size_kb": fk, "write_size_kb": wk, "hbm_bytes_per_launch": int((2 * fk + wk) * 1024), "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"],
                      "kernel_ms_under_pmc": b["roofline"]["kernel_ms"], "Msamples/s": b["value"]}))
PY
