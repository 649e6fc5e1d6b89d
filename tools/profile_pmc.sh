#!/bin/bash
# PMC passes of a short bench run: tools/profile_pmc.sh <tag> "<bench args>"
TAG=${1:-x}; ARGS=${2:-"--steps 2 --warmup 1 --spp 128"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 bench.py $ARGS --no-cpu-baseline > $OUT/$1.json 2> $OUT/$1.err; }
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"
run sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"
run sq3 "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAVE64_INSTS SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT"
python3 - <<PY
import csv, glob, collections
for name in ['sq1','sq2','sq3']:
    fs = glob.glob('$OUT/'+name+'/*/*_counter_collection.csv')
    if not fs: print(name,'no data', open('$OUT/'+name+'.err').read()[-300:]); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'terra_render_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(f"{k:26s} {sum(v)/len(v):.6g}")
PY
