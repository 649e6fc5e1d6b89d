"""Turns the rocprofv3 output of tools/profile_bench.sh (+ optionally tools/profile_pmc.sh) into the committed summary:
    python tools/summarize_profile.py gpurun_out/prof_<tag> [gpurun_out/pmc_<tag2>] > profiles/<name>.md
and prints the traffic entry for profiles/roofline_traffic.json on stderr."""
import collections, csv, glob, json, sys

prof = sys.argv[1]
pmc_dir = sys.argv[2] if len(sys.argv) > 2 else None


def counters(directory, sub):
    fs = glob.glob(f"{directory}/{sub}/*/*_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); meta = {}
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "terra_" not in k: continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {x: r[x] for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Grid_Size", "Workgroup_Size")}
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}, meta


print("## kernel stats (rocprofv3 --kernel-trace --stats)\n```")
for f in glob.glob(f"{prof}/trace/*/*_kernel_stats.csv"):
    rows = list(csv.reader(open(f)))
    print(",".join(f'"{x}"' for x in rows[0]))
    for r in rows[1:]:
        if "terra_" in r[0]: print(",".join(f'"{x}"' if i == 0 else x for i, x in enumerate(r)))
print("```")
bench = json.loads(open(f"{prof}/trace_bench.json").read().strip().splitlines()[-1])
print(f"bench.py's own HIP-event timing in the same run: kernel_ms = {bench['roofline']['kernel_ms']} (render + resolve kernels of one step).\n")

fetch, meta = counters(prof, "pmc_fetch"); write, _ = counters(prof, "pmc_write"); tcc, _ = counters(prof, "pmc_tcc")
print("## HBM traffic per launch (PMC, separate passes; MI355X_MICROARCH.md HBM section: bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 on gfx950)\n```")
total = 0.0; entry = {}
for k in sorted(set(fetch) | set(write)):
    fkb, wkb = fetch.get(k, {}).get("FETCH_SIZE", 0.0), write.get(k, {}).get("WRITE_SIZE", 0.0)
    b = (2 * fkb + wkb) * 1024; total += b
    short = k.split("(")[0].replace("void ", "")
    print(f"{short:44s} FETCH_SIZE {fkb:12.1f} KB  WRITE_SIZE {wkb:12.1f} KB  -> {b / 1e6:9.1f} MB" + (f"   TCC hit {tcc[k].get('TCC_HIT_sum', 0):.0f} miss {tcc[k].get('TCC_MISS_sum', 0):.0f}" if k in tcc else ""))
    entry[short] = {"fetch_size_kb": fkb, "write_size_kb": wkb}
print(f"one step (all terra kernels): {total / 1e6:.1f} MB; ALGORITHMIC bytes per launch: {bench['roofline']['algorithmic_bytes_per_launch'] / 1e6:.1f} MB")
print("```")
sys.stderr.write(json.dumps({bench["config"]["workload"]: {"hbm_bytes_per_launch": int(total), "kernels": entry}}, indent=1) + "\n")

for label, directory, subs in (("SQ counters per launch (bench defaults)", prof, ["pmc_sq"]), ("SQ counters per launch (tools/profile_pmc.sh run)", pmc_dir, ["sq1", "sq2", "sq3"])):
    if not directory: continue
    allc = {}
    for sub in subs:
        c, m = counters(directory, sub)
        for k, d in c.items():
            if "terra_render_kernel" in k: allc.update(d); meta_r = m[k]
    if not allc: continue
    print(f"\n## {label}\n```")
    for k in sorted(allc): print(f"{k:26s} {allc[k]:.6g}")
    print("```")
    g = allc.get
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        print(f"VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = {100 * g('SQ_THREAD_CYCLES_VALU') / (64 * g('SQ_ACTIVE_INST_VALU')):.0f} %")
    if g("SQ_WAVE_CYCLES") and g("SQ_BUSY_CYCLES"):
        print(f"waves per SIMD (average) = 4 x SQ_WAVE_CYCLES / (SQ_BUSY_CYCLES / 32 x 1024) = {4 * g('SQ_WAVE_CYCLES') / (g('SQ_BUSY_CYCLES') / 32 * 1024):.2f}")
    print("render kernel resources:", meta_r)
print("\nbench line of the traced run:\n```\n" + json.dumps(bench) + "\n```")
