"""rocprofv3 passes of the bench workloads -> <out>/<round>_pmc.json + <out>/<round>_rocprof_summary.md (run on the GPU box; copy both
into profiles/ afterwards).

    python3 tools/profile_round.py [--round r03] [--only KEY_SUBSTRING] [--out gpurun_out/r03_prof]

For every configuration below: one `--kernel-trace --stats` run and four separate `--pmc` runs (SQ group 1, SQ group 2,
FETCH_SIZE, WRITE_SIZE; counters are never combined with tracing) of `python3 bench.py <args> --no-cpu-baseline
--no-workloads --no-host-api`. This script itself never touches the GPU: every run is a child process with the program right after `--`.
The JSON it writes is what bench.py's `roofline` reads (keyed like bench.pmc_key); the markdown is the human summary.
HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section).
"""
import argparse
import collections
import re
import csv
import glob
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (imports nothing GPU-related at module level)

CORNELL_SPLIT = 8          # what bench.DEFAULT_SPLIT = 0 (the library's automatic split) resolves to for the whole 1080p frame of an LDS-resident scene: the key bench.py looks its record up under
CONFIGS = [
    # (workload, tree, integrator, split, extra bench args)
    ("cornell_1080p_512spp", "auto", "simple", CORNELL_SPLIT, ["--steps", "2", "--warmup", "1"]),
    ("cornell_1080p_512spp", "reference", "simple", CORNELL_SPLIT, ["--steps", "2", "--warmup", "1"]),
    ("cornell_1080p_512spp_direct", "auto", "direct", CORNELL_SPLIT, ["--steps", "2", "--warmup", "1"]),
    ("hall_1080p_256spp", "auto", "simple", 32, ["--steps", "2", "--warmup", "1"]),
    ("hall_1080p_256spp", "reference", "simple", 32, ["--steps", "1", "--warmup", "0"]),
    ("spheres_1080p_1024spp", "auto", "simple", 32, ["--steps", "1", "--warmup", "1"]),
    ("hall_x100_1080p_64spp", "auto", "simple", 4, ["--steps", "2", "--warmup", "1"]),
    ("hall_1080p_64spp_direct", "auto", "direct", 8, ["--steps", "2", "--warmup", "1"]),
]
PASSES = {
    "sq1": ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_VALU"],
    "sq2": ["SQ_THREAD_CYCLES_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_BRANCH", "SQ_WAIT_INST_ANY"],
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    # dynamic VALU instruction mix by class (bench.valu_mix_ceiling prices it with the measured per-class issue cycles)
    "mix1": ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64"],
    "mix2": ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_ADD_F16", "SQ_INSTS_VALU_MUL_F16", "SQ_INSTS_VALU_FMA_F16", "SQ_INSTS_VALU_TRANS_F16"],
}
# passes for the workloads that read the scene from global memory (what binds a divergent gather: wave stalls, the texture addresser, the L1 tags, the L2): collected
# like the others, each in its own run. TA / TCP / TCC counters are summed over their instances (the `_sum` derived names).
GATHER_PASSES = {
    "sq3": ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_FLAT", "SQ_INSTS_FLAT", "SQ_INST_CYCLES_VMEM_RD", "SQ_WAVE_CYCLES"],
    # TA / TCP take two counters per pass on gfx950 (four: "Request exceeds the capabilities of the hardware", and rocprofv3 then hangs in its abort handler: every run below has a time limit)
    "ta": ["TA_TA_BUSY_sum", "TA_FLAT_READ_WAVEFRONTS_sum"],
    "ta2": ["TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum"],
    "ta3": ["TA_BUSY_avr", "TA_BUFFER_READ_WAVEFRONTS_sum"],
    "tcp": ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"],
    "tcp2": ["TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum"],
    "tcp3": ["TCP_TCC_READ_REQ_LATENCY_sum", "TCP_GATE_EN1_sum"],
    "tcp4": ["TCP_TOTAL_ACCESSES_sum", "TCP_TA_TCP_STATE_READ_sum"],
    "tcc": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum"],
    "tcc2": ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"],
    "grbm": ["GRBM_GUI_ACTIVE", "GRBM_COUNT"],
}


def run(cmd, log, limit=300):
    """one child run (the program right after `--`); killed -- with its own process group -- when it exceeds `limit` seconds"""
    import signal
    env = dict(os.environ); env["TMPDIR"] = "/tmp"
    with open(log, "w") as f:
        p = subprocess.Popen(cmd, cwd=str(ROOT), env=env, stdout=subprocess.PIPE, stderr=f, text=True, start_new_session=True)
        try:
            so, _ = p.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)          # the exact group this call started
            except ProcessLookupError:
                pass
            so, _ = p.communicate()
            f.write(f"\nprofile_round: killed after {limit} s\n")
            return 124, so or ""
    return p.returncode, so


def counters(directory):
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); meta = {}
    for fcsv in glob.glob(f"{directory}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fcsv)):
            k = r["Kernel_Name"]
            if "terra_" not in k:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {x: r.get(x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Grid_Size", "Workgroup_Size")}
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}, meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--round", default="r04")
    ap.add_argument("--passes", default="", help="comma-separated pass names (default: the four standard passes; GATHER_PASSES names are accepted too, `gather` = all of them)")
    ap.add_argument("--no-trace", action="store_true", help="skip the --kernel-trace --stats run (needs a bench_line.json from an earlier run under --out)")
    ap.add_argument("--reaggregate", action="store_true", help="no runs: rebuild the JSON and the summary from the per-pass CSVs (and bench lines) an earlier run left under --out")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    out = Path(a.out or str(ROOT / "gpurun_out" / f"{a.round}_prof")); out.mkdir(parents=True, exist_ok=True)
    jpath = out / f"{a.round}_pmc.json"
    result = json.loads(jpath.read_text()) if jpath.exists() else {}
    md = []
    digest = bench.source_digest()
    allp = dict(PASSES); allp.update(GATHER_PASSES)
    names = [n for n in a.passes.split(",") if n]
    if "gather" in names:
        names = [n for n in names if n != "gather"] + list(GATHER_PASSES)
    passes = {n: allp[n] for n in names} if names else PASSES
    auto_gather = not names          # default invocation: the workloads read from global memory also get the gather passes
    for wl, tree, integ, split, extra in CONFIGS:
        key = bench.pmc_key(wl, tree, integ, split, 0)
        if a.only and a.only not in key:
            continue
        tag = key.replace("|", "_").replace("=", "-")
        args = ["bench.py", "--workload", wl, "--tree", tree, "--sample-split", str(split), "--no-cpu-baseline", "--no-workloads", "--no-host-api"] + extra
        d = out / tag; d.mkdir(exist_ok=True)
        print("==", key, flush=True)
        if a.reaggregate or (a.no_trace and (d / "bench_line.json").exists()):
            if (d / "bench_line.json").exists():
                bl = json.loads((d / "bench_line.json").read_text())
            elif key in result:          # an earlier version did not keep the line: rebuild the fields the summary needs from the JSON record
                o = result[key]
                bl = {"roofline": {"kernel_ms": o["kernel_ms"], "algorithmic_bytes_per_launch": bench.algorithmic_bytes(o["counters_per_launch"])}, "ms_per_step": o["ms_per_step"], "value": o["value"],
                      "counters_per_launch": o["counters_per_launch"], "config": {"traversal": o["traversal"]}}
            else:
                continue
        else:
            rc, so = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", str(d / "trace"), "--", "python3"] + args, d / "trace.err")
            line = [ln for ln in so.splitlines() if ln.startswith("{")]
            if rc != 0 or not line:
                print("  trace run failed", rc, open(d / "trace.err").read()[-500:]); continue
            bl = json.loads(line[-1])
            (d / "bench_line.json").write_text(json.dumps(bl))
        rec = {"source_digest": digest, "kernel_ms": bl["roofline"]["kernel_ms"], "ms_per_step": bl["ms_per_step"], "value": bl["value"], "counters_per_launch": bl["counters_per_launch"],
               "traversal": bl["config"]["traversal"]}
        stats_rows = []
        for fcsv in glob.glob(f"{d}/trace/**/*_kernel_stats.csv", recursive=True):
            rows = list(csv.reader(open(fcsv)))
            stats_rows = [rows[0]] + [r for r in rows[1:] if "terra_" in r[0] and "sincos24" not in r[0]]
        allc = {}; meta_r = {}; per_kernel = {}
        if key in result:            # passes of an earlier run (other pass names) are kept
            allc.update({k: v for k, v in result[key].get("pmc", {}).items()})
        these = dict(passes)
        if auto_gather and "fast tree" in rec["traversal"] or (auto_gather and "replica" in rec["traversal"] and "hall" in wl):
            these.update(GATHER_PASSES)
        for pname, ctrs in these.items():
            if not a.reaggregate:
                rc, so = run(["rocprofv3", "--pmc"] + ctrs + ["--output-format", "csv", "-d", str(d / pname), "--", "python3"] + args, d / f"{pname}.err")
                if rc != 0:
                    print("  pass", pname, "failed", rc, open(d / f"{pname}.err").read()[-300:]); continue
            c, m = counters(d / pname)
            # bench.py's timed launches run WITHOUT work counters (template argument COUNT = 0); its one extra counting launch is another instance
            # of the kernel (COUNT = 2) and must not be mixed in: the record is the timed kernel's
            timed = [k for k in c if re.search(r"terra_render_kernel<\d+, 0,", k)] or [k for k in c if "terra_render_kernel" in k]
            for k, v in c.items():
                per_kernel.setdefault(k, {}).update(v)
                if k in timed[:1]:
                    allc.update(v); meta_r = m[k]; rec["kernel_name"] = k.split("(")[0].replace("void ", "")
            print("  pass", pname, "ok", flush=True)
        g = allc.get
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            if g(k) is not None:
                rec[k] = g(k)
        if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            rec["lane_util"] = round(g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU")), 4)
        if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
            rec["lds_bank_conflict_frac"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4)
        if g("SQ_WAVE_CYCLES") and g("SQ_BUSY_CYCLES"):
            rec["waves_per_simd"] = round(4 * g("SQ_WAVE_CYCLES") / (g("SQ_BUSY_CYCLES") / 32 * 1024), 3)
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            rec["fetch_size_kb"] = g("FETCH_SIZE"); rec["write_size_kb"] = g("WRITE_SIZE")
            rec["hbm_bytes_per_launch"] = int((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024)
        rec["resources"] = meta_r or result.get(key, {}).get("resources", {})
        rec["pmc"] = dict(allc)      # every counter collected for the timed kernel, per launch (all passes so far)
        static = {}
        for sf in sorted((ROOT / "profiles").glob("r[0-9][0-9]_static_valu_mix.json")):
            static = json.loads(sf.read_text())
        if rec.get("kernel_name") in static:      # tools/valu_mix.py: the issue cost of the opcodes the class counters do not name, from this kernel's ISA
            rec["other_cycles"] = static[rec["kernel_name"]]["other_cycles"]
        for k2 in ("hbm_bytes_per_launch", "fetch_size_kb", "write_size_kb", "lane_util", "lds_bank_conflict_frac", "waves_per_simd", "kernel_name"):
            if k2 not in rec and k2 in result.get(key, {}):
                rec[k2] = result[key][k2]
        result[key] = rec
        jpath.write_text(json.dumps(result, indent=1))
        md.append(f"## {key}\n")
        md.append("`rocprofv3 --kernel-trace --stats -- python3 " + " ".join(args) + "` (+ separate `--pmc` passes: " + ", ".join(PASSES) + ")\n")
        md.append("```\n" + "\n".join(",".join(r) for r in stats_rows) + "\n```")
        md.append(f"bench.py's own HIP-event timing in the traced run: kernel_ms = {bl['roofline']['kernel_ms']}, ms_per_step = {bl['ms_per_step']}, {bl['value']} Msamples/s; traversal: {bl['config']['traversal']}\n")
        md.append(f"render kernel ({rec.get('kernel_name', '?')}: the timed launches, no work counters), per launch:\n```")
        for k in sorted(allc):
            md.append(f"{k:26s} {allc[k]:.6g}")
        md.append("```")
        t = bl["roofline"]["kernel_ms"] * 1e-3
        if g("SQ_INSTS_VALU"):
            md.append(f"VALU issue = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x 2.4 GHz x {t * 1e3:.2f} ms) = {g('SQ_INSTS_VALU') * 2 / (1024 * 2.4e9 * t):.3f}" +
                      (f"; lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = {rec.get('lane_util')}" if rec.get("lane_util") else "") +
                      (f"; LDS bank-conflict share = {rec.get('lds_bank_conflict_frac')}" if rec.get("lds_bank_conflict_frac") is not None else "") +
                      (f"; waves per SIMD = {rec.get('waves_per_simd')}" if rec.get("waves_per_simd") else ""))
        if rec.get("hbm_bytes_per_launch"):
            md.append(f"L2 fabric-side bytes per launch = (2 x {g('FETCH_SIZE'):.1f} + {g('WRITE_SIZE'):.1f}) KB x 1024 = {rec['hbm_bytes_per_launch'] / 1e6:.1f} MB -> {rec['hbm_bytes_per_launch'] / t / 1e9:.1f} GB/s "
                      f"({rec['hbm_bytes_per_launch'] / t / 8e12:.4f} of the 8 TB/s HBM peak); algorithmic bytes per launch {bl['roofline']['algorithmic_bytes_per_launch'] / 1e6:.1f} MB")
        md.append(f"resources: {meta_r}\n")
        for k, v in per_kernel.items():
            if "terra_render_kernel" not in k and "FETCH_SIZE" in v:
                md.append(f"(other kernel {k.split('(')[0]}: FETCH_SIZE {v.get('FETCH_SIZE', 0):.1f} KB, WRITE_SIZE {v.get('WRITE_SIZE', 0):.1f} KB per launch)")
        md.append("")
        (out / f"{a.round}_rocprof_summary.md").write_text(f"# rocprofv3 summaries, round {a.round} (tools/profile_round.py; source digest " + digest + ")\n\n" + "\n".join(md))
    print("wrote", jpath)


if __name__ == "__main__":
    main()
