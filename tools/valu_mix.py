"""Static half of the VALU mix ceiling (dev tool, no GPU): for each render kernel named, the mean issue cost of the VALU opcodes that the SQ_INSTS_VALU_* class
counters do NOT name ("other": compares, selects, min / max, permutes, moves, shifts, bit operations), weighted by their static frequency in the kernel's ISA (x 8 per loop level) and
priced with the measured per-opcode issue cycles (profiles/r02_measurements/valu_rates.log; 4.2 cycles for opcodes not measured: the common class).

    tools/kernel_resources.sh                      # writes /tmp/terra_isa/*.s (hipcc --save-temps of render_kernels.hip, ~3 min)
    python tools/valu_mix.py [--out profiles/r04_static_valu_mix.json]

bench.valu_mix_ceiling() combines `other_cycles` with the dynamic class counts of the PMC passes mix1 / mix2 (tools/profile_round.py copies it into the PMC record).
"""
import argparse
import collections
import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
S = "/tmp/terra_isa/render_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"
WEIGHT = 8          # an instruction at loop depth d counts WEIGHT^d times: the static stand-in for "inner loops run more often" (the named classes come from hardware counters; only the
                    # relative frequencies INSIDE the unnamed class are estimated this way)
# opcodes the hardware counters put in a named class (so NOT "other")
NAMED = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mad|mac)_(f16|f32|f64)|^v_(rcp|rsq|sqrt|exp|log|sin|cos)_|^v_cvt_|^v_(add|sub|subrev|addc|subb|mul_lo|mul_hi|mad|mul)_(co_)?(u|i)(32|64|24)|^v_(lshl_add|add3|add_lshl|lshl_or|and_or|or3|xad)_u32|^v_fma_mix|^v_pk_|^v_mad_u64|^v_lshl_add_u64|^v_div_")


def measured_cycles():
    t = {}
    for ln in (ROOT / "profiles/r02_measurements/valu_rates.log").read_text().splitlines():
        m = re.match(r"^(v_\w+)(?: (\w+))?\s+waves/SIMD.*= ([\d.]+) cycles", ln)
        if m and not m.group(2):
            t[m.group(1)] = float(m.group(3))
    t["v_cndmask_b32"] = 4.24; t["v_cmp"] = 4.2; t["v_mov_b32"] = 2.63; t["v_max3_f32"] = t.get("v_min3_f32", 4.13); t["v_perm_b32"] = 4.27; t["v_alignbit_b32"] = 4.27
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(ROOT / "profiles" / "r04_static_valu_mix.json"))
    ap.add_argument("--isa", default=S)
    a = ap.parse_args()
    cyc = measured_cycles()
    txt = open(a.isa).read().split("\n")
    out = {}
    i = 0
    while i < len(txt):
        m = re.match(r"^_Z19terra_render_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EEv15DevRenderParams:", txt[i])
        if not m:
            i += 1; continue
        name = "terra_render_kernel<%s, %s, %s, %s>" % m.groups()
        ops = collections.Counter(); i += 1
        depth = 0          # loop depth of the current block, from the compiler's block comments ("in Loop: Header=BB0_86 Depth=2"; "Parent Loop ... Depth=1" on a header)
        while i < len(txt) and "s_endpgm" not in txt[i]:
            ln = txt[i]; t = ln.strip(); i += 1
            if re.match(r"^\.LBB\d+_\d+:", ln) or t.startswith("; %bb."):
                m2 = re.findall(r"Depth=(\d+)", ln)
                depth = max(int(x) for x in m2) if m2 else 0
                if "Parent Loop" in ln and "in Loop" not in ln: depth += 1          # a loop header inside a parent loop: one level deeper than the parent named
                continue
            if t.startswith("v_"):
                ops[re.sub(r"_(e32|e64|sdwa|dpp)$", "", t.split()[0])] += WEIGHT ** depth
        other = {k: v for k, v in ops.items() if not NAMED.match(k)}
        n = sum(other.values())

        def price(op):
            if op.startswith("v_cmp"):
                return cyc["v_cmp"]
            if op in cyc:
                return cyc[op]
            base = re.sub(r"_(u|i|b)(16|32|64)$", "", op)
            for k, v in cyc.items():
                if k.startswith(base):
                    return v
            return 4.2
        out[name] = {"other_cycles": round(sum(price(k) * v for k, v in other.items()) / max(1, n), 3), "static_valu": sum(ops.values()), "static_other": n,
                     "other_top": dict(collections.Counter(other).most_common(8))}
    Path(a.out).write_text(json.dumps(out, indent=1))
    for k in sorted(out):
        if re.search(r"<[012], 0, [12], 1>", k):
            print(k, out[k])
    print("wrote", a.out, len(out), "kernels")


if __name__ == "__main__":
    main()
