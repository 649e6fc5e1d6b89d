"""Static cost model of one kernel's ISA (dev tool): per basic block, instructions by issue class and an estimate of SIMD
cycles from the measured per-class issue rates (profiles/r02_measurements/valu_rates.log: cycles per wave64 instruction at
2.4 GHz with 4 waves per SIMD). Usage: tools/kernel_resources.sh; python tools/isa_cost.py [mangled-name-substring]"""
import re, sys, collections
S = "/tmp/terra_isa/render_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"
name = sys.argv[1] if len(sys.argv) > 1 else "ILi0ELi0ELi1ELi1E"
FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_accvgpr"}
MED = {"v_and_b32": 3.3, "v_or_b32": 3.3, "v_xor_b32": 3.4, "v_add_u32": 3.5, "v_sub_u32": 3.5, "v_subrev_u32": 3.5, "v_lshrrev_b32": 2.9, "v_fma_f32": 3.9, "v_fmac_f32": 4.1, "v_not_b32": 3.3}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}
def cost(op):
    base = op.replace("_e32", "").replace("_e64", "").replace("_sdwa", "").replace("_dpp", "")
    if base in FAST: return 2.55, "fast"
    if base in MED: return MED[base], "med"
    if base in TRANS: return (16.2 if base.endswith("f64") else 8.1), "trans"
    return 4.2, "slow"
txt = open(S).read().split("\n")
start = next(i for i, l in enumerate(txt) if re.match(r"^_Z19terra_render_kernel" + name, l))
blocks = []; cur = ["entry", []]; blocks.append(cur)
for l in txt[start + 1:]:
    if "s_endpgm" in l: break
    m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; (%bb\.\d+):", l)      # fall-through blocks carry only a comment label
    if m: cur = [m.group(1), []]; blocks.append(cur); continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    cur[1].append(t)
tot = collections.Counter(); totc = 0.0
for nm, ins in blocks:
    c = collections.Counter(); cyc = 0.0; slow = collections.Counter()
    for i in ins:
        op = i.split()[0]
        if op.startswith("v_"):
            k, cl = cost(op); cyc += k; c[cl] += 1
            if cl == "slow": slow[op.replace("_e32", "").replace("_e64", "")] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        else: c["mem"] += 1
    tot.update(c); totc += cyc
    if sum(c.values()) >= 12:
        print(f"{nm:11s} valu cycles {cyc:7.1f}  fast {c['fast']:3d} med {c['med']:3d} slow {c['slow']:3d} trans {c['trans']:2d} | lds {c['lds']:2d} salu {c['salu']:3d}  slow: " + " ".join(f"{k}x{v}" for k, v in slow.most_common(7)))
print("total", dict(tot), "valu cycles", round(totc))
