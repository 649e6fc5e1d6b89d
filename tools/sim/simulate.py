"""Lane-scheduling simulator for the LDS-resident render kernel (dev tool; see collect_traces.py).
Compares the shipped coupled loop (all lanes trace one ray, then shade together) with a unified state machine that
executes ONE step type per wave iteration, chosen by vote."""
import pickle, sys
import numpy as np

COST = dict(node=62, leaf=60, shade=600, cam=150, init=40, iter_ovh=30, sched=14)

def lane_programs(lanes):
    return lanes   # list (64) of list of (nodes, tris, hit, is_cam)

def coupled(lanes):
    """current kernel: per outer iteration every live lane has one ray"""
    idx = [0] * 64; n = [len(s) for s in lanes]
    valu = 0; it = dict(ray=0, node=0, leaf=0, shade=0, cam=0); lanes_sum = dict(node=0, leaf=0, shade=0, cam=0, ray=0)
    while True:
        live = [l for l in range(64) if idx[l] < n[l]]
        if not live: break
        rays = [lanes[l][idx[l]] for l in live]
        it["ray"] += 1; lanes_sum["ray"] += len(live)
        if any(r[3] for r in rays): it["cam"] += 1; valu += COST["cam"]; lanes_sum["cam"] += sum(r[3] for r in rays)
        mn = max(r[0] for r in rays); ml = max(r[1] for r in rays)
        it["node"] += mn; it["leaf"] += ml; valu += mn * COST["node"] + ml * COST["leaf"] + COST["init"] + COST["iter_ovh"]
        lanes_sum["node"] += sum(r[0] for r in rays); lanes_sum["leaf"] += sum(r[1] for r in rays)
        if any(r[2] for r in rays): it["shade"] += 1; valu += COST["shade"]; lanes_sum["shade"] += sum(r[2] for r in rays)
        for l in live: idx[l] += 1
    return valu, it, lanes_sum

def unified(lanes, policy):
    """state machine: per lane (ray index, nodes left, leaves pending, leaves not yet discovered). One step type per iteration.
    Leaves are discovered evenly along the node steps. Types: N node, L leaf, S = ray finished (shade if hit; cam if path ended; init next ray)."""
    n = [len(s) for s in lanes]
    idx = [0] * 64
    nodes_left = [0] * 64; leaf_pend = [0] * 64; leaf_future = [0] * 64; nodes_total = [1] * 64; leaves_total = [0] * 64
    state = ["S"] * 64      # S: needs a new ray (before the first ray: cam)
    started = [False] * 64
    valu = 0; it = dict(N=0, L=0, S=0); act = dict(N=0, L=0, S=0, shade_exec=0, cam_exec=0)
    def start_ray(l):
        r = lanes[l][idx[l]]
        nodes_left[l] = r[0]; nodes_total[l] = max(1, r[0]); leaves_total[l] = r[1]; leaf_pend[l] = 0; leaf_future[l] = r[1]
    while True:
        readyN = [l for l in range(64) if state[l] == "T" and nodes_left[l] > 0 and leaf_pend[l] <= policy.get("leaf_cap", 14)]
        readyL = [l for l in range(64) if state[l] == "T" and leaf_pend[l] > 0]
        readyS = [l for l in range(64) if state[l] == "S" and idx[l] < n[l] or (state[l] == "S" and started[l] and idx[l] == n[l] - 0 and False)]
        # lanes that finished traversal (state T, nothing left) become S
        for l in range(64):
            if state[l] == "T" and nodes_left[l] == 0 and leaf_pend[l] == 0 and leaf_future[l] == 0:
                state[l] = "F"      # finished ray, waiting for the S step
        readyS = [l for l in range(64) if state[l] == "F" or (state[l] == "S" and idx[l] < n[l])]
        if not readyN and not readyL and not readyS: break
        choice = policy["choose"](len(readyN), len(readyL), len(readyS))
        valu += COST["sched"]
        if choice == "N":
            it["N"] += 1; act["N"] += len(readyN); valu += COST["node"]
            for l in readyN:
                nodes_left[l] -= 1
                done = nodes_total[l] - nodes_left[l]
                disc = (leaves_total[l] * done) // nodes_total[l]        # leaves discovered so far
                newly = disc - (leaves_total[l] - leaf_future[l])
                leaf_future[l] -= newly; leaf_pend[l] += newly
        elif choice == "L":
            it["L"] += 1; act["L"] += len(readyL); valu += COST["leaf"]
            for l in readyL: leaf_pend[l] -= 1
        else:
            it["S"] += 1; act["S"] += len(readyS)
            shade = [l for l in readyS if state[l] == "F" and lanes[l][idx[l]][2]]
            if shade: valu += COST["shade"]; act["shade_exec"] += 1
            need_cam = False
            for l in readyS:
                if state[l] == "F": idx[l] += 1
                if idx[l] < n[l]:
                    if lanes[l][idx[l]][3]: need_cam = True
                    start_ray(l); state[l] = "T"; started[l] = True
                else:
                    state[l] = "D"
            if need_cam: valu += COST["cam"]; act["cam_exec"] += 1
            valu += COST["init"]
    return valu, it, act

def greedy(wN=1.0, wL=1.0, wS=1.0, s_min=0):
    def choose(n, l, s):
        if s and s >= s_min and s * wS >= n * wN and s * wS >= l * wL: return "S"
        if n == 0 and l == 0: return "S"
        return "N" if n * wN >= l * wL else "L"
    return choose

if __name__ == "__main__":
    tr = pickle.load(open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/cornell_traces.pkl", "rb"))
    inbox = [k for k, v in tr.items() if sum(len(s) for s in v) > 64 * len(v[0]) * 0 + 64 * 70]
    print("in-box packets", len(inbox))
    tot = dict()
    def add(name, v): tot[name] = tot.get(name, 0) + v
    rays = 0
    for k in inbox:
        lanes = tr[k]; rays += sum(len(s) for s in lanes)
        v, it, ls = coupled(lanes); add("coupled", v)
        for kk in it: add("c_it_" + kk, it[kk])
        for name, pol in [("greedy", dict(choose=greedy())), ("greedy_s32", dict(choose=greedy(s_min=32))), ("greedy_s24_w", dict(choose=greedy(wS=0.5, s_min=24))),
                          ("greedy_s40", dict(choose=greedy(s_min=40))), ("greedy_s48", dict(choose=greedy(s_min=48))), ("greedy_wS2", dict(choose=greedy(wS=2.0))),
                          ("greedy_s32_cap6", dict(choose=greedy(s_min=32), leaf_cap=6))]:
            v, it, act = unified(lanes, pol); add(name, v)
            for kk in it: add(name + "_it_" + kk, it[kk]); 
            for kk in act: add(name + "_act_" + kk, act[kk])
    print("rays", rays, "ideal lane-steps: VALU if perfectly packed", )
    print("coupled VALU/ray-lane %.1f  per ray-iter: node %.2f leaf %.2f shade %.2f cam %.2f" % (tot["coupled"] / rays * 1.0, tot["c_it_node"] / tot["c_it_ray"], tot["c_it_leaf"] / tot["c_it_ray"], tot["c_it_shade"] / tot["c_it_ray"], tot["c_it_cam"] / tot["c_it_ray"]))
    for name in ("greedy", "greedy_s32", "greedy_s24_w", "greedy_s40", "greedy_s48", "greedy_wS2", "greedy_s32_cap6"):
        print("%-16s VALU ratio vs coupled %.3f   util N %.2f L %.2f S %.2f  iters N %d L %d S %d (shade execs %d cam %d)" % (name, tot[name] / tot["coupled"],
              tot[name + "_act_N"] / (64 * tot[name + "_it_N"]), tot[name + "_act_L"] / (64 * tot[name + "_it_L"]), tot[name + "_act_S"] / (64 * tot[name + "_it_S"]),
              tot[name + "_it_N"], tot[name + "_it_L"], tot[name + "_it_S"], tot[name + "_act_shade_exec"], tot[name + "_act_cam_exec"]))


def dual(lanes2, policy, cost=COST, park=40, act=30):
    """two paths per lane. lanes2: list of 64 pairs of ray sequences. Per lane: slots 0/1, each in state
    'R' ready (ray parked, not started), 'T' traversing (the lane's active path), 'P' pending shade, 'D' done.
    Steps: N, L (active path), S (shade one pending path per lane -> R; also starts the first ray), A (activate a ready path when no path is active)."""
    nl = len(lanes2)
    seq = [[lanes2[l][0], lanes2[l][1]] for l in range(nl)]
    idx = [[0, 0] for _ in range(nl)]
    st = [["P0", "P0"] for _ in range(nl)]       # P0: needs its first ray (camera) -- handled by the S step
    active = [-1] * nl
    nodes_left = [0] * nl; leaf_pend = [0] * nl; leaf_future = [0] * nl; nodes_total = [1] * nl; leaves_total = [0] * nl
    valu = 0; it = dict(N=0, L=0, S=0, A=0); act_l = dict(N=0, L=0, S=0, A=0, shade_exec=0, cam_exec=0)
    while True:
        for l in range(nl):
            a = active[l]
            if a >= 0 and nodes_left[l] == 0 and leaf_pend[l] == 0 and leaf_future[l] == 0:
                st[l][a] = "P"; active[l] = -1
        readyN = [l for l in range(nl) if active[l] >= 0 and nodes_left[l] > 0]
        readyL = [l for l in range(nl) if active[l] >= 0 and leaf_pend[l] > 0]
        readyS = [l for l in range(nl) if any(s in ("P", "P0") for s in st[l])]
        readyA = [l for l in range(nl) if active[l] < 0 and any(s == "R" for s in st[l])]
        blocked = [l for l in range(nl) if active[l] < 0 and not any(s == "R" for s in st[l]) and any(s in ("P", "P0") for s in st[l])]
        if not (readyN or readyL or readyS or readyA): break
        choice = policy(len(readyN), len(readyL), len(readyS), len(readyA), len(blocked))
        valu += cost["sched"]
        if choice == "N":
            it["N"] += 1; act_l["N"] += len(readyN); valu += cost["node"]
            for l in readyN:
                nodes_left[l] -= 1
                done = nodes_total[l] - nodes_left[l]
                disc = (leaves_total[l] * done) // nodes_total[l]
                newly = disc - (leaves_total[l] - leaf_future[l])
                leaf_future[l] -= newly; leaf_pend[l] += newly
        elif choice == "L":
            it["L"] += 1; act_l["L"] += len(readyL); valu += cost["leaf"]
            for l in readyL: leaf_pend[l] -= 1
        elif choice == "A":
            it["A"] += 1; act_l["A"] += len(readyA); valu += act
            for l in readyA:
                k = 0 if st[l][0] == "R" else 1
                r = seq[l][k][idx[l][k]]
                st[l][k] = "T"; active[l] = k
                nodes_left[l] = r[0]; nodes_total[l] = max(1, r[0]); leaves_total[l] = r[1]; leaf_pend[l] = 0; leaf_future[l] = r[1]
        else:
            it["S"] += 1; act_l["S"] += len(readyS)
            any_shade = False; any_cam = False
            for l in readyS:
                k = 0 if st[l][0] in ("P", "P0") else 1
                if st[l][k] == "P":
                    if seq[l][k][idx[l][k]][2]: any_shade = True
                    idx[l][k] += 1
                if idx[l][k] < len(seq[l][k]):
                    if seq[l][k][idx[l][k]][3]: any_cam = True
                    st[l][k] = "R"
                else:
                    st[l][k] = "D"
            if any_shade: valu += cost["shade"]; act_l["shade_exec"] += 1
            if any_cam: valu += cost["cam"]; act_l["cam_exec"] += 1
            valu += park
    return valu, it, act_l

def dual_policy(s_min=40, wS=1.0):
    def choose(n, l, s, a, blocked):
        if n == 0 and l == 0 and a == 0: return "S"
        if s >= s_min or blocked >= 24: 
            return "S"
        best = max((n, "N"), (l, "L"), (a * 2, "A"))
        return best[1]
    return choose

if __name__ == "__main__":
    import itertools
    keys = inbox
    tot2 = {}
    rays2 = 0
    base = 0
    for k in keys:
        lanes = tr[k]
        # pair each lane's trace with a second trace: the same pixel's other half (split the sequence of samples in two halves)
        pairs = []
        for s in lanes:
            # split at a camera-ray boundary near the middle
            cams = [i for i, r in enumerate(s) if r[3]]
            mid = cams[len(cams) // 2]
            pairs.append((s[:mid], s[mid:]))
        rays2 += sum(len(s) for s in lanes)
        v, _, _ = coupled(lanes); base += v
        for name, pol in [("dual_s32", dual_policy(32)), ("dual_s40", dual_policy(40)), ("dual_s48", dual_policy(48)), ("dual_s56", dual_policy(56))]:
            v, it, a = dual(pairs, pol)
            tot2[name] = tot2.get(name, 0) + v
            for kk in it: tot2[name + "_it_" + kk] = tot2.get(name + "_it_" + kk, 0) + it[kk]
            for kk in a: tot2[name + "_act_" + kk] = tot2.get(name + "_act_" + kk, 0) + a[kk]
    for name in ("dual_s32", "dual_s40", "dual_s48", "dual_s56"):
        print("%-10s VALU ratio vs coupled %.3f  util N %.2f L %.2f S %.2f A %.2f  iters N %d L %d S %d A %d shade_exec %d cam_exec %d" % (name, tot2[name] / base,
              tot2[name + "_act_N"] / (64 * tot2[name + "_it_N"]), tot2[name + "_act_L"] / (64 * tot2[name + "_it_L"]), tot2[name + "_act_S"] / (64 * tot2[name + "_it_S"]), tot2[name + "_act_A"] / (64 * max(1, tot2[name + "_it_A"])),
              tot2[name + "_it_N"], tot2[name + "_it_L"], tot2[name + "_it_S"], tot2[name + "_it_A"], tot2[name + "_act_shade_exec"], tot2[name + "_act_cam_exec"]))
