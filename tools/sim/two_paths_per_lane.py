"""Model of the decoupled fast-tree loop with ONE path per lane (what is built: traverse until 12/16 of the lanes that entered have finished, shade them, refill) against
TWO paths per lane (a lane whose ray has finished parks that path for shading and continues with its second path's ray; shading runs when 75 % of the lanes have a parked path):
wave-level steps of unit cost, ray lengths gamma distributed with the hall's mean. The one-path model reproduces the measured lane utilisations (traversal 0.62, shading 0.76).
Result (cost per ray, node step 110 / shading pass 800 / switch pass 100 wave instructions): one path 70.1; two paths 63.3 at best (switch after 16 finished lanes) -- a 10 % gain that
needs the second path's ~29 registers, i.e. 4 waves per SIMD instead of 5 (measured on round 3's kernel: -14 %). Not built; DESIGN.md section 14. python tools/sim/two_paths_per_lane.py"""
import numpy as np
rs=np.random.RandomState(1)
W=64
def raylen():
    # node+tri steps per ray: gamma with mean 20, shape 2.5 (long tail)
    return max(1,int(rs.gamma(2.5, 8.0)))
def sim_single(exit_frac=0.75, n_rays=200000):
    # current policy: one slot per lane; traverse until exit_frac of lanes that entered have finished; then shade those (one pass, cost S) and refill
    rem=np.array([raylen() for _ in range(W)])
    steps=0; lane_steps=0; shade_pass=0; shade_lanes=0; done=0
    while done<n_rays:
        entered=(rem>0).sum(); target=entered-int(entered*exit_frac) if entered else 0
        target=min(target, entered-1) if entered>0 else 0
        while (rem>0).sum()>target:
            act=rem>0; steps+=1; lane_steps+=act.sum(); rem[act]-=1
        fin=rem==0
        shade_pass+=1; shade_lanes+=fin.sum(); done+=fin.sum()
        rem[fin]=[raylen() for _ in range(fin.sum())]
    return lane_steps/(steps*W), shade_lanes/(shade_pass*W), steps/done, shade_pass/done
def sim_double(thr=0.75, n_rays=200000):
    # two slots: P traverses; Q parked: 0 ready(has ray), 1 finished(waiting shade). shade Q when lanes with finished Q >= thr*W (or nobody can traverse)
    P=np.array([raylen() for _ in range(W)]); Qready=np.array([raylen() for _ in range(W)]); Qfin=np.zeros(W,bool)
    steps=0; lane_steps=0; shade_pass=0; shade_lanes=0; done=0; swaps=0
    while done<n_rays:
        # switch: lanes with P finished and Q ready(not fin) swap
        sw=(P==0)&(~Qfin)
        if sw.any():
            P[sw]=Qready[sw]; Qfin[sw]=True; swaps+=1
        trav=P>0
        nfin=Qfin.sum()
        if nfin>=thr*W or not trav.any():
            shade_pass+=1; shade_lanes+=nfin; done+=nfin
            Qready[Qfin]=[raylen() for _ in range(nfin)]; Qfin[:]=False
            continue
        # traverse until some lane finishes such that a decision point arrives: step until number of lanes with P==0 increases by >= 8 (batch switch) 
        idle0=(P==0).sum()
        while True:
            act=P>0
            if not act.any(): break
            steps+=1; lane_steps+=act.sum(); P[act]-=1
            if (P==0).sum()-idle0>=8: break
    return lane_steps/(steps*W), shade_lanes/(shade_pass*W), steps/done, shade_pass/done, swaps/done
for f in (0.5,0.75,0.9): print('single exit',f, [round(x,3) for x in sim_single(f)])
for t in (0.5,0.75,0.9): print('double thr',t, [round(x,3) for x in sim_double(t)])
# cost model: node step 119, shade 800
for name,res in (('single .75',sim_single(0.75)),('double .75',sim_double(0.75)),('double .9',sim_double(0.9)),('double .5',sim_double(0.5))):
    print(name,'cost per ray', round(res[2]*110+res[3]*800,1))
print('--- with switch cost')
def sim_double2(thr=0.75, k=8, n_rays=200000, SW=100, NODE=110, SHADE=800):
    P=np.array([raylen() for _ in range(W)]); Qready=np.array([raylen() for _ in range(W)]); Qfin=np.zeros(W,bool)
    cost=0; done=0
    while done<n_rays:
        sw=(P==0)&(~Qfin)
        if sw.any():
            P[sw]=Qready[sw]; Qfin[sw]=True; cost+=SW
        trav=P>0
        nfin=Qfin.sum()
        if nfin>=thr*W or not trav.any():
            cost+=SHADE; done+=nfin
            Qready[Qfin]=[raylen() for _ in range(nfin)]; Qfin[:]=False
            continue
        idle0=(P==0).sum()
        while True:
            act=P>0
            if not act.any(): break
            cost+=NODE; P[act]-=1
            if (P==0).sum()-idle0>=k: break
    return cost/done
def sim_single2(exit_frac=0.75, n_rays=200000, NODE=110, SHADE=800):
    rem=np.array([raylen() for _ in range(W)]); cost=0; done=0
    while done<n_rays:
        entered=(rem>0).sum(); target=entered-int(entered*exit_frac)
        while (rem>0).sum()>target:
            act=rem>0; cost+=NODE; rem[act]-=1
        fin=rem==0; cost+=SHADE; done+=fin.sum()
        rem[fin]=[raylen() for _ in range(fin.sum())]
    return cost/done
print('single', round(sim_single2(),1))
for k in (4,8,16,24,32):
    for thr in (0.6,0.75,0.9):
        print('double k',k,'thr',thr, round(sim_double2(thr,k),1))
