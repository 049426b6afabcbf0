"""Simulation of how k_wave_pairs<levenshtein> deals cfg5's rows into batches: what share of the lane-steps holds a cell, and where
the rest goes (jobs shorter than their batch, empty lanes, rounding to the block height, the diagonal skew).
  python bench_support/sim_lev_batches.py          # 32 / 64 rows per lane, 64 ... 512 rows ranked together
"""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, workload as w
n=64*3000
la=np.empty(n,dtype=np.uint32); lb=np.empty(n,dtype=np.uint32)
w.lib().synth_lengths_host(5, w.UNIFORM, 1, 1024, 0, n, la.ctypes.data, lb.ctypes.data)
def sim(BW, keyf, accept=None, GROUP=64, JOBS=16, ARENA=8192, PAD=48, LANES=64, verbose=False):
    tot=dict(cells=0, area=0, mismatch=0, empty=0, pad=0, skew=0); nb=0
    st={'jobs':[], 'lanes':0,'used':0,'T':0}
    def flush():
        nonlocal nb
        if not st['jobs']: return
        T=st['T']; nb+=1
        tot['area']+=LANES*T*BW
        tot['empty']+=(LANES-st['lanes'])*T*BW
        for (B,mx,mn) in st['jobs']:
            tot['mismatch']+=B*(T-(mx+B-1))*BW
            tot['skew']+=B*(B-1)*BW
            tot['pad']+=(B*BW-mn)*mx
            tot['cells']+=mx*mn
        st.update(jobs=[],lanes=0,used=0,T=0)
    for ci,c0 in enumerate(range(0,n,GROUP)):
        a=la[c0:c0+GROUP]; b=lb[c0:c0+GROUP]
        mx=np.maximum(a,b).astype(int); mn=np.minimum(a,b).astype(int)
        B=(mn+BW-1)//BW
        order=np.argsort(-keyf(mx,mn,B),kind='stable')
        todo=list(order)
        while todo:
            rest=[]
            for r in todo:
                need=B[r]
                if st['lanes']+need>LANES: rest.append(r); continue
                if accept and st['jobs'] and not accept(st['T'], mx[r]+need-1): rest.append(r); continue
                slot=(2*PAD+mx[r]+3)&~3
                if len(st['jobs'])==JOBS or st['used']+slot>ARENA: flush()
                st['jobs'].append((need,mx[r],mn[r])); st['lanes']+=need; st['used']+=slot; st['T']=max(st['T'],mx[r]+need-1)
                if len(st['jobs'])==JOBS or st['lanes']==LANES: flush()
            if rest: flush()
            todo=rest
    flush()
    a=tot['area']
    return {k: round(v/a,4) for k,v in tot.items()}, nb
if __name__ == "__main__":
    key = lambda mx, mn, B: mx + (mn + 31) // 32
    for bw in (32, 64):
        for group in (64, 128, 256, 512):
            print("rows per lane", bw, "rows ranked together", group, sim(bw, key, GROUP=group))
