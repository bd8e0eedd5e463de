"""Developer analysis of tools/wave_timeline.py --raw dumps: when the waves of every XCD (= schedule list) ended, and how much of the spread
is inside a workgroup, between workgroups, between compute units.  usage: python tools/timeline_by_xcd.py dump.npz"""
import numpy as np, sys
z=np.load(sys.argv[1]); d=z['d']; ev=z['ev']; wave=z['wave']
t0=d[:,0].min()
start=(d[:,0]-t0)*0.01; dry=(d[:,1]-t0)*0.01; end=(d[:,2]-t0)*0.01
hw=ev[:,15]; xcc=(hw>>16)&0xF; hwid=hw&0xFFFF
cu=(hwid>>8)&0xF; sh=(hwid>>12)&1; se=(hwid>>13)&7; simd=(hwid>>4)&3
blk=wave//4; lst=blk%8
print("waves",len(d),"xcc vals",np.unique(xcc),"se",np.unique(se),"sh",np.unique(sh),"cu",np.unique(cu))
print("xcc==blk%8 ?", (xcc==lst).mean())
def stats(name,key):
    print(name)
    for k in np.unique(key):
        m=key==k
        print(f"  {k}: n {m.sum():5d} end mean {end[m].mean():6.1f} p5 {np.percentile(end[m],5):6.1f} p95 {np.percentile(end[m],95):6.1f} max {end[m].max():6.1f} dry mean {dry[m].mean():6.1f} rounds {d[m,3].mean():6.1f} gens {(ev[m,1:15]!=0).sum(1).mean():.2f}")
stats("by xcc",xcc)
# variance decomposition: within workgroup vs between
import collections
wg_end=collections.defaultdict(list)
for b,e in zip(blk,end): wg_end[b].append(e)
within=np.mean([np.std(v) for v in wg_end.values() if len(v)==4]); between=np.std([np.mean(v) for v in wg_end.values()])
print("std of end: total",end.std(),"within workgroup (mean of std)",within,"between workgroup means",between)
# by CU (xcc,se,sh,cu)
cukey=xcc*1000+se*100+sh*10*0+cu   # sh may be unused
cu_end=collections.defaultdict(list)
for k,e in zip(cukey,end): cu_end[k].append(e)
print("n CUs",len(cu_end),"waves per CU",np.mean([len(v) for v in cu_end.values()]))
print("within CU std",np.mean([np.std(v) for v in cu_end.values()]),"between CU means std",np.std([np.mean(v) for v in cu_end.values()]))
cm=np.array([np.max(v) for v in cu_end.values()]); print("CU max-end percentiles",np.percentile(cm,[0,25,50,75,100]))
# last waves
o=np.argsort(end)[-30:]
for i in o: print("late wave",wave[i],"xcc",xcc[i],"se",se[i],"cu",cu[i],"simd",simd[i],"end",round(end[i],1),"dry",round(dry[i],1),"rounds",d[i,3],"gens",(ev[i,1:15]!=0).sum(),"lastgen",round((ev[i,1:15].max()-t0)*0.01,1))
