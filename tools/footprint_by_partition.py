"""Developer analysis (CPU only, uses the oracle): how many 128-byte lines of the node array the eight lists of a strip schedule -- one per
XCD, each with an L2 of its own -- have to fetch on the benchmark frame, for several ways of dealing the strips to the lists.  A line that two
lists need is fetched twice; profiles/r05_footprint_by_partition.txt has the output.  The strip classes come from the oracle's step counts.
usage: gcc -O2 -ffp-contract=off -shared -fPIC -o /tmp/libfp.so tools/footprint_by_partition.c -lm -lpthread && python tools/footprint_by_partition.py"""
import sys, time, ctypes as C
import numpy as np
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import __graft_entry__ as entry
pkg = entry.load_package()
from oracle import oracle as O
from concurrent.futures import ThreadPoolExecutor
cam, look = pkg.scenes.terrain_camera(0, 16)
words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
W,H=1920,1080
u = O.make_uniforms(cam, look, 90.0, W, H, flags=0)
lib = C.CDLL('/tmp/libfp.so')
rec = O.trace_frame(words, O.make_uniforms(cam, look, 90.0, W, H), threads=8)
steps=(rec['info']&0xFF).astype(np.int32)
b=steps.reshape(H//8,8,W//8,8).transpose(0,2,1,3).reshape(-1,64)
mx=b.max(1); sm=b.sum(1)+64; bpr=W//8; ns=len(mx)
cls=np.minimum(mx>>2,31)
def lists_current():
    L=[[] for _ in range(8)]
    for c in range(31,-1,-1):
        idx=np.nonzero(cls==c)[0]; n=len(idx)
        if n==0: continue
        sl=max((n+7)//8,1)
        for l in range(8): L[l].extend(idx[min(l*sl,n):min(l*sl+sl,n)].tolist())
    return L
def lists_bands():   # contiguous in screen (row-major strip) order, equal total cost
    cum=np.cumsum(sm); tot=cum[-1]
    cuts=[0]+[int(np.searchsorted(cum, tot*k/8)) for k in range(1,8)]+[ns]
    return [list(range(cuts[l],cuts[l+1])) for l in range(8)]
def lists_interleave(run=16):
    L=[[] for _ in range(8)]
    for c in range(31,-1,-1):
        idx=np.nonzero(cls==c)[0]
        for r,s in enumerate(idx): L[(r//run)%8].append(int(s))
    return L
def footprint(L,name):
    def one(l):
        v=np.zeros(words.size,dtype=np.uint32); st=np.array(l,dtype=np.uint32)
        lib.fp_mark(words.ctypes.data_as(C.c_void_p), C.c_size_t(words.size), C.byref(u), st.ctypes.data_as(C.c_void_p), C.c_size_t(len(st)), C.c_int(bpr), v.ctypes.data_as(C.c_void_p))
        w=np.nonzero(v)[0]; return np.unique(w>>5), len(w)
    with ThreadPoolExecutor(8) as ex: res=list(ex.map(one,L))
    lines=[r[0] for r in res]; per=[len(x) for x in lines]; uni=len(np.unique(np.concatenate(lines)))
    print(f"{name}: lines per list {per} sum {sum(per)} ({sum(per)*128/1e6:.1f} MB) union {uni} ({uni*128/1e6:.1f} MB) dup {sum(per)/uni:.2f}; words touched (sum over lists) {sum(r[1] for r in res)}; work per list {[int(sm[l].sum()) for l in L]}",flush=True)
t0=time.time()
footprint(lists_current(),"current (per-class segments)")
print(time.time()-t0)
footprint(lists_bands(),"bands (equal cost)")
footprint(lists_interleave(16),"interleave run16")
def lists_cols(n=8):
    bx=np.arange(ns)%bpr
    return [np.nonzero((bx*n)//bpr==l)[0].tolist() for l in range(n)]
def lists_tiles(nx,ny):
    bx=np.arange(ns)%bpr; by=np.arange(ns)//bpr; rows=H//8
    t=((by*ny)//rows)*nx+(bx*nx)//bpr
    return [np.nonzero(t==l)[0].tolist() for l in range(nx*ny)]
footprint(lists_cols(8),"8 vertical stripes equal width")
footprint(lists_tiles(4,2),"tiles 4x2")
footprint(lists_tiles(2,4),"tiles 2x4")
def lists_colmajor(width_blocks=1):
    rows=H//8
    bx=np.arange(ns)%bpr; by=np.arange(ns)//bpr
    key=(bx//width_blocks)*(rows*width_blocks)+by*width_blocks+bx%width_blocks   # column-major over stripes of width_blocks
    order=np.argsort(key,kind='stable')
    L=[[] for _ in range(8)]
    for c in range(31,-1,-1):
        idx=order[cls[order]==c]; n=len(idx)
        if n==0: continue
        sl=max((n+7)//8,1)
        for l in range(8): L[l].extend(idx[min(l*sl,n):min(l*sl+sl,n)].tolist())
    return L
footprint(lists_colmajor(1),"per-class segments, column-major")
footprint(lists_colmajor(4),"per-class segments, column-major stripes of 4 blocks")
def lists_fold(nstripes):
    bx=np.arange(ns)%bpr
    st=(bx*nstripes)//bpr
    m=st%16
    l=np.where(m<8,m,15-m)
    return [np.nonzero(l==k)[0].tolist() for k in range(8)]
footprint(lists_fold(16),"16 stripes folded")
footprint(lists_fold(32),"32 stripes folded")
def lists_groupcol(bounds):
    rows=H//8
    bx=np.arange(ns)%bpr; by=np.arange(ns)//bpr
    order=np.argsort(bx*rows+by,kind='stable')
    grp=np.digitize(cls,bounds)   # group id per strip
    lst=np.zeros(ns,dtype=np.int64)
    for g in np.unique(grp):
        idx=order[grp[order]==g]; n=len(idx); sl=max((n+7)//8,1)
        lst[idx]=np.minimum(np.arange(n)//sl,7)
    L=[]
    for l in range(8):
        mine=np.nonzero(lst==l)[0]
        # classes descending, col-major within
        key=(31-cls[mine])*ns + (bx[mine]*rows+by[mine])
        L.append(mine[np.argsort(key,kind='stable')].tolist())
    return L
def classmix(L,name):
    print(name,"class>=15 per list",[int((cls[l]>=15).sum()) for l in L],"class 8..14",[int(((cls[l]>=8)&(cls[l]<15)).sum()) for l in L])
for b in ([8],[6,10,15],[4,6,8,12,15]):
    L=lists_groupcol(b); footprint(L,f"group-col {b}"); classmix(L,"   ")
# coarser classes for the short strips (one sweep of a slab instead of four)
cls_orig = cls.copy()
for name, f in (("classes 4..7 merged", lambda c: np.where((c >= 4) & (c < 8), 4, c)), ("classes 2..7 merged", lambda c: np.where((c >= 2) & (c < 8), 2, c)),
                ("classes 4..7 and 8..11 merged", lambda c: np.where((c >= 4) & (c < 8), 4, np.where((c >= 8) & (c < 12), 8, c)))):
    cls = f(cls_orig)
    footprint(lists_colmajor(1), "column-major, " + name)
cls = cls_orig
# one common set of cuts for the classes below `lo`, per-class cuts from `lo` up (what strip_order_kernel can do with its rank compare)
def lists_common_below(lo):
    rows=H//8
    bx=np.arange(ns)%bpr; by=np.arange(ns)//bpr
    pos=bx*rows+by
    order=np.argsort(pos,kind='stable')
    lst=np.zeros(ns,dtype=np.int64)
    low=order[cls[order]<lo]; n=len(low); sl=max((n+7)//8,1); lst[low]=np.minimum(np.arange(n)//sl,7)
    for c in range(lo,32):
        idx=order[cls[order]==c]; n=len(idx)
        if n: sl=max((n+7)//8,1); lst[idx]=np.minimum(np.arange(n)//sl,7)
    L=[]
    for l in range(8):
        mine=np.nonzero(lst==l)[0]
        key=(31-cls[mine])*ns + pos[mine]
        L.append(mine[np.argsort(key,kind='stable')].tolist())
    return L
for lo in (15, 12, 8):
    L=lists_common_below(lo); footprint(L,f"common cuts below class {lo}")
    print("    per list: class>=15", [int((cls[l]>=15).sum()) for l in L], "class 8..14", [int(((cls[l]>=8)&(cls[l]<15)).sum()) for l in L], "class 4..7", [int(((cls[l]>=4)&(cls[l]<8)).sum()) for l in L], "class<4", [int((cls[l]<4).sum()) for l in L])
