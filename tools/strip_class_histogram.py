"""Developer analysis: cost classes of the 8x8 strips of a frame traced by the oracle (saved by an earlier script as /tmp/frame_WxH.npz)."""
import numpy as np
z=np.load('/tmp/frame_1920x1080.npz'); info=z['info']; st=z['st']
steps=(info&0xFF).astype(np.int32); H,W=steps.shape
# descents per ray = steps+1 for rays that entered (all inside cube here)
b=steps.reshape(H//8,8,W//8,8).transpose(0,2,1,3).reshape(-1,64)
mx=b.max(1); sm=b.sum(1)+64
cls=np.minimum(mx>>2,15)
print("strips",len(mx),"total ray-descents",sm.sum())
cum=0
for c in range(15,-1,-1):
    m=cls==c
    cum+=m.sum()
    print(f"class {c:2d}: strips {m.sum():6d} cum {cum:6d}  mean max {mx[m].mean() if m.any() else 0:6.1f}  mean sum/64 {sm[m].mean()/64 if m.any() else 0:6.1f}  work share {sm[m].sum()/sm.sum():.3f}")
# words per descent
print("restart words/ray",st[...,0].mean(),"stack words/ray",st[...,1].mean())
