import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import yasph2d_amd as y
from yasph2d_amd.tiles import *
from util import dam_break
pos, boundary = dam_break(1.0)
cuts = quantile_cuts(cell_coord(pos, 1), 2)
print("cuts", cuts)
ctxs=[y.SphxContext(), y.SphxContext()]
bs=[GpuTileBackend(c) for c in ctxs]
tiles=[]
for r in range(2):
    b=bs[r]; lo,hi=cuts[r],cuts[r+1]
    c=cell_coord(pos,1); mine=(c>=lo)&(c<hi)
    b.configure(1, lo, hi, 16, r>0, r<1)
    cap=4096
    b.reserve(int(mine.sum()*1.25)+4*cap+4096)
    bc=cell_coord(boundary,1); keep=(bc+18>=lo)&(bc<hi+18)
    b.set_boundary(boundary[keep])
    b.upload(pos[mine], np.zeros_like(pos[mine]), np.arange(len(pos),dtype=np.uint32)[mine])
    tiles.append((b, b.make_buffers(cap), cap))
def refresh():
    for r in range(2):
        b,(sl,sr,rl,rr),cap=tiles[r]
        b.pack(sl,sr,cap)
        for nm,t in (("L",sl),("R",sr)):
            h=t.cpu().numpy().view(HALO_DTYPE); n=h['id'][0]
            if n: 
                cc=cell_coord(h['pv'][1:1+n,:2].copy(),1); print(" rank",r,"send",nm,n,"cells",cc.min(),cc.max())
    tiles[0][1][3].copy_(tiles[1][1][0]); tiles[1][1][2].copy_(tiles[0][1][1]); torch.cuda.synchronize()
    for r in range(2):
        b,(sl,sr,rl,rr),cap=tiles[r]
        b.apply(rl if r>0 else None, rr if r<1 else None, cap)
        n=b.regrid()
        d=b.download(); cc=cell_coord(d["pos"],1); own=d["owned"]
        print(" rank",r,"n_local",n,"owned",own.sum(),"owned cells",cc[own].min(),cc[own].max(),"ghost cells",cc[~own].min(),cc[~own].max(), "uniq ids", len(np.unique(d["ids"])))
refresh()
for r in range(2):
    tiles[r][0].nonpressure(4.1667e-5); tiles[r][0].predict(8e-5); tiles[r][0].iteration(False, 8e-5, True); tiles[r][0].advect(8e-5)
refresh()
print("---- direct apply check")
b,(sl,sr,rl,rr),cap=tiles[0]
b.pack(sl,sr,cap); tiles[1][0].pack(*tiles[1][1][:2],cap)
rr.copy_(tiles[1][1][0]); torch.cuda.synchronize()
h=rr.cpu().numpy().view(HALO_DTYPE); print("rr header", h['id'][0], "first rec", h[1])
n0=b.ctx.n
b.apply(None, rr, cap)
d=b.ctx.download()
print("N after apply", b.ctx.n, "n0", n0)
seg=d["pos"][n0+cap:n0+cap+5]; print("right region first pos", seg, "ids", d["ids"][n0+cap:n0+cap+5])
seg=d["pos"][n0:n0+3]; print("left region first pos", seg)
