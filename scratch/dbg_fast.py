import sys, numpy as np, torch
sys.path.insert(0,'.')
from oracle import oracle
import cwipc_util_amd as cw
cell=float(sys.argv[1]) if len(sys.argv)>1 else 0.3
pts,cs=oracle.synthetic(100000,0.7)
pc=cw.cwipc_from_numpy_array(pts,1); pc._set_cellsize(cs)
got=cw.cwipc_downsample(pc,cell).get_numpy_array()
e,_=oracle.downsample(pts,cs,cell)
print('hip',len(got),'oracle',len(e))
leaf=np.float32(max(cell,cs)); inv=np.float32(1)/leaf
def keys(a): return [np.floor(a[f]*inv).astype(np.int64) for f in 'xyz']
ge=keys(e); gg=keys(got)
se=sorted(zip(*ge)); sg=sorted(zip(*gg))
from collections import Counter
ce=Counter(se); cg=Counter(sg)
for k in sorted(set(ce)|set(cg)):
    if ce[k]!=cg[k]: print('voxel',k,'oracle',ce[k],'hip',cg[k])
print('first point',pts[0])
