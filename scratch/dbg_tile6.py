import sys, math, numpy as np, torch
sys.path.insert(0,'.')
from oracle import oracle
import cwipc_util_amd as cw
from cwipc_util_amd.capture import rotation_about_y, capture_tile
def tile(npts,t):
    a=t*2*math.pi/8
    pts,cs=oracle.synthetic(npts,a)
    pts=oracle.tilemap(pts,bytes([1<<t])*256)
    if t: pts=oracle.transform(pts,rotation_about_y(a))
    return pts,cs
inv=np.float32(1)/np.float32(0.01)
def keys(e):
    g=[np.floor(e[f]*inv).astype(np.int64) for f in 'xyz']
    return g
for npts in (300000, 2000000):
  for t in (2,6):
    pts,cs=tile(npts,t)
    pc=cw.cwipc_from_numpy_array(pts,1); pc._set_cellsize(cs)
    for variant in ('direct','tilefiltered'):
        src = pc if variant=='direct' else cw.cwipc_tilefilter(pc, 1<<t)
        got=cw.cwipc_downsample(src,0.01).get_numpy_array()
        info={}
        e,_=oracle.downsample(pts,cs,0.01,info)
        print(npts,t,variant,'hip',len(got),'oracle',len(e),info)
        if len(got)!=len(e):
            ge=keys(e); gg=keys(got)
            ke=(ge[2]+1000)*4000000+(ge[1]+1000)*2000+(ge[0]+1000)
            kg=(gg[2]+1000)*4000000+(gg[1]+1000)*2000+(gg[0]+1000)
            ue,ce=np.unique(ke,return_counts=True); ug,cg=np.unique(kg,return_counts=True)
            print('  distinct voxels: oracle',len(ue),'hip',len(ug),'same set',np.array_equal(ue,ug))
            # voxels whose multiplicity differs
            if np.array_equal(ue,ug):
                d=np.flatnonzero(ce!=cg)
                vx=ue[d]%2000-1000; vy=(ue[d]//2000)%2000-1000; vz=ue[d]//4000000-1000
                print('  multiplicity differs at',len(d),'voxels; x range',vx.min(),vx.max(),'y range',vy.min(),vy.max(),'z range',vz.min(),vz.max())
                print('  oracle mult',np.unique(ce[d],return_counts=True),'hip mult',np.unique(cg[d],return_counts=True))
                print('  sample',list(zip(vx[:6],vy[:6],vz[:6])))
            # points of the cloud in the first differing voxel
