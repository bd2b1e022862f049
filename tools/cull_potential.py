"""How much of the multi-frame sweep could a conservative DEPTH cull skip?  (round 5; CPU only, numpy)

47 % of the sweep's voxel visits update nothing (most of them lie more than sdf_trunc behind the surface).  A visit can only be
skipped before its projection if a whole wave's worth of voxels is known to be rejected: the box of a wave task (4 x 16 x 16
voxels), or of one of its gather groups (4 x 16 x 4), lies entirely behind  max depth over the box's pixel footprint + sdf_trunc.
This script counts, on frames of the bench's stream, the share of visits such a test would remove - with the exact footprint
maximum and with the maxima of 8 x 8 / 16 x 16 pixel tiles a pack pass could produce.  Result (profiles/r05/cull_potential.txt):
5 - 13 % of the visits, not 47: the rejected region is a thin, tilted slab and a box 8 cm long in y almost always reaches into
the accepted side somewhere.  Not built.
usage: python tools/cull_potential.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
s, depth, rgb, T = bench.load_frames("synthetic_640x480_5mm", 32)
fx,fy,cx,cy = s.intrinsics
W,H = s.width, s.height
VOXEL, TRUNC, DT = bench.VOXEL, bench.SDF_TRUNC, bench.DEPTH_TRUNC
UNIT = VOXEL*16
print(W,H,VOXEL,TRUNC,DT, T.shape)
def touched(d, T_cw):
    T_wc = np.linalg.inv(T_cw)
    ii, jj = np.mgrid[0:H:4, 0:W:4]
    dd = d[ii,jj].astype(np.float64)
    ok = (dd>0)&(dd<DT)
    p_cam = np.stack([(jj-cx)*dd/fx,(ii-cy)*dd/fy,dd],-1)[ok]
    p = p_cam@T_wc[:3,:3].T + T_wc[:3,3]
    lo = np.floor((p-TRUNC)/UNIT).astype(np.int64); hi=np.floor((p+TRUNC)/UNIT).astype(np.int64)
    out=[]
    for c in range(8):
        pick=np.array([(c>>a)&1 for a in range(3)],bool)
        out.append(np.where(pick,hi,lo))
    return np.unique(np.concatenate(out),axis=0)
def tilemax(d, ts):
    dv = np.where((d>0)&(d<DT), d, 0.0)
    Hh, Ww = H//ts, W//ts
    return dv[:Hh*ts,:Ww*ts].reshape(Hh,ts,Ww,ts).max((1,3))
tot=0; res={}
for f in range(0,32,4):
    d = depth[f].astype(np.float64); Tcw = T[f].astype(np.float64)
    U = touched(d,Tcw)
    R,t = Tcw[:3,:3], Tcw[:3,3]
    # boxes: task cg (x 4 voxels), group g (4 z)
    for name,(nx,nz,ts) in {"task_t16":(4,1,16),"group_t16":(4,4,16),"group_t8":(4,4,8),"group_exact":(4,4,1), "task_exact":(4,1,1), "unit_t16":(1,1,16)}.items():
        tm = tilemax(d, ts)
        dx = 16//nx; dz = 16//nz
        cgs, gs = np.meshgrid(np.arange(nx), np.arange(nz), indexing='ij')
        base = U[:,None,None,:]*UNIT + np.stack([cgs*dx*VOXEL, np.zeros_like(cgs), gs*dz*VOXEL],-1)[None] + VOXEL/2  # first voxel centre
        ext = np.array([(dx-1)*VOXEL, 15*VOXEL, (dz-1)*VOXEL])
        corners = np.stack([base + ext*np.array([(c&1),(c>>1)&1,(c>>2)&1]) for c in range(8)],-2)  # [U,nx,nz,8,3]
        pc = corners@R.T + t
        z = pc[...,2]
        zmin = z.min(-1)
        valid = zmin>0.05
        u = pc[...,0]*fx/np.maximum(z,1e-6)+cx+0.5; v = pc[...,1]*fy/np.maximum(z,1e-6)+cy+0.5
        u0 = np.clip(np.floor(u.min(-1))-1,0,W-1).astype(int); u1=np.clip(np.floor(u.max(-1))+1,0,W-1).astype(int)
        v0 = np.clip(np.floor(v.min(-1))-1,0,H-1).astype(int); v1=np.clip(np.floor(v.max(-1))+1,0,H-1).astype(int)
        out_of_img = (u.max(-1)<0)|(u.min(-1)>W)|(v.max(-1)<0)|(v.min(-1)>H)
        # max over rect of tilemax
        sh = zmin.shape
        dmax = np.zeros(sh)
        it = np.nditer(zmin, flags=['multi_index'])
        a0=(u0//ts); a1=np.minimum(u1//ts, tm.shape[1]-1); b0=(v0//ts); b1=np.minimum(v1//ts, tm.shape[0]-1)
        # if rect extends past the tiled area (H not multiple) fine for 480/16
        for idx in np.ndindex(sh):
            dmax[idx] = tm[b0[idx]:b1[idx]+1, a0[idx]:a1[idx]+1].max()
        cull = valid & ((zmin > dmax + TRUNC + 1e-3) | out_of_img)
        r = res.setdefault(name,[0,0]); r[0]+=cull.sum()*(4096//(nx*nz)); r[1]+=cull.size*(4096//(nx*nz))
for k,(a,b) in res.items(): print(k, "culled visit fraction", a/b)
