#!/usr/bin/env python3
"""How much of the target does ONE workgroup of the NN kernels need?  (design study for the LDS-staged search of round 4 -- nn_stage, measured slower and removed: commit 09fbe87 has it; CPU only)
For pairs of the configs[2] fragment set: the 256 consecutive cell-sorted source points of every workgroup, transformed by the pair's
guess, and the union of their 27-cell neighbourhoods in the target grid: distinct (y, z) rows, staged points, staged cell bounds.
Result on the 250 k-point fragments: <= 185 rows, <= 930 points, <= 890 cell bounds per workgroup -- a dense (y, z) table over the
bounding rectangle of the rows would NOT fit (p90 of 1200-3600 entries: consecutive points jump between surfaces), a hash table does."""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from elasticreconstruction_amd import synth
n_frag=25
frs = synth.fragment_set(n_frag, 250000)
def grid(x, cell):
    lo = x.min(0); hi = x.max(0)
    dim = np.floor((hi-lo)/cell).astype(int)+1
    q = np.clip(np.floor((x-lo)/cell).astype(int), 0, dim-1)
    c = (q[:,2]*dim[1]+q[:,1])*dim[0]+q[:,0]
    order = np.argsort(c, kind='stable')
    cs = np.zeros(dim.prod()+1, np.int64); np.add.at(cs, c+1, 1); cs = np.cumsum(cs)
    return lo, dim, order, cs
cell = 0.03*1.001
G = [grid(f[0], cell) for f in frs[:6]]
for (a,b,rot) in [(0,1,2.0),(0,2,2.0),(0,3,2.0),(0,1,6.0)]:
    T = np.linalg.inv(frs[a][2]) @ frs[b][2] @ synth.perturbation(700, rot, rot/100)
    lo, dim, order, cs = G[a]
    src = frs[b][0][G[b][2]]          # source in its own cell-sorted order
    q = src @ T[:3,:3].T + T[:3,3]
    u = (q-lo)/cell; ic = np.floor(u).astype(int)
    inside = ((ic>=-1)&(ic<=dim)).all(1)
    rows_l, pts_l, cells_l, nin = [], [], [], []
    for s in range(0, len(q), 256):
        m = inside[s:s+256]; c = ic[s:s+256][m]
        if len(c)==0: rows_l.append(0); pts_l.append(0); cells_l.append(0); continue
        ymin,ymax,zmin,zmax = c[:,1].min(),c[:,1].max(),c[:,2].min(),c[:,2].max()
        nyb, nzb = ymax-ymin+3, zmax-zmin+3
        ext = {}
        for (ix,iy,iz) in set(map(tuple,c)):
            xa, xb = max(ix-1,0), min(ix+1,dim[0]-1)
            if xa>xb: continue
            for dy in (-1,0,1):
                for dz in (-1,0,1):
                    y,z = iy+dy, iz+dz
                    if 0<=y<dim[1] and 0<=z<dim[2]:
                        e=(y,z); lo_,hi_ = ext.get(e,(10**9,-1)); ext[e]=(min(lo_,xa),max(hi_,xb))
        npts=0; ncell=0
        for (y,z),(xa,xb) in ext.items():
            row=(z*dim[1]+y)*dim[0]
            npts += cs[row+xb+1]-cs[row+xa]; ncell += xb-xa+2
        rows_l.append(len(ext)); pts_l.append(npts); cells_l.append(ncell)
    rows_l, pts_l, cells_l = map(np.array,(rows_l,pts_l,cells_l))
    pc = lambda v: [int(np.percentile(v,p)) for p in (50,90,99,100)]
    print((a,b,rot), "blocks", len(rows_l), "distinct rows p50/90/99/max", pc(rows_l), "pts", pc(pts_l), "cells", pc(cells_l),
          "frac ok(rows<=192,pts<=1024,cells<=1024): %.3f" % np.mean((rows_l<=192)&(pts_l<=1024)&(cells_l<=1024)),
          "(256,1024,1024): %.3f" % np.mean((rows_l<=256)&(pts_l<=1024)&(cells_l<=1024)))
