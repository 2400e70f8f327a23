import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from helpers import random_soup, product_from
from test_gpu_parity import _moller_trumbore_f64
for n_tris,seed,extent in ((300,11,4.0),(6000,12,10.0)):
    d=random_soup(n_tris,seed,extent=extent,size=1.0); r=product_from(d,16,16,2)
    rng=np.random.default_rng(seed); n=6000
    org=rng.uniform(-extent,extent,(n,3)).astype(np.float32); dr=rng.normal(size=(n,3)).astype(np.float32); dr/=np.linalg.norm(dr,axis=1,keepdims=True)
    ip,uvt=r.QueryClosest(org,dr,0.01,5000.0)
    wt=r.GetWorldTriangles().astype(np.float64).reshape(-1,3,3)
    bt,bi,st=_moller_trumbore_f64(wt,org.astype(np.float64),dr.astype(np.float64),0.01,5000.0)
    clear=(bi>=0)&((st-bt)>1e-6*bt)
    tri=wt[bi[clear]]; o=org[clear].astype(np.float64); dd=dr[clear].astype(np.float64)
    e1,e2=tri[:,1]-tri[:,0],tri[:,2]-tri[:,0]; nrm=np.cross(e1,e2); cos=np.abs(np.einsum('ij,ij->i',nrm,dd))/np.linalg.norm(nrm,axis=1)
    dt=np.abs(uvt[clear,2].astype(np.float64)-bt[clear]); t=bt[clear]
    scale=np.abs(o).max(axis=1)+np.abs(tri).max(axis=(1,2))
    print(n_tris,'rel max',(dt/t).max(),'abs max',dt.max(),'dt*cos/scale max',(dt*cos/scale).max(),'p99.9',np.quantile(dt*cos/scale,0.999), 'dt*cos/(scale+t)', (dt*cos/(scale+t)).max())
    r.close()
