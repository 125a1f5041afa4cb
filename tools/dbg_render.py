import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from raytracer_amd import scenes
from raytracer_amd.host import DeviceScene, render_params
from tests import orc
which = sys.argv[1]
w,h=64,64
if which=='bunny':
    sc=scenes.small_bunny(w,h,env_pixels=scenes.synthetic_env())
else:
    sc=scenes.veach_mi(w,h)
tree=sc.initKdTree(); flat=tree.flatten(1); dev=DeviceScene(flat)
for depth in (0,1,5):
    p=render_params(w,h,1,1,max_depth=depth,seed=3)
    gpu=np.zeros((h,w,3),np.float32); rs=dev.render(p,gpu)
    cpu,crs,_,_=orc.oracle_render(flat.desc,p,0,threads=16)
    rel=np.abs(gpu-cpu)/(np.abs(cpu)+1e-3)
    bad=rel.max(axis=2)>1e-4
    print(which,'depth',depth,'bad pixels',bad.sum(),'rays gpu',rs.rays_closest,rs.rays_any,'cpu',crs.rays_closest,crs.rays_any)
    if bad.sum():
        ys,xs=np.nonzero(bad)
        for k in range(min(5,len(ys))):
            print('   px',xs[k],ys[k],'gpu',gpu[ys[k],xs[k]],'cpu',cpu[ys[k],xs[k]])
