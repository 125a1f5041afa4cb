import sys
sys.path.insert(0,'/root/repo')
import torch
from raytracer_amd import scenes
from raytracer_amd.host import DeviceScene, render_params
W,H=1920,1080
sc=scenes.bunny_pair(W,H)
dev=DeviceScene(sc.initKdTree().flatten(1))
img=torch.zeros((H,W,3),device='cuda')
prev=None
for d in range(0,6):
    p=render_params(W,H,16,16,max_depth=d,seed=1)
    dev.render(p,img)
    rs=dev.render(p,img)
    cur=(rs.rays_closest,rs.rays_any,rs.rays_mis,rs.extend_ms,rs.connect_ms,rs.shade_ms)
    if prev is None: dl=cur
    else: dl=tuple(a-b for a,b in zip(cur,prev))
    print("depth %d: +closest %.1fM (+mis %.1fM) +any %.1fM | +extend %.1f ms +connect %.1f ms +shade %.1f ms"%(d,dl[0]/1e6,dl[2]/1e6,dl[1]/1e6,dl[3],dl[4],dl[5]),flush=True)
    prev=cur
