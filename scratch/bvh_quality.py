import sys
sys.path.insert(0,'/root/repo')
import numpy as np, bench
from oracle import oracle_api as oa
from raytracing_project_amd import api, scenes
name = sys.argv[1]
wl = bench.build_workload(name, api, scenes)
W,H=wl["W"],wl["H"]
scene = api.Scene(wl["verts"], wl["idx"], wl["tri_mat"], [api.make_material(**m) for m in wl["mats"]]); scene.build(); info=scene.info()
nodes, order = scene.bvh(); tris = scene.tris()
_, osc = bench.oracle_scene(wl); lo,hi = osc.root_box()
ocam = oa.make_camera(wl["view"], W, H); o,d1 = oa.camera_rays(ocam,(0,0,W,H))
pick=np.arange(0,W*H,97)
t,f,nv,tt,gs,tf = oa.bvh_trace(nodes,tris,order,lo,hi,o[pick],d1[pick],threads=8,nwild=info.nwild,layout_counts=True)
hit=f>=0
dd=d1[pick][hit]; d2=dd/np.sqrt((dd*dd).sum(1,dtype=np.float32))[:,None]
P=(o[pick][hit]+t[hit,None]*d2).astype(np.float32); L=np.float32(wl["lights"][0]); sd=(L[None]-P).astype(np.float32)
sd1=(sd/np.sqrt((sd*sd).sum(1,dtype=np.float32))[:,None]).astype(np.float32)
t2,f2,nv2,tt2,gs2,tf2 = oa.bvh_trace(nodes,tris,order,lo,hi,P,sd1,threads=8,nwild=info.nwild,layout_counts=True)
def loads(gs,tt,tf): return 4*gs.mean()+tt.mean()+3*tf.mean()
print("%s groups=%d nwild=%d | primary: steps %.2f tris %.2f full %.2f loads %.1f | shadow: steps %.2f tris %.2f loads %.1f" % (
  name[:10], info.nnodes//4, info.nwild, gs.mean(), tt.mean(), tf.mean(), loads(gs,tt,tf), gs2.mean(), tt2.mean(), loads(gs2,tt2,tf2)))
