import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from oracle import render as orr, field as ofield
from util import oracle_of_neus, oracle_flat_grads, rel_l2, leaf
import test_fullsize_configs as T
from neuralsim_amd import scenarios as sc
from neuralsim_amd.losses import mono_depth_loss, mono_normal_loss
torch.set_num_threads(8)
dev=torch.device('cpu')
world=sc.indoor_world()
m=sc.indoor_model(dev,'f32',seed=42,small=False,world=world)
p=oracle_of_neus(m)
a=m.accel.aabb
t0=time.time()
val,occ=orr.build_occ_grid(p,a[0],a[1],[64,64,64],n_pts=2**19,n_steps=2)
print('occ',float(occ.float().mean()),time.time()-t0)
intr,c2w,WH=sc.indoor_rig(V=40,H=800,W=800,f=0.56*800)
N=int(sys.argv[1]) if len(sys.argv)>1 else 512
r=T._rays(intr,c2w,WH,N,seed=32,C=64)
tr_gt=world.trace(r['o'],r['d'])
_,gt_n=sc.mono_priors(tr_gt["t"],tr_gt["normal"],r["o"]+tr_gt["t"][:,None]*r["d"])
def run(dtype):
    c=lambda t: t.detach().to(dtype).clone()
    q=ofield.FieldParams(spec=p.spec,grid=c(p.grid),sdf_w=[c(w) for w in p.sdf_w],sdf_b=[c(b) for b in p.sdf_b],
        rad_w=[c(w) for w in p.rad_w],rad_b=[c(b) for b in p.rad_b],ln_inv_s=c(p.ln_inv_s),ln_inv_s_factor=p.ln_inv_s_factor)
    for t_ in q.tensors(): t_.requires_grad_(True)
    kw=T._qkw(m,r); kw['jitter']=kw['jitter'].to(dtype); kw['jitter_c']=kw['jitter_c'].to(dtype)
    ret=orr.ray_query(q,r['o'].to(dtype),r['d'].to(dtype),r['ha'].to(dtype),occ,a[0].to(dtype),a[1].to(dtype),[64,64,64],near=0.01,far=None,depth_use_normalized_vw=False,**kw)
    rr=ret['rendered']
    occm=(rr['mask_volume'].detach()>0.5).to(dtype)
    mono_normal_loss(rr['normals_volume'],gt_n.to(dtype),occm).backward()
    return oracle_flat_grads(q), ret
g32,r32=run(torch.float32)
print('f32 done',time.time()-t0)
g64,r64=run(torch.float64)
print('samples',r32['volume_buffer']['t'].shape, r64['volume_buffer']['t'].shape)
for k in g32: print(k, rel_l2(g32[k].double(), g64[k]))
