"""Offline float64 emulation of split-precision convolutions (decides which fp16/bf16 split keeps the 1e-4 logit budget).
Needs only tests/golden; prints the logit error each split adds relative to exact arithmetic."""
import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from bokego_amd.bkw import load_bkw
torch.set_grad_enabled(False)
G='/root/repo/tests/golden'
f=torch.from_numpy(np.load(f'{G}/features.npz')['incremental'].astype(np.float64))
gold=np.load(f'{G}/nets.npz')
def folded(sd):
    Ws,Bs=[],[]
    for c,b in zip((0,3,6,9,12,15,18),(1,4,7,10,13,16,19)):
        w=sd[f'conv.{c}.weight'].astype(np.float64); s=sd[f'conv.{b}.weight'].astype(np.float64)/np.sqrt(sd[f'conv.{b}.running_var'].astype(np.float64)+1e-5)
        Ws.append((w*s[:,None,None,None]).astype(np.float32))
        Bs.append((((sd[f'conv.{c}.bias'].astype(np.float64)-sd[f'conv.{b}.running_mean'])*s+sd[f'conv.{b}.bias'])).astype(np.float32))
    return Ws,Bs
def split(x, scale, n, dt=torch.float16):
    """x float64 tensor (holding fp32 values) -> list of n pieces (float64 holding half values), unscaled"""
    r=x*scale; out=[]
    for _ in range(n):
        p=r.to(dt).to(torch.float64); out.append(p/scale); r=r-p
    return out
def run(sd, mode, sa=16.0, sw=4096.0, dt=torch.float16):
    Ws,Bs=folded(sd)
    x=f.clone()
    for l in range(7):
        w=torch.from_numpy(Ws[l].astype(np.float64)); b=torch.from_numpy(Bs[l].astype(np.float64))
        pad=2 if l==0 else 1
        if mode=='exact':
            y=torch.nn.functional.conv2d(x,w,b,padding=pad)
        else:
            na,nw,terms=mode
            ap=split(x,sa,na,dt); wp=split(w,sw,nw,dt)
            y=b.view(1,-1,1,1).expand(x.shape[0],-1,9,9).clone()
            for (i,j) in terms:
                y=y+torch.nn.functional.conv2d(ap[i],wp[j],None,padding=pad)
        x=torch.relu(y).to(torch.float32).to(torch.float64)   # activations are stored as fp32 (or as the split of fp32)
    hw=torch.from_numpy(sd['conv.21.weight'].astype(np.float64)); hb=torch.from_numpy(sd['conv.21.bias'].astype(np.float64))
    return (torch.nn.functional.conv2d(x,hw)+hb).reshape(-1,81).numpy()
sd=load_bkw(f'{G}/policy_19.bkw')
ex=run(sd,'exact')
print('exact(fp64 accumulate) vs reference goldens: %.3g'%np.abs(ex-gold['logits_b1']).max())
for name,mode,dt in [
  ('f16 2x2 3 products',(2,2,[(0,0),(0,1),(1,0)]),torch.float16),
  ('f16 2x2 4 products',(2,2,[(0,0),(0,1),(1,0),(1,1)]),torch.float16),
  ('f16 a2 x w3 5 products',(2,3,[(0,0),(0,1),(1,0),(1,1),(0,2)]),torch.float16),
  ('bf16 3x3 6 products',(3,3,[(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)]),torch.bfloat16),
  ('bf16 2x3 5 products',(2,3,[(0,0),(0,1),(1,0),(1,1),(0,2)]),torch.bfloat16),
]:
    sa,sw=(16.0,4096.0) if dt==torch.float16 else (1.0,1.0)
    y=run(sd,mode,sa,sw,dt)
    print('%-26s split-induced max|dlogit| %.3g   vs goldens %.3g'%(name,np.abs(y-ex).max(),np.abs(y-gold['logits_b1']).max()))

# ---- can some layers live with 2 products? --------------------------------------------------------
print()
for name, mode in [
    ('f16 w22 x a11 (hh+lh), all layers', (1, 2, [(0, 0), (0, 1)])),
    ('f16 w11 x a22 (hh+hl), all layers', (2, 1, [(0, 0), (1, 0)])),
]:
    y = run(sd, mode, 16.0, 4096.0, torch.float16)
    print('%-36s split-induced max|dlogit| %.3g   vs goldens %.3g' % (name, np.abs(y - ex).max(), np.abs(y - gold['logits_b1']).max()))
