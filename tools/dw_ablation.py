import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pime_amd import ops
from pime_amd.elegantrl.net import CriticAdv
from pime_amd.elegantrl.net_residual import ActorResidualIntegratorModularPPO
DEV="cuda:0"
torch.manual_seed(0)
cri=CriticAdv(3,128).to(DEV); act=ActorResidualIntegratorModularPPO(128,3,1,1).to(DEV)
L=819200; B=65536
state=torch.randn(L,3,device=DEV)*3+5; action=torch.randn(L,device=DEV); lp=torch.randn(L,device=DEV)*0.1-1; adv=torch.randn(L,device=DEV); rs=torch.randn(L,device=DEV)*10
f=ops.FusedPPOGrad(act,cri,B)
scale=torch.ones(1,device=DEV)
idx=torch.randint(L,(B,),device=DEV)
for _ in range(3): f(state,action,lp,adv,rs,idx,0.2,0.02,scale)
torch.cuda.synchronize()
s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): f(state,action,lp,adv,rs,idx,0.2,0.02,scale)
e.record(); torch.cuda.synchronize()
print(os.environ.get("PIME_DW_DEBUG","0"), "us per minibatch_grad call:", s.elapsed_time(e)/20*1e3)
