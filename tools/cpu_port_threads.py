import time, torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle.torch_port import TorchLinear, TorchMatMul
g=torch.Generator().manual_seed(0)
hp=dict(metric="hessian",eq_alpha=0.01,eq_beta=1.2,eq_n=100,search_round=1)
def lin(K,N):
    w=torch.randn(N,K,generator=g)*0.02; b=torch.zeros(N); x=torch.randn(4,197,K,generator=g)
    out=torch.nn.functional.linear(x,w,b); grad=torch.randn(out.shape,generator=g)*1e-3
    return lambda ch: TorchLinear(w,b,w_bit=8,a_bit=8,n_V=1,chunk=ch,**hp).calibration_step2(x,out,grad), 2*2*100*788*K*N
def mm():
    A=torch.randn(4,12,197,64,generator=g)*0.125; B=torch.randn(4,12,64,197,generator=g); out=A@B; grad=torch.randn(out.shape,generator=g)*1e-3
    return lambda ch: TorchMatMul(A_bit=8,B_bit=8,chunk=ch,**hp).calibration_step2(A,B,out,grad), 2*200*4*12*197*64*197
print("logical cpus", os.cpu_count(), "default torch threads", torch.get_num_threads())
for name,(run,ops) in (("proj",lin(768,768)),("fc1",lin(768,3072)),("qk",mm())):
    for th in (8,16,32,64,128):
        torch.set_num_threads(th)
        for ch in ((4,10) if name!="qk" else (2,4)):
            t=time.time(); run(ch); dt=time.time()-t
            print(name, "threads",th,"chunk",ch, round(dt,3),"s", round(ops/dt/1e12,3),"TFLOP/s", flush=True)
