import sys, torch
sys.path.insert(0, '/root/repo')
from vrdone_amd import ops
torch.set_grad_enabled(False)
B,H,Tq,Tk,hd=2048,4,9,36,64
C=H*hd
q=torch.randn(B,Tq,C,device='cuda'); k=torch.randn(B,Tk,C,device='cuda'); v=torch.randn(B,Tk,C,device='cuda')
mask=torch.ones(B,Tk,dtype=torch.bool,device='cuda'); mask[:,32:]=False
outs={}
for algo in (1,2):
    for _ in range(3): o=ops.attention(q,k,v,mask,H,algo=algo)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): o=ops.attention(q,k,v,mask,H,algo=algo)
    e1.record(); torch.cuda.synchronize()
    outs[algo]=o
    print('algo',algo, e0.elapsed_time(e1)/10,'ms')
print('max diff', (outs[1]-outs[2]).abs().max().item())
