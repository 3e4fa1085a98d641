import torch
torch.manual_seed(0)
for (M,N,K) in ((30016,4096,1024),(120064,4096,1024),(120064,4096,5120)):
    A=(torch.relu(torch.rand(M,K,device='cuda')*2-1)*0.3).half(); W=((torch.rand(N,K,device='cuda')*2-1)*0.03).half()
    for _ in range(3): C=A@W.t()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): C=A@W.t()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(M,N,K,f"{ms:.3f} ms {2.0*M*N*K/ms/1e9:.0f} TF/s",flush=True)
    del A,W,C
