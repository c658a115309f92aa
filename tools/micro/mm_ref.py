import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
for (M,K,N) in ((983040,480,480),(983040,240,480),(491520,480,480)):
    a=torch.randn(M,K,device='cuda'); b=torch.randn(K,N,device='cuda')
    for _ in range(3): c=a@b
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): c=a@b
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
    print(M,K,N,"%.3f ms %.1f TFLOP/s"%(dt*1e3, 2*M*K*N/dt/1e12))
