import torch, time, sys
sys.path.insert(0,'.')
from ieee_amd.metrics.distance import _distmat
q=torch.randn(10000,768,device='cuda').abs(); g=torch.randn(100000,768,device='cuda').abs()
for dt in (torch.bfloat16, None):
    for _ in range(2): d=_distmat(q,g,0,compute_dtype=dt)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(5): d=_distmat(q,g,0,compute_dtype=dt)
    torch.cuda.synchronize(); ms=(time.time()-t)/5*1e3
    print(dt, round(ms,3),'ms', round(2*10000*100000*768/ms/1e9,1),'TFLOP/s')
