import sys; sys.path.insert(0,'/root/repo')
import torch
from perf_amd import ops
from perf_amd.grid import MlpConfig
from tools.microbench import timeit
dev='cuda'; n=1<<20
geo=MlpConfig(16,1,1,'Exponential')
w=(torch.randn(geo.n_params,device=dev)*0.2).to(torch.bfloat16)
feat=torch.rand(16,n,2,device=dev).to(torch.bfloat16)
dg=torch.randn(n,1,device=dev)
print('no absmax', timeit(lambda: ops.mlp_bwd(geo,w,feat,dg))*1e3)
print('absmax   ', timeit(lambda: ops.mlp_bwd(geo,w,feat,dg,want_absmax=True))*1e3)
print('no dfeat ', timeit(lambda: ops.mlp_bwd(geo,w,feat,dg,need_dfeat=False))*1e3)
