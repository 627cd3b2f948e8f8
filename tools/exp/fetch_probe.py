"""Does FETCH_SIZE see fine-grained random reads of a table that no cache holds?  (run under rocprofv3 --pmc FETCH_SIZE)"""
import torch
n_tab = 1 << 32            # 4 Gi float32 = 16 GiB
tab = torch.empty(n_tab, dtype=torch.float32, device='cuda'); tab.fill_(1.0)
idx = torch.randint(0, n_tab, (1 << 26,), device='cuda')
torch.cuda.synchronize()
for _ in range(2):
    out = tab[idx]            # 64 Mi random 4-byte gathers: >= 4 GiB of 64-byte sectors
    torch.cuda.synchronize()
seq = tab[: 1 << 28].sum()    # 1 GiB streamed
torch.cuda.synchronize()
print(float(out.sum()), float(seq))
