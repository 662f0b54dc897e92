"""Dump torch's fused Adam states over three steps (tools/adam_fit.py fits ATen's arithmetic to them, expression by expression)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
n = 60000
w = torch.randn(n, generator=g).to(dev)
gr = [torch.randn(n, generator=g).to(dev) * (0.5 + k) for k in range(3)]
p = torch.nn.Parameter(w.clone())
o = torch.optim.Adam([p], lr=1e-3, weight_decay=1e-2, fused=True)
out = {"p0": w.cpu()}
for k in range(3):
    p.grad = gr[k].clone(); o.step(); torch.cuda.synchronize()
    out[f"g{k}"] = gr[k].cpu(); out[f"p{k+1}"] = p.detach().cpu().clone()
    out[f"m{k+1}"] = o.state[p]["exp_avg"].cpu().clone(); out[f"v{k+1}"] = o.state[p]["exp_avg_sq"].cpu().clone()
torch.save(out, "gpurun_out/adam_dump.pt")
print("ok")
