import sys, os
sys.path.insert(0, os.getcwd())
import torch
from objectdetectionpl_amd.optim import Adam
dev = torch.device("cuda:0")
for wd in (0.0, 1e-5):
    g = torch.Generator().manual_seed(1)
    w = torch.randn(100000, generator=g).to(dev)
    gr = [torch.randn(100000, generator=g).to(dev) * (0.5 + k) for k in range(3)]
    p1, p2 = torch.nn.Parameter(w.clone()), torch.nn.Parameter(w.clone())
    o1, o2 = Adam([p1], lr=1e-3, weight_decay=wd), torch.optim.Adam([p2], lr=1e-3, weight_decay=wd, fused=True)
    for k in range(3):
        p1.grad = gr[k].clone(); p2.grad = gr[k].clone()
        o1.step(); o2.step()
        torch.cuda.synchronize()
        for name, a, b in (("param", p1.detach(), p2.detach()), ("exp_avg", o1.state[p1]["exp_avg"], o2.state[p2]["exp_avg"]),
                           ("exp_avg_sq", o1.state[p1]["exp_avg_sq"], o2.state[p2]["exp_avg_sq"])):
            ne = (a != b)
            ulp = (a.view(torch.int32) - b.view(torch.int32)).abs()
            print(f"wd={wd} step={k+1} {name}: differing {int(ne.sum())}/{a.numel()}  max ulp {int(ulp.max())}  max abs {float((a-b).abs().max()):.3e}")
