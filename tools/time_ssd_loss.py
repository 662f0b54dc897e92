#!/usr/bin/env python3
"""SSDLoss (csrc/odpl_ssd.hip) at B = 32, 8 732 priors, 21 classes, 8 boxes per image: time per call, dense and rows-only; with a
-DODPL_SSD_TRACE build (`--trace`) the image kernel's phases from wall-clock stamps."""
import torch, sys
sys.path.insert(0, ".")
import torch.nn as nn
from objectdetectionpl_amd.LightningFunc.losses_dense import SSDLoss
from objectdetectionpl_amd.LightningFunc.utils.SSDUtils import get_dboxes
d=torch.device("cuda:0"); g=torch.Generator().manual_seed(1)
B,C=32,21; pri=get_dboxes(); Pn=pri.shape[0]
off=(torch.randn(B,Pn,4,generator=g)*0.5).to(d); cls=torch.randn(B,Pn,C,generator=g).to(d)
n=B*8; wh=torch.rand(n,2,generator=g)*0.45+0.05
y=torch.cat([torch.arange(B).repeat_interleave(8).float()[:,None], torch.randint(1,C,(n,1),generator=g).float(), torch.rand(n,2,generator=g)*0.6+0.2, wh],1).to(d)
crit=SSDLoss(pri, nn.CrossEntropyLoss, "smooth_l1_loss", C, 300)
go,gc=torch.empty_like(off),torch.empty_like(cls)
act=torch.empty(B*Pn,dtype=torch.int32,device=d); cnt=torch.empty(B,dtype=torch.int32,device=d)
for mode in ("dense","rows_only"):
    kw={} if mode=="dense" else {"rows_only":(act,cnt)}
    for _ in range(3): crit._launch(off,cls,go,gc,y,**kw)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): crit._launch(off,cls,go,gc,y,**kw)
    e1.record(); torch.cuda.synchronize(); print("ssd loss", mode, "us per call", round(e0.elapsed_time(e1)/20*1e3,1))

if "--trace" in sys.argv:
    import ctypes, json
    import numpy as np
    from objectdetectionpl_amd import _lib
    L = _lib.lib(); L.odpl_ssd_trace.argtypes = [ctypes.c_void_p]
    buf = torch.zeros(B * 16, dtype=torch.int64, device=d)
    assert L.odpl_ssd_trace(buf.data_ptr()) == 0
    crit._launch(off, cls, go, gc, y); torch.cuda.synchronize(); L.odpl_ssd_trace(None)
    t = buf.cpu().numpy().reshape(B, 16).astype(np.float64) / 100.0
    names = ["lens + image rank", "best annotation per prior", "best prior per annotation", "classification loss per prior", "hard-negative threshold", "dense zero fill", "gradients + sums"]
    print(json.dumps({"span_us": round(float(t[:, 7].max() - t[:, 0].min()), 1), **{names[i] + "_us": round(float((t[:, i + 1] - t[:, i]).mean()), 1) for i in range(7)}}))
