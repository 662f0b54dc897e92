"""Rank program of tests/test_gpu_lossdriven_ddp.py: two ranks share cuda:0 over gloo (RCCL refuses duplicate devices), each runs ONE
loss-driven RetinaNet training step on its own batch behind DataParallelHost; rank 0 also computes both batches alone and compares."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import torch.distributed as dist

from objectdetectionpl_amd import parallel
from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
from objectdetectionpl_amd.model.RetinaNet import RetinaNet
from objectdetectionpl_amd.trainer import _ForwardVia


def targets(g, B, per, C):
    n = B * per
    wh = torch.rand(n, 2, generator=g) * 0.35 + 0.15
    cxy = torch.rand(n, 2, generator=g) * (1 - wh) + wh / 2
    return torch.cat([torch.arange(B).repeat_interleave(per).float()[:, None], torch.randint(0, C, (n, 1), generator=g).float(), cxy, wh], 1)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda:0")
    C, size, B = 4, 256, 2
    args = argparse.Namespace(model_name="RetinaNet", cls_criterion="focal_loss", coord_criterion="smooth_l1_loss", data_module="synthetic")
    RetinaNet.img_size = size
    sc.min_pixels = 500
    torch.manual_seed(100 + rank)                        # different initial weights per rank: the wrapper must broadcast rank 0's
    model = RetinaNet([str(i) for i in range(C)], args).to(dev).to(memory_format=torch.channels_last).train()
    net = parallel.wrap_data_parallel(model, dev, impl="bucketed")
    via = _ForwardVia(model, net)
    batches = []
    for r in range(world):
        g = torch.Generator().manual_seed(7 + r)
        x = torch.rand(B, 3, size, size, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        batches.append((x, targets(g, B, 3, C).to(dev)))
    x, y = batches[rank]
    assert model.loss_driven_mask(x, y) != 0 and via.forward_and_loss is not None
    taken = []
    orig = model.forward_and_loss
    model.forward_and_loss = lambda a, b, mask=None: (taken.append(mask), orig(a, b, mask=mask))[1]
    d = type(model).training_step(via, (x, y), 0)
    d["loss"].backward()
    torch.cuda.synchronize()
    dist.barrier()
    assert len(taken) == 1 and taken[0]                  # one fold, handed the route the wrapper decided (not re-decided inside)
    got = torch.cat([p.grad.flatten().double() for p in model.parameters()])
    ok = True
    if rank == 0:
        ref = None
        for xb, yb in batches:                           # the same two steps alone, dense route, averaged
            RetinaNet.loss_driven_step = False
            model.zero_grad(set_to_none=True)
            with net.no_sync():
                dd = model.training_step((xb, yb), 0)
                dd["loss"].backward()
            gr = torch.cat([p.grad.flatten().double() for p in model.parameters()])
            ref = gr if ref is None else ref + gr
        ref /= world
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm()))
        ratio = float(got.norm() / ref.norm())
        print(f"cos {cos:.6f} norm ratio {ratio:.4f}", flush=True)
        ok = cos > 0.999 and abs(ratio - 1) < 0.02
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.broadcast(flag, 0)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1.0 else 3)


if __name__ == "__main__":
    main()
