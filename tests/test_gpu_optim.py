"""The one-launch Adam step (csrc/odpl_adam.hip through optim.Adam) against torch's fused Adam on the device."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(make_opt, shapes, steps, seed, channels_last=True):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    ps = []
    for s in shapes:
        w = torch.randn(*s, generator=g).to(dev)
        if channels_last and len(s) == 4:
            w = w.contiguous(memory_format=torch.channels_last)
        ps.append(torch.nn.Parameter(w))
    opt = make_opt(ps)
    for k in range(steps):
        for p in ps:                                      # autograd hands the optimizer NEW gradient tensors every step
            gr = torch.randn(p.shape, generator=g).to(dev) * (0.5 + k)
            p.grad = gr.contiguous(memory_format=torch.channels_last) if (channels_last and p.dim() == 4) else gr
        opt.step()
    torch.cuda.synchronize()
    return ps, opt


SHAPES = [(64, 32, 3, 3), (255, 1024, 1, 1), (1024,), (3,), (5, 7), (33, 16, 1, 1), (4097,), (1,), (128, 64, 3, 3)]


@pytest.mark.parametrize("wd", [1e-5, 0.0])
def test_one_launch_adam_equals_torchs_fused_adam_bit_for_bit(wd):
    """three steps, parameters of every alignment class (4096-multiple, ragged tails, fewer than four elements,
    channels-last 4-d weights), weight decay on and off: parameters, exp_avg, exp_avg_sq and step equal torch.optim.Adam(fused=True)."""
    from objectdetectionpl_amd.optim import Adam
    mine, om = _run(lambda ps: Adam(ps, lr=1e-3, weight_decay=wd), SHAPES, 3, 4)
    ref, ot = _run(lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=wd, fused=True), SHAPES, 3, 4)
    assert om._flat, "the HIP step did not run"
    for p, q in zip(mine, ref):
        assert torch.equal(p, q), float((p - q).abs().max())
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(om.state[p][k], ot.state[q][k]), k
        assert float(om.state[p]["step"]) == float(ot.state[q]["step"]) == 3.0


def test_state_dict_round_trip_and_fallbacks():
    from objectdetectionpl_amd.optim import Adam
    mine, om = _run(lambda ps: Adam(ps, lr=2e-3, weight_decay=1e-4), SHAPES[:4], 2, 5)
    sd = om.state_dict()
    ref, ot = _run(lambda ps: torch.optim.Adam(ps, lr=2e-3, weight_decay=1e-4, fused=True), SHAPES[:4], 2, 5)
    for a, b in zip(sd["state"].values(), ot.state_dict()["state"].values()):
        assert float(a["step"]) == float(b["step"]) and torch.equal(a["exp_avg"], b["exp_avg"])
    # load into a fresh optimizer of each kind and take one more (identical) step
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(6)
    grads = [torch.randn(p.shape, generator=g).to(dev) for p in mine]
    p1 = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    p2 = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    o1, o2 = Adam(p1, lr=2e-3, weight_decay=1e-4), torch.optim.Adam(p2, lr=2e-3, weight_decay=1e-4, fused=True)
    import copy
    o1.load_state_dict(copy.deepcopy(sd)); o2.load_state_dict(copy.deepcopy(sd))      # (load_state_dict keeps the tensors it is given)
    for p, q, gr in zip(p1, p2, grads):
        p.grad = gr.clone().contiguous(memory_format=torch.channels_last) if p.dim() == 4 else gr.clone()
        q.grad = p.grad.clone()
    o1.step(); o2.step()
    for p, q in zip(p1, p2):
        assert torch.equal(p, q)
    assert float(o1.state[p1[0]]["step"]) == 3.0
    # a gradient whose strides differ from its parameter's is re-laid in the parameter's format, then stepped by the HIP launch
    w = torch.randn(8, 4, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    gn = torch.randn(8, 4, 3, 3, device=dev)                                                # NCHW-contiguous gradient
    pa, pb = torch.nn.Parameter(w.clone()), torch.nn.Parameter(w.clone())
    o3, o4 = Adam([pa], lr=1e-3), torch.optim.Adam([pb], lr=1e-3, fused=True)
    pa.grad = gn.clone()
    pb.grad = gn.clone().contiguous(memory_format=torch.channels_last)
    o3.step(); o4.step()
    assert o3._flat and torch.equal(pa, pb)
    # a parameter whose storage is replaced after the first step (model.to(memory_format=...), p.data = ...): the address table is rebuilt
    q1, q2 = torch.nn.Parameter(w.clone()), torch.nn.Parameter(w.clone())
    o5, o6 = Adam([q1], lr=1e-3), torch.optim.Adam([q2], lr=1e-3, fused=True)
    for k in range(2):
        g5 = torch.randn(8, 4, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
        q1.grad, q2.grad = g5.clone(), g5.clone()
        o5.step(); o6.step()
        if k == 0:
            q1.data = q1.data.clone()              # same values and strides, another allocation
    assert torch.equal(q1, q2)


def test_more_tensors_than_one_launch_carries():
    """ODPL_ADAM_MAX_TENSORS = 384 gradient addresses fit the kernel-argument block: 900 parameters take three launches"""
    from objectdetectionpl_amd.optim import Adam
    shapes = [(17 + i % 5, 3) for i in range(900)]
    mine, om = _run(lambda ps: Adam(ps, lr=1e-3, weight_decay=1e-5), shapes, 2, 8)
    ref, ot = _run(lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=1e-5, fused=True), shapes, 2, 8)
    assert om._flat
    for p, q in zip(mine, ref):
        assert torch.equal(p, q)


def test_adam_one_launch_step_bumps_the_parameters_version_counters():
    """The kernel writes the parameters through raw pointers; autograd must still see an in-place update: a backward through a graph that
    saved the OLD weights raises (as after torch.optim.Adam.step), it does not silently use the new ones."""
    import pytest as _pt
    from objectdetectionpl_amd.optim import Adam
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 8).to(d)
    opt = Adam(lin.parameters(), lr=1e-2)
    x = torch.randn(4, 8, device=d, requires_grad=True)
    lin(x).sum().backward()
    v0 = lin.weight._version
    y = lin(x)                                   # saves lin.weight for the backward wrt x
    opt.step()
    assert lin.weight._version > v0
    with _pt.raises(RuntimeError, match="modified by an inplace operation"):
        y.sum().backward()
