"""`objectdetectionpl_amd.optim.Adam`: torch.optim.Adam's state and arithmetic.  CPU here (the torch step is taken off the device);
the one-launch HIP step against torch's fused kernel is tests/test_gpu_optim.py."""
import torch


def test_cpu_parameters_take_torchs_own_step_and_the_state_dict_round_trips():
    from objectdetectionpl_amd.optim import Adam
    torch.manual_seed(0)
    w0 = [torch.randn(7, 5), torch.randn(11)]
    ps = [[torch.nn.Parameter(w.clone()) for w in w0] for _ in range(2)]
    a, b = Adam(ps[0], lr=1e-2, weight_decay=1e-3), torch.optim.Adam(ps[1], lr=1e-2, weight_decay=1e-3)
    for k in range(3):
        for opt, pp in ((a, ps[0]), (b, ps[1])):
            opt.zero_grad(set_to_none=True)
            sum(((p * (i + 1 + k)) ** 2).sum() for i, p in enumerate(pp)).backward()
            opt.step()
    for p, q in zip(*ps):
        assert torch.equal(p, q)
    c = torch.optim.Adam([torch.nn.Parameter(w.clone()) for w in w0], lr=1e-2, weight_decay=1e-3)
    c.load_state_dict(a.state_dict())                                    # a plain torch.optim.Adam reads this subclass's state
    assert c.state_dict()["state"][0]["step"] == b.state_dict()["state"][0]["step"]


def test_configure_optimizers_returns_the_subclass_with_the_references_hyperparameters():
    import argparse

    from objectdetectionpl_amd.LightningFunc.optimizer import configure_optimizers
    from objectdetectionpl_amd.optim import Adam

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l = torch.nn.Linear(3, 2)
            self.args = argparse.Namespace(optimizer="Adam", lr=1e-3, weight_decay=1e-5, lr_scheduler="ReduceLROnPlateau")
    d = configure_optimizers(M())
    assert isinstance(d["optimizer"], Adam) and isinstance(d["optimizer"], torch.optim.Adam)
    g = d["optimizer"].param_groups[0]
    assert g["lr"] == 1e-3 and g["weight_decay"] == 1e-5 and g["betas"] == (0.9, 0.999) and g["eps"] == 1e-8     # optimizer.py:7-10
    assert d["monitor"] == "val_loss"
