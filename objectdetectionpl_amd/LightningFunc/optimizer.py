"""`configure_optimizers` of the reference (`LightningFunc/optimizer.py:6-38`): same `args` fields, same choices and
hyper-parameters (including `ReduceLROnPlateau(mode='max')` on `val_loss`), returned in the same
`{'optimizer', 'lr_scheduler', 'monitor'}` dict.  Adam is `optim.Adam`: torch.optim.Adam (same state, same arithmetic) whose update is
one HIP launch over every parameter when they live on the GPU (csrc/odpl_adam.hip), torch's own step otherwise."""
from torch.optim import SGD, Adagrad, RMSprop, lr_scheduler as _sched

from ..optim import Adam

_OPTIMIZERS = {
    "Adam": lambda p, a: Adam(p, lr=a.lr, weight_decay=a.weight_decay),
    "Adagrad": lambda p, a: Adagrad(p, lr=a.lr, lr_decay=a.lr_decay, weight_decay=a.weight_decay),
    "RMSprop": lambda p, a: RMSprop(p, lr=a.lr, alpha=a.alpha, momentum=a.momentum, weight_decay=a.weight_decay),
    "SGD": lambda p, a: SGD(p, lr=a.lr, momentum=a.momentum, weight_decay=a.weight_decay),
}
_SCHEDULERS = {
    "ReduceLROnPlateau": lambda o: _sched.ReduceLROnPlateau(o, mode="max", patience=3, threshold=0.9),
    "StepLR": lambda o: _sched.StepLR(o, step_size=3, gamma=0.8),
    "MultiStepLR": lambda o: _sched.MultiStepLR(o, milestones=[70, 140, 190], gamma=0.1),
    "ExponentialLR": lambda o: _sched.ExponentialLR(o, gamma=0.99),
    "CosineAnnealingLR": lambda o: _sched.CosineAnnealingLR(o, T_max=20),
    "LambdaLR": lambda o: _sched.LambdaLR(o, lr_lambda=lambda x: 1),
    "CyclicLR": lambda o: _sched.CyclicLR(o, base_lr=1e-5, max_lr=0.1),
}


def configure_optimizers(self, mode="sgd"):
    a = self.args
    if a.optimizer not in _OPTIMIZERS:
        raise UnboundLocalError(f"unknown optimizer {a.optimizer!r}")           # the reference falls through to an unbound local
    if a.lr_scheduler not in _SCHEDULERS:
        raise UnboundLocalError(f"unknown lr_scheduler {a.lr_scheduler!r}")
    optimizer = _OPTIMIZERS[a.optimizer](list(self.parameters()), a)
    return {"optimizer": optimizer, "lr_scheduler": _SCHEDULERS[a.lr_scheduler](optimizer), "monitor": "val_loss"}


def get_lr(optimizer):
    for group in optimizer.param_groups:
        return group["lr"]
