"""`torch.optim.Adam` with its update as ONE HIP launch over every parameter (csrc/odpl_adam.hip).

The reference's `configure_optimizers` (LightningFunc/optimizer.py:6-38) returns `torch.optim.Adam(params, lr, weight_decay)`.  On the
GPU torch's fused implementation is a multi-tensor launch whose pointer lists must fit the kernel-argument block: YOLOv4's 362 tensors
take 10 launches (0.61 ms per step, 2.9 TB/s).  This subclass keeps torch's state layout (`state[p]` = step / exp_avg / exp_avg_sq,
so `state_dict()` round-trips with a plain torch.optim.Adam) and torch's arithmetic -- ATen's fused kernel restated expression by
expression -- and replaces the launches: the addresses that never change live in a device table built once, the gradient addresses
travel as kernel arguments, a block map turns the step into one flat stream.  Anything outside that shape (amsgrad, a GradScaler,
tensor learning rates, CPU or non-fp32 parameters) takes torch's own step; a gradient whose strides differ from its parameter's is
re-laid in the parameter's memory format first."""
import ctypes

import torch

from . import _lib


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, maximize=False, **kw):
        params = list(params)
        flat = [p for g in params for p in g["params"]] if params and isinstance(params[0], dict) else params
        kw.setdefault("fused", bool(flat) and all(torch.is_tensor(p) and p.is_cuda and p.is_floating_point() for p in flat))
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize, **kw)
        self._flat = {}                 # group index -> cached tables
        self.register_load_state_dict_post_hook(lambda opt: opt._flat.clear())

    one_launch = True                   # class switch (bench.py --torch-adam): False = torch's own step

    def _tables(self, gi, params):
        # the cached table holds ADDRESSES: it is valid for these parameter objects at these storages only (a `model.to(...)` or a
        # `p.data = ...` after the first step moves a parameter without changing its identity)
        key = tuple((id(p), p.data_ptr()) for p in params)
        c = self._flat.get(gi)
        if c is not None and c["key"] == key:
            return c
        dev = params[0].device
        steps = torch.zeros(len(params), dtype=torch.float32, device=dev)
        rows, numel, blk0 = [], [], [0]
        for i, p in enumerate(params):
            st = self.state[p]
            if len(st) == 0:            # torch.optim.Adam._init_group, fused branch
                st["step"] = steps[i]
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif not (torch.is_tensor(st["step"]) and st["step"].is_cuda and st["step"].dtype == torch.float32):
                return None             # a state this path did not create and cannot address
            else:
                steps[i] = st["step"]   # adopt an existing count (a loaded state_dict): the state keeps ONE flat step vector
                st["step"] = steps[i]
            if st["exp_avg"].stride() != p.stride() or st["exp_avg_sq"].stride() != p.stride() or st["exp_avg"].dtype != torch.float32:
                return None
            rows.append([p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), steps[i].data_ptr()])
            numel.append(p.numel())
            blk0.append(blk0[-1] + (p.numel() + _lib.ADAM_BLOCK - 1) // _lib.ADAM_BLOCK)
        blk_tensor = torch.repeat_interleave(torch.arange(len(params), dtype=torch.int32),
                                             torch.tensor([b - a for a, b in zip(blk0[:-1], blk0[1:])]))
        c = {"key": key, "steps": steps,
             "table": torch.tensor(rows, dtype=torch.int64).to(dev), "numel": torch.tensor(numel, dtype=torch.int32).to(dev),
             "blk0": torch.tensor(blk0, dtype=torch.int32).to(dev), "blk_tensor": blk_tensor.to(dev),
             "host_blk0": (ctypes.c_uint32 * len(blk0))(*blk0), "n_blocks": blk0[-1], "strides": [p.stride() for p in params]}
        self._flat[gi] = c
        return c

    @torch.no_grad()
    def step(self, closure=None):
        if not self.one_launch or closure is not None or not all(p.is_cuda for g in self.param_groups for p in g["params"]) \
                or torch.cuda.is_current_stream_capturing():      # (a captured step keeps torch's capturable kernel)
            return super().step(closure)
        plans = []
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            lr = group["lr"]
            ok = (not group["amsgrad"] and not group.get("capturable") and not group.get("differentiable") and not torch.is_tensor(lr)
                  and not group.get("decoupled_weight_decay") and getattr(self, "grad_scale", None) is None
                  and getattr(self, "found_inf", None) is None and len(params) < (1 << 20))
            dev = params[0].device
            grads = []
            for p in params:
                g = p.grad
                if not (ok and p.is_cuda and p.device == dev and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse
                        and p.numel() < (1 << 31)):
                    ok = False
                    break
                if g.stride() != p.stride():        # (e.g. an NCHW-contiguous gradient for a channels-last weight): element i of the
                    g = torch.empty_like(p).copy_(g)   # flat stream must be the same element of p, g, exp_avg and exp_avg_sq
                    p.grad = g
                grads.append(g)
            c = self._tables(gi, params) if ok else None
            if c is None:
                return super().step()   # one group outside the shape: the whole step is torch's (no parameter is stepped twice)
            plans.append((group, c, grads, dev, params))
        L = _lib.lib()
        for group, c, grads, dev, params in plans:
            c["steps"].add_(1)          # torch._foreach_add_(state_steps, 1): here the counts are one vector
            d = _lib.AdamDesc()
            d.table, d.numel, d.blk0, d.blk_tensor = c["table"].data_ptr(), c["numel"].data_ptr(), c["blk0"].data_ptr(), c["blk_tensor"].data_ptr()
            d.host_blk0 = ctypes.cast(c["host_blk0"], ctypes.c_void_p)
            d.n_tensors, d.n_blocks, d.maximize = len(grads), c["n_blocks"], 1 if group["maximize"] else 0
            b1, b2 = group["betas"]
            d.lr, d.beta1, d.beta2, d.eps, d.weight_decay = float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"])
            ptrs = (ctypes.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
            _lib.check(L.odpl_adam_step(ctypes.byref(d), ptrs, _lib.current_stream(dev)), "odpl_adam_step")
            # the kernel wrote the parameters through raw pointers: tell autograd, as torch's in-place update does (a backward through a
            # graph that saved the old weights must raise, not silently use the new ones)
            torch.autograd.graph.increment_version(params)
        return None
