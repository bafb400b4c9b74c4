"""Adam whose step is ONE library launch over all parameter tensors (`ms3d_adam_step`, csrc/optim.hip) -- the optimizer
the reference builds through Hydra (`torch.optim.Adam`, config/model/base.yaml:23-28) with the same state
(`step`, `exp_avg`, `exp_avg_sq` per parameter: checkpoints are interchangeable with torch.optim.Adam's) and the same
arithmetic as torch's fused implementation.  Anything the kernel does not cover (amsgrad, maximize, capturable, sparse or
non-f32 parameters, CPU parameters) takes torch's own step."""
import ctypes as C

import torch

from . import _lib
from .backend import get_backend

_RING = 4


class Adam(torch.optim.Adam):
    def __init__(self, params, **kw):
        kw.pop("fused", None)            # the library step below is the fused path
        super().__init__(params, **kw)
        self._plans = {}

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _eligible(group, params):
        if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
            return False
        if torch.is_tensor(group["lr"]):
            return False
        for p in params:
            g = p.grad
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g is not None and not g.is_sparse
                    and g.dtype == torch.float32 and g.is_contiguous() and g.device == p.device):
                return False
        return len(params) > 0

    def _plan(self, gi, params):
        key = tuple(id(p) for p in params)
        plan = self._plans.get(gi)
        if plan is not None and plan["key"] == key:
            return plan
        lib = get_backend().lib
        chunk = lib.ms3d_adam_chunk_elems()
        dev = params[0].device
        rows = []
        for t, p in enumerate(params):
            rows += [(t, c) for c in range(-(-p.numel() // chunk))]
        plan = dict(key=key, lib=lib, n=len(params), n_chunks=len(rows),
                    chunks=torch.tensor(rows, dtype=torch.int32).reshape(-1, 2).to(dev),
                    sizes=torch.tensor([p.numel() for p in params], dtype=torch.int64).to(dev),
                    host=[torch.empty((4, len(params)), dtype=torch.int64).pin_memory() for _ in range(_RING)],
                    dev=torch.zeros((4, len(params)), dtype=torch.int64, device=dev), last=None, turn=0)
        self._plans[gi] = plan
        return plan

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        groups = [(gi, g, [p for p in g["params"] if p.grad is not None]) for gi, g in enumerate(self.param_groups)]
        if not all(self._eligible(g, ps) for _, g, ps in groups if ps):
            return super().step() if closure is None else (super().step(), loss)[1]
        for gi, group, params in groups:
            if not params:
                continue
            beta1, beta2 = group["betas"]
            # state, exactly torch.optim.Adam's; the step counter is ONE host tensor shared by the group's parameters
            shared = None
            for p in params:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if shared is None:
                    shared = st["step"] if st["step"].device.type == "cpu" else st["step"].cpu()
                st["step"] = shared
            shared += 1
            t = float(shared)
            plan = self._plan(gi, params)
            host = plan["host"][plan["turn"]]
            plan["turn"] = (plan["turn"] + 1) % _RING
            ptrs = [[p.data_ptr() for p in params], [p.grad.data_ptr() for p in params],
                    [self.state[p]["exp_avg"].data_ptr() for p in params],
                    [self.state[p]["exp_avg_sq"].data_ptr() for p in params]]
            host.copy_(torch.tensor(ptrs, dtype=torch.int64))
            plan["dev"].copy_(host, non_blocking=True)
            d, n = plan["dev"].data_ptr(), plan["n"] * 8
            _lib.check(plan["lib"].ms3d_adam_step(
                _lib.ptr(plan["chunks"]), plan["n_chunks"], C.c_void_p(d), C.c_void_p(d + n), C.c_void_p(d + 2 * n),
                C.c_void_p(d + 3 * n), _lib.ptr(plan["sizes"]), C.c_float(float(group["lr"])), C.c_float(beta1),
                C.c_float(beta2), C.c_float(group["eps"]), C.c_float(group["weight_decay"]),
                C.c_double(1.0 - beta1 ** t), C.c_double(1.0 - beta2 ** t), _lib.stream_handle()), "ms3d_adam_step")
        return loss
