"""Adam whose step is ONE library launch over all parameter tensors (`ms3d_adam_step`, csrc/optim.hip) -- the optimizer
the reference builds through Hydra (`torch.optim.Adam`, config/model/base.yaml:23-28) with the same state
(`step`, `exp_avg`, `exp_avg_sq` per parameter -- every parameter has its OWN step counter, started lazily with its first
gradient, as in torch.optim.Adam: the ScoreNet / refinement branches get their first gradients after `prepare_epochs`
and must start with bias corrections of step 1, not the backbone's; checkpoints are interchangeable with
torch.optim.Adam's) and the same arithmetic as torch's fused implementation.  Anything the kernel does not cover (amsgrad, maximize, capturable, sparse or
non-f32 parameters, CPU parameters) takes torch's own step."""
import ctypes as C

import numpy as np
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
        if plan is not None and plan["key"] == key and self.state[params[0]].get("step") is plan["first_view"]:
            return plan
        # (re)built when the set of parameters with gradients changes -- a branch switched on after `prepare_epochs` --
        # or after load_state_dict; every parameter's OWN counter is read from the state
        lib = get_backend().lib
        chunk = lib.ms3d_adam_chunk_elems()
        dev = params[0].device
        rows = []
        for t, p in enumerate(params):
            rows += [(t, c) for c in range(-(-p.numel() // chunk))]
        # The per-parameter step counters are 0-dim views into ONE host tensor: `steps += 1` is a single operation per
        # step (250 scalar tensor increments would cost the interpreter ~1 ms) while state[p]['step'] stays what
        # torch.optim.Adam keeps there -- a float32 scalar tensor per parameter, each with its own value.
        steps = torch.zeros(len(params), dtype=torch.float32)
        for t, p in enumerate(params):
            st = self.state[p]
            if len(st) == 0:                     # torch.optim.Adam's lazy state: a parameter's counter starts with its
                st["step"] = torch.tensor(0.0, dtype=torch.float32)          # first gradient (torch/optim/adam.py)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            steps[t] = float(st["step"])
            st["step"] = steps[t]
        host = [torch.empty((5, len(params)), dtype=torch.int64).pin_memory() for _ in range(_RING)]
        plan = dict(key=key, lib=lib, n=len(params), n_chunks=len(rows), steps=steps, steps_np=steps.numpy(),
                    first_view=self.state[params[0]]["step"],
                    chunks=torch.tensor(rows, dtype=torch.int32).reshape(-1, 2).to(dev),
                    sizes=torch.tensor([p.numel() for p in params], dtype=torch.int64).to(dev),
                    host=host, host_np=[h.numpy() for h in host],
                    dev=torch.zeros((5, len(params)), dtype=torch.int64, device=dev), turn=0)
        self._plans[gi] = plan
        return plan

    def state_dict(self):
        """torch.optim.Adam's layout; the step counters are handed out as independent scalar tensors (torch's
        load_state_dict keeps the very tensor objects it is given)"""
        sd = super().state_dict()
        sd["state"] = {k: {n: (v.clone() if n == "step" and torch.is_tensor(v) else v) for n, v in st.items()}
                       for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        self._plans = {}
        return super().load_state_dict(state_dict)

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        groups = [(gi, g, [p for p in g["params"] if p.grad is not None]) for gi, g in enumerate(self.param_groups)]
        if not all(self._eligible(g, ps) for _, g, ps in groups if ps):
            super().step()
            return loss
        for gi, group, params in groups:
            if not params:
                continue
            beta1, beta2 = group["betas"]
            plan = self._plan(gi, params)
            plan["steps"] += 1.0
            t = plan["steps_np"].astype(np.float64)
            turn = plan["turn"]
            plan["turn"] = (turn + 1) % _RING
            hn = plan["host_np"][turn]
            hn[0] = [p.data_ptr() for p in params]
            hn[1] = [p.grad.data_ptr() for p in params]
            hn[2] = [self.state[p]["exp_avg"].data_ptr() for p in params]
            hn[3] = [self.state[p]["exp_avg_sq"].data_ptr() for p in params]
            # per-tensor (lr / bias_correction1, sqrt(bias_correction2)) as two floats in the fifth row's 8-byte slots
            coef = hn[4].view(np.float32).reshape(-1, 2)
            coef[:, 0] = float(group["lr"]) / (1.0 - beta1 ** t)
            coef[:, 1] = np.sqrt(1.0 - beta2 ** t)
            plan["dev"].copy_(plan["host"][turn], non_blocking=True)
            d, n = plan["dev"].data_ptr(), plan["n"] * 8
            _lib.check(plan["lib"].ms3d_adam_step_multi(
                _lib.ptr(plan["chunks"]), plan["n_chunks"], C.c_void_p(d), C.c_void_p(d + n), C.c_void_p(d + 2 * n),
                C.c_void_p(d + 3 * n), _lib.ptr(plan["sizes"]), C.c_void_p(d + 4 * n), C.c_float(beta1),
                C.c_float(beta2), C.c_float(group["eps"]), C.c_float(group["weight_decay"]), _lib.stream_handle()),
                "ms3d_adam_step_multi")
        return loss
