"""``FlatAdam`` — ``torch.optim.Adam`` (defaults; reference ``vae_scripts/train_vae.py:301,304``) as ONE HIP
kernel over a model's flat fp32 parameter arena (``AutoencoderKL``: instead of ~220 per-tensor updates;
``PatchDiscriminator``: the same kernel over its arena).

``state_dict()`` / ``load_state_dict()`` use ``torch.optim.Adam``'s format over ``VAEModel.parameters()``
order, so ``checkpoint_epoch*.pth`` files (train_vae.py:752-765) are interchangeable with the
reference's ``optimizer_g_state_dict``.
"""
from __future__ import annotations

import torch

from . import ops


class FlatAdam:
    def __init__(self, net, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.net = net
        self.lr, self.betas, self.eps = float(lr), tuple(betas), float(eps)
        self.step_count = 0
        arena = net.param_arena
        self.exp_avg = torch.zeros_like(arena)
        self.exp_avg_sq = torch.zeros_like(arena)

    def zero_grad(self, set_to_none: bool = True):
        # gradients live in the arena; the trainer clears it with one memset per step
        self.net.grad_arena.zero_()

    def step(self, grad_scale: float = 1.0):
        self.step_count += 1
        net = self.net
        ops.adam_step(net.param_arena, net.grad_arena, self.exp_avg, self.exp_avg_sq, lr=self.lr, beta1=self.betas[0],
                      beta2=self.betas[1], eps=self.eps, step=self.step_count, grad_scale=grad_scale)
        net.mark_weights_dirty()

    # ---- torch.optim.Adam-compatible (de)serialisation ---------------------------------------------
    def _ordered(self):
        return [(n, p) for n, p in self.net.named_parameters()]

    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (n, p) in enumerate(self._ordered()):
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.net.slot_view(self.exp_avg, n).clone(memory_format=torch.contiguous_format),
                            "exp_avg_sq": self.net.slot_view(self.exp_avg_sq, n).clone(memory_format=torch.contiguous_format)}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(self._ordered())))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])
        names = [n for n, _ in self._ordered()]
        if len(g["params"]) != len(names):
            raise ValueError("optimizer state does not match the model's parameter list")
        self.step_count = 0
        for i, n in enumerate(names):
            st = sd["state"].get(i)
            if st is None:
                continue
            self.net.slot_view(self.exp_avg, n).copy_(st["exp_avg"])
            self.net.slot_view(self.exp_avg_sq, n).copy_(st["exp_avg_sq"])
            self.step_count = int(float(st["step"]))
