"""Native training step of the VAE hot path: the body of the reference's ``train_epoch`` loop
(``vae_scripts/train_vae.py:380-445``) run directly on the HIP engine, without the autograd tape:

    zero_grad -> forward (encode, sample, decode) -> recon (L1|L2) + kl_weight*KL  -> backward
              -> gradient all-reduce (overlapped with backward, buckets of the flat arena) -> Adam

Perceptual (LPIPS) and adversarial terms are NOT part of this path (unavailable offline / inactive
before epoch 6 — SURVEY.md §2); ``perceptual_weight`` must be 0 here and the drop-in autograd path
(``VAEModel.forward`` + any torch loss) remains available for everything else.  No ``.item()`` on
the step path: loss scalars come back as device tensors.
"""
from __future__ import annotations

import collections
import os

import torch

from . import ops
from .data_parallel import FlatGradAllReducer, broadcast_parameters
from .optim import FlatAdam


class VAETrainer:
    def __init__(self, model, *, lr: float, world_size: int = 1, process_group=None, recon_loss: str = "l1",
                 kl_weight: float = 1e-3, kl_input_is_logvar: bool = True, bucket_bytes: int = 4 << 20,
                 rank_eps_offset: int = 0):
        self.model = model
        self.net = net = model.autoencoder
        self.eng = net.engine()
        self.world = world_size
        # reference: lr scaled by world size (train_vae.py:301); gradients averaged like DDP
        self.opt = FlatAdam(net, lr * world_size)
        self.l2 = recon_loss == "l2"
        self.kl_weight = float(kl_weight)
        self.third_mode = 0 if kl_input_is_logvar else 1
        net.attach_grads()
        self.reducer = FlatGradAllReducer(net.grad_arena, process_group, bucket_bytes)
        if self.reducer.world != world_size:
            raise ValueError(f"world_size={world_size} but the process group has {self.reducer.world} ranks")
        broadcast_parameters(net.param_arena, process_group)
        net.mark_weights_dirty()
        self.gen = torch.Generator(device=net.param_arena.device)
        self.gen.manual_seed(42 + rank_eps_offset)
        self.eng.grad_ready_cb = None
        # the host enqueues a step in about half the time the GPU needs for it; left alone it runs ahead until the
        # HIP queues saturate, and that showed up as ONE 0.3-0.5 s host stall some 15 steps into a run
        # (tools/step_jitter.py).  The step therefore waits for the step issued `max_steps_in_flight` steps earlier:
        # the GPU always has the next step queued, the host never gets further ahead than that.
        self.max_steps_in_flight = int(os.environ.get("PTI_MAX_STEPS_IN_FLIGHT", "2"))
        self._step_done = collections.deque()

    def step(self, images: torch.Tensor, eps: torch.Tensor | None = None):
        """One optimiser step on ``images`` [B,C,H,W] fp32 (already on the device).  Returns a dict of
        DEVICE scalars {"loss", "recon", "kl"} (no host sync with THIS step; see ``max_steps_in_flight``)."""
        net, eng, red = self.net, self.eng, self.reducer
        while len(self._step_done) >= max(1, self.max_steps_in_flight):
            self._step_done.popleft().synchronize()
        net.grad_arena.zero_()
        red.begin_step()
        eng.grad_ready_cb = red.ready if self.world > 1 else None
        try:
            mu, sigma, c_enc = eng.encode_forward(images, save=True)
            if eps is None:
                eps = torch.randn(sigma.shape, generator=self.gen, device=sigma.device, dtype=sigma.dtype)
            z = torch.addcmul(mu, eps, sigma)
            recon, c_dec = eng.decode_forward(z, save=True)
            third = sigma if net.third_output == "sigma" else 2.0 * torch.log(sigma)
            out2 = torch.zeros(2, dtype=torch.float32, device=recon.device)
            d_recon, d_mu, d_third = torch.empty_like(recon), torch.empty_like(mu), torch.empty_like(third)
            ops.vae_loss(recon, images.contiguous().float(), mu, third, out2, d_recon, d_mu, d_third, l2=self.l2,
                         third_mode=self.third_mode, kl_weight=self.kl_weight)
            dz = eng.decode_backward(c_dec, d_recon, want_dz=True, join=False)   # encode_backward joins the side stream
            # z = mu + eps*sigma ; third = sigma (or 2 log sigma)
            d_sigma = d_third if net.third_output == "sigma" else d_third * (2.0 / sigma)
            d_mu = d_mu + dz
            d_sigma = torch.addcmul(d_sigma, dz, eps)
            eng.encode_backward(c_enc, d_mu, d_sigma, want_dx=False)
            red.finish()
        finally:
            eng.grad_ready_cb = None
        self.opt.step(grad_scale=1.0 / self.world)
        done = torch.cuda.Event()
        done.record()
        self._step_done.append(done)
        return {"loss": out2[0] + self.kl_weight * out2[1], "recon": out2[0], "kl": out2[1]}

    @torch.no_grad()
    def eval_losses(self, images: torch.Tensor):
        """Validation forward as the reference does it (``validate``: SAMPLED forward under no_grad,
        train_vae.py:555-560).  Returns device scalars {"recon", "kl"} and the reconstruction."""
        mu, sigma, _ = self.eng.encode_forward(images, save=False)
        eps = torch.randn(sigma.shape, generator=self.gen, device=sigma.device, dtype=sigma.dtype)
        recon, _ = self.eng.decode_forward(torch.addcmul(mu, eps, sigma), save=False)
        third = sigma if self.net.third_output == "sigma" else 2.0 * torch.log(sigma)
        out2 = torch.zeros(2, dtype=torch.float32, device=recon.device)
        ops.vae_loss(recon, images.contiguous().float(), mu, third, out2, None, None, None, l2=self.l2,
                     third_mode=self.third_mode, kl_weight=self.kl_weight)
        return {"recon": out2[0], "kl": out2[1]}, recon
