"""Native training step of the VAE hot path: the body of the reference's ``train_epoch`` loop
(``vae_scripts/train_vae.py:380-445``) run directly on the HIP engine, without the autograd tape:

    zero_grad -> forward (encode, sample, decode) -> recon (L1|L2) + kl_weight*KL (+ gamma * AR-VAE term)
              -> backward -> gradient all-reduce (overlapped with backward, buckets of the flat arena) -> Adam

``prepare_batch`` is the reference's ``_prepare_batch`` (train_vae.py:183-243): every container form a dataloader
may hand over (tensor / ``(images, attributes)`` / list of ``(image, attributes)`` pairs / ``[images, dict]``) becomes
device tensors.  The AR-VAE term (``compute_ar_vae_loss``, losses.py:69-166, called at train_vae.py:408-415 on
``z_mu.mean(dim=(2, 3))``) runs as one HIP kernel (``pti_ar_vae_loss``) that also adds its gradient to ``d z_mu``.

The adversarial branch (train_vae.py:399-401 generator term, :447-458 discriminator step; active when ``adv_enabled``
and ``epoch > 5``) runs on the PatchDiscriminator engine (``disc_engine.py``): pass ``discriminator=`` /
``adv_weight=`` and call ``step(..., adversarial=True)``.  The perceptual (LPIPS) term (train_vae.py:395-397) is a
torch module the caller supplies (``perceptual=``, e.g. ``models.perceptual.PerceptualLoss`` with locally provided
weights -- they cannot be fetched here): it is evaluated on the reconstruction under autograd and its
gradient joins ``d_recon`` before the HIP backward.  No ``.item()`` on the step path: loss scalars come back as device
tensors.
"""
from __future__ import annotations

import collections
import os
import random

import torch

from . import ops
from .data_parallel import FlatGradAllReducer, broadcast_parameters
from .optim import FlatAdam


def prepare_batch(batch, device, ar_vae_enabled: bool):
    """Reference ``_prepare_batch`` (train_vae.py:183-243): -> ``(images on device, attributes dict on device | None)``.

    Accepted: a tensor; a tuple ``(images, attributes)``; a list of ``(image, attributes)`` pairs (a collate that did
    not stack); a two-element list ``[images, attributes dict]``.  Raises ``ValueError`` for an empty list or for AR-VAE
    without attributes, ``TypeError`` for anything else -- the reference's exception types and messages."""
    attrs = None
    if isinstance(batch, list):
        if not batch:
            raise ValueError("Empty batch received from dataloader.")
        if all(isinstance(it, tuple) and len(it) == 2 for it in batch):
            images = torch.stack([torch.as_tensor(img) for img, _ in batch], dim=0)
            buf: dict[str, list] = {}
            for _, a in batch:
                if a is not None:
                    for k, v in a.items():
                        buf.setdefault(k, []).append(torch.as_tensor(v))
            if buf:
                attrs = {k: torch.stack(v, dim=0).to(torch.float32) for k, v in buf.items()}
        elif len(batch) == 2 and isinstance(batch[0], torch.Tensor) and isinstance(batch[1], dict):
            images, attrs = batch[0], {k: torch.as_tensor(v) for k, v in batch[1].items()}
        else:
            raise TypeError(f"Unsupported list batch elements: {[type(it) for it in batch]}")
    elif isinstance(batch, tuple):
        images, attrs = batch
    else:
        images = batch
    if not isinstance(images, torch.Tensor):
        raise TypeError(f"Unsupported batch type: {type(images)}")
    images = images.to(device, non_blocking=True)
    if attrs is not None:
        attrs = {k: v.to(device, non_blocking=True) for k, v in attrs.items()}
    elif ar_vae_enabled:
        raise ValueError("AR-VAE is enabled but attributes are missing from the batch.")
    return images, attrs


class ARSettings:
    """The AR-VAE block of a run (``regularized_attributes`` of the config, train_vae.py:376-378,776-792) resolved once:
    attribute names in mapping order, their latent channels and tanh slopes (per-attribute ``delta`` or the
    ``delta_global`` fall-back, losses.py:118-128), pair mode and gamma.  Validation errors are the reference's."""

    def __init__(self, attribute_latent_mapping: dict, *, gamma: float, pairwise: str = "all", subset_pairs=None,
                 delta_global: dict | None = None, latent_channels: int | None = None):
        if pairwise not in {"all", "subset"}:
            raise ValueError(f"pairwise must be 'all' or 'subset', got {pairwise}")
        if pairwise == "subset" and (subset_pairs is None or subset_pairs <= 0):
            raise ValueError("subset_pairs must be a positive integer when pairwise='subset'")
        self.names, self.channels, self.deltas = [], [], []
        for name, m in attribute_latent_mapping.items():
            if str(name).startswith("_"):
                continue
            ch = int(m["latent_channel"])
            if latent_channels is not None and ch >= latent_channels:
                raise ValueError(f"Latent channel {ch} for attribute {name} exceeds latent size {latent_channels}")
            delta = m.get("delta")
            if delta is None and delta_global and delta_global.get("enabled", False):
                delta = delta_global.get("value")
            if delta is None:
                raise ValueError(f"Delta not provided for {name} and no delta_global fallback.")
            self.names.append(name)
            self.channels.append(ch)
            self.deltas.append(float(delta))
        if not self.names:
            raise ValueError("attribute_latent_mapping must be provided when AR-VAE is enabled.")
        self.gamma, self.pairwise, self.subset_pairs = float(gamma), pairwise, subset_pairs

    @classmethod
    def from_config(cls, regularized_attributes: dict, gamma: float, latent_channels: int | None = None):
        ra = regularized_attributes or {}
        return cls(ra.get("attribute_latent_mapping", {}), gamma=gamma, pairwise=ra.get("pairwise", "all"),
                   subset_pairs=ra.get("subset_pairs"), delta_global=ra.get("delta_global", {}),
                   latent_channels=latent_channels)


class VAETrainer:
    def __init__(self, model, *, lr: float, world_size: int = 1, process_group=None, recon_loss: str = "l1",
                 kl_weight: float = 1e-3, kl_input_is_logvar: bool = True, bucket_bytes: int = 4 << 20,
                 rank_eps_offset: int = 0, ar: ARSettings | None = None, discriminator=None, adv_weight: float = 0.0,
                 lr_d: float | None = None, adv_no_activation_leastsq: bool = False, perceptual=None,
                 perceptual_weight: float = 0.0):
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
        # HIP-graph mode for the plain step (one GPU, no AR / perceptual / adversarial term): forward + loss +
        # backward of a fixed batch shape are captured once and replayed, Adam stays eager (its bias corrections change
        # every step).  Host time per step 5.2 -> ~3 ms: matters where the step is host-bound (batch <= 8 on config A).
        # PTI_STEP_GRAPH: 1 = always, 0 = never, unset = "auto": only where the step is host-bound -- at most 6 x 256^2
        # pixels per batch (measured on config A: batch 4 5.52 -> 4.50 ms, batch 6 5.46 -> 5.13, batch 8 5.30 -> 5.71,
        # batch 32 12.58 -> 13.05: the graph executes the two-stream DAG with less overlap than the eager streams)
        self.step_graph = {"1": True, "0": False}.get(os.environ.get("PTI_STEP_GRAPH", ""), "auto")
        self._graphs, self._eager_steps = {}, 0
        # adversarial branch: PatchDiscriminator with its own flat Adam (train_vae.py:304: same lr x world) and its own
        # gradient exchange (one bucket: the discriminator is 2.8 MB of fp32 gradients)
        self.disc, self.adv_weight = discriminator, float(adv_weight)
        # PatchAdversarialLoss("least_squares"): LeakyReLU(0.05) on the logits unless no_activation_leastsq (MONAI's option)
        self.adv_slope = 1.0 if adv_no_activation_leastsq else 0.05
        if discriminator is not None:
            self.disc_eng = discriminator.engine()
            discriminator.attach_grads()
            self.opt_d = FlatAdam(discriminator, (lr if lr_d is None else lr_d) * world_size)
            self.reducer_d = FlatGradAllReducer(discriminator.grad_arena, process_group, bucket_bytes)
            broadcast_parameters(discriminator.param_arena, process_group)
            discriminator.mark_weights_dirty()
            # the discriminator's own step (losses on fake / real, backward, exchange, Adam) depends on the reconstruction
            # only: it runs on its own stream beside the VAE backward (PTI_ADV_STREAM=0: after Adam on the main stream)
            dev_ = discriminator.param_arena.device
            self._adv_stream = (torch.cuda.Stream(device=dev_)
                                if dev_.type == "cuda" and os.environ.get("PTI_ADV_STREAM", "1") != "0" else None)
        # perceptual term: any torch module  f(reconstruction, images) -> scalar  on the device (frozen weights)
        self.perceptual, self.perceptual_weight = perceptual, float(perceptual_weight)
        if perceptual is None and self.perceptual_weight != 0.0:
            raise ValueError("perceptual_weight != 0 needs a perceptual loss module (VAETrainer(perceptual=...))")
        self.ar = ar
        if ar is not None:
            if max(ar.channels) >= net.latent_channels:
                raise ValueError(f"AR-VAE latent channel {max(ar.channels)} exceeds latent size {net.latent_channels}")
            dev = net.param_arena.device
            self._ar_ch = torch.tensor(ar.channels, dtype=torch.int32, device=dev)
            self._ar_delta = torch.tensor(ar.deltas, dtype=torch.float32, device=dev)

    # ---- AR-VAE term -------------------------------------------------------------------------------------------
    def _ar_inputs(self, attributes, batch):
        """[na, b] attribute table (mapping order) and, for pairwise="subset", the sampled-pair mask: Python
        ``random.sample`` over the ordered pair list, one draw per attribute -- the reference's sampling (losses.py:132-136)."""
        ar = self.ar
        if attributes is None:
            raise ValueError("AR-VAE is enabled but attributes are missing from the batch.")
        rows = []
        for name in ar.names:
            a = attributes.get(name)
            if a is None:
                raise KeyError(f"Missing attribute values for {name} in batch.")
            rows.append(a.to(self._ar_delta.device, torch.float32).reshape(-1))
        table = torch.stack(rows, 0).contiguous()
        if table.shape[1] != batch:
            raise ValueError(f"attributes hold {table.shape[1]} values per name for a batch of {batch}")
        mask = None
        if ar.pairwise == "subset":
            pairs = [(i, j) for i in range(batch) for j in range(batch) if i != j]
            m = torch.zeros(len(ar.names), batch, batch, dtype=torch.uint8)
            for q in range(len(ar.names)):
                for i, j in random.sample(pairs, min(len(pairs), int(ar.subset_pairs))):
                    m[q, i, j] = 1
            mask = m.to(table.device)
        return table, mask

    def _ar_term(self, mu, attributes, d_mu):
        """-> (sum of the per-attribute losses, per-attribute losses [na], pair counts [na]) as device tensors; adds
        gamma * gradient into ``d_mu`` when given."""
        table, mask = self._ar_inputs(attributes, mu.shape[0])
        na = table.shape[0]
        per = torch.empty(na, dtype=torch.float32, device=mu.device)
        cnt = torch.empty(na, dtype=torch.int32, device=mu.device)
        ops.ar_vae_loss(mu, table, self._ar_ch, self._ar_delta, per, cnt, gamma=self.ar.gamma, d_mu=d_mu, pair_mask=mask)
        return per.sum(), per, cnt

    # ---- adversarial branch -----------------------------------------------------------------------------------------
    def _adv_generator_term(self, recon, d_recon, accumulate=True):
        """train_vae.py:399-401: adv_loss(discriminator(recon)[-1], target_is_real=True, for_discriminator=False); adds
        adv_weight * its gradient into ``d_recon`` (when given; ``accumulate=False``: writes it there instead).
        Returns (loss, the pass's context)."""
        eng = self.disc_eng
        ctx = eng.forward(recon, save=d_recon is not None)
        loss, d = eng.lsgan(ctx, target_is_real=True, weight=self.adv_weight, want_grad=d_recon is not None,
                            slope=self.adv_slope)
        if d_recon is not None:
            eng.backward(ctx, d, want_wgrad=False, d_img=d_recon, accumulate_dx=accumulate)
        return loss, ctx

    def _adv_discriminator_losses(self, fake_ctx, images, train: bool):
        """train_vae.py:447-458: 0.5 * (adv_loss(D(recon.detach()), fake) + adv_loss(D(images), real)); the fake pass is
        the generator term's (the discriminator's weights have not changed in between).  ``train``: also the backward
        passes, gradient exchange and the discriminator's Adam step on adv_weight * that loss."""
        eng, disc = self.disc_eng, self.disc
        if train:
            disc.grad_arena.zero_()
            self.reducer_d.begin_step()
        l_fake, d_fake = eng.lsgan(fake_ctx, target_is_real=False, weight=0.5 * self.adv_weight, want_grad=train,
                                   slope=self.adv_slope)
        if train:
            eng.backward(fake_ctx, d_fake, want_wgrad=True)
        real_ctx = eng.forward(images.float(), save=train)
        l_real, d_real = eng.lsgan(real_ctx, target_is_real=True, weight=0.5 * self.adv_weight, want_grad=train,
                                   slope=self.adv_slope)
        if train:
            eng.backward(real_ctx, d_real, want_wgrad=True)
            if self.world > 1:
                self.reducer_d.ready(0, disc.grad_arena.numel())
            self.reducer_d.finish()
            self.opt_d.step(grad_scale=1.0 / self.world)
        return 0.5 * (l_fake + l_real)

    def _perceptual_target_taps(self, images):
        """The target half of the perceptual term does not depend on the reconstruction: enqueue it on the side stream
        (idle during the VAE forward) at the start of the step.  -> (taps, event) or None (no side stream / module
        without the split API)."""
        side = self.eng.wgrad_stream
        if side is None or not hasattr(self.perceptual, "target_taps"):
            return None
        main = torch.cuda.current_stream()
        side.wait_stream(main)                       # the images were produced on the main stream
        with torch.cuda.stream(side):
            taps = self.perceptual.target_taps(images)
            ev = torch.cuda.Event()
            ev.record(side)
        for t in taps:                               # allocated on the side stream, read on the main one
            t.record_stream(main)
        return taps, ev

    def _perceptual_term(self, recon, images, d_recon, target=None):
        """train_vae.py:395-397: p_loss = loss_perceptual(ensure_three_channels(recon), ensure_three_channels(images));
        adds perceptual_weight * d p_loss / d recon into ``d_recon`` (when given).  ``target``: what
        ``_perceptual_target_taps(images)`` returned at the start of the step."""
        from .utils.losses import ensure_three_channels as three
        if hasattr(self.perceptual, "from_taps"):     # this package's module repeats the channel itself -- and takes
            three = lambda t: t                       # one-channel images through its folded first layer  # noqa: E731
        if d_recon is None:
            with torch.no_grad():
                return self.perceptual(three(recon.float()), three(images.float()))
        r = recon.detach().requires_grad_(True)
        with torch.enable_grad():
            if target is not None:
                torch.cuda.current_stream().wait_event(target[1])
                p = self.perceptual.from_taps(three(r.float()), target[0])
            else:
                p = self.perceptual(three(r.float()), three(images.float()))
            (g,) = torch.autograd.grad(p, r)
        d_recon.add_(g, alpha=self.perceptual_weight)
        return p.detach()

    def _plain_fwd_dec_bwd(self, images, eps, join):
        """First half of the plain step: zero_grad -> forward -> L1|L2 + kl_weight*KL -> DECODER backward, without any
        optional term and without host-side decisions.  ``join``: wait for the side-stream weight gradients, so that the
        post_quant + decoder region of the gradient arena is final when this returns (the data-parallel graph mode
        all-reduces it between its two captured halves).  Returns (the [recon, kl] device pair, what the second half needs)."""
        net, eng = self.net, self.eng
        net.grad_arena.zero_()
        mu, sigma, c_enc = eng.encode_forward(images, save=True)
        z = torch.addcmul(mu, eps, sigma)
        recon, c_dec = eng.decode_forward(z, save=True)
        third = sigma if net.third_output == "sigma" else 2.0 * torch.log(sigma)
        out2 = torch.zeros(2, dtype=torch.float32, device=recon.device)
        d_recon, d_mu, d_third = torch.empty_like(recon), torch.empty_like(mu), torch.empty_like(third)
        ops.vae_loss(recon, images, mu, third, out2, d_recon, d_mu, d_third, l2=self.l2, third_mode=self.third_mode,
                     kl_weight=self.kl_weight)
        dz = eng.decode_backward(c_dec, d_recon, want_dz=True, join=join)
        d_sigma = d_third if net.third_output == "sigma" else d_third * (2.0 / sigma)
        d_mu = d_mu + dz
        d_sigma = torch.addcmul(d_sigma, dz, eps)
        return out2, (c_enc, d_mu, d_sigma)

    def _plain_enc_bwd(self, carry):
        """Second half: the ENCODER backward (joins the side stream)."""
        c_enc, d_mu, d_sigma = carry
        self.eng.encode_backward(c_enc, d_mu, d_sigma, want_dx=False)

    def _plain_fwd_bwd(self, images, eps):
        """The part of the plain step ONE HIP graph can hold (single GPU).  Returns the [recon, kl] device pair."""
        out2, carry = self._plain_fwd_dec_bwd(images, eps, join=False)    # encode_backward's join covers the side stream
        self._plain_enc_bwd(carry)
        return out2

    def _step_graphed(self, images, eps):
        """One plain step through a captured graph (see ``step_graph``); None when this call must run eagerly (the first
        two steps, which settle the lazily built state, and more than four batch shapes)."""
        # everything the captured launches bake in besides the weights: shape, loss settings, stream layout
        key = (tuple(images.shape), self.l2, self.kl_weight, self.third_mode, self.eng.wgrad_stream is None,
               self.eng.wgrad_batch_max)
        st = self._graphs.get(key)
        if st is None:
            if self._eager_steps < 2 or len(self._graphs) >= 4 or ops.KERNEL_PROFILE is not None:
                return None
            net = self.net
            down = 2 ** (len(net.channels) - 1)
            gx = images.detach().float().contiguous().clone()
            geps = torch.zeros(images.shape[0], net.latent_channels, images.shape[2] // down, images.shape[3] // down,
                               dtype=torch.float32, device=images.device)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # the weight re-pack launches must be PART of the graph (every replay follows an optimiser step): if a
            # no_grad forward (validation, eval_losses) ran just before this call the packs are clean and
            # refresh_weights() would return early during capture -- every replay would then train on the weights as
            # last packed eagerly.  Force the dirty state for the capture.
            self.eng.packed_version = -1
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):   # loader threads may issue copies meanwhile
                    out2 = self._plain_fwd_bwd(gx, geps)
            except Exception as ex:   # a capture that fails must not take the run down: eager from here on
                import warnings
                warnings.warn(f"HIP-graph capture of the training step failed ({ex!r}); continuing with eager launches")
                self.step_graph = False
                torch.cuda.synchronize()
                return None
            st = self._graphs[key] = (g, gx, geps, out2)
        g, gx, geps, out2 = st
        gx.copy_(images)
        if eps is None:
            geps.normal_(generator=self.gen)        # the same draws torch.randn(..., generator=self.gen) makes eagerly
        else:
            geps.copy_(eps)
        g.replay()
        return out2.clone()

    def _step_graphed_dp(self, images, eps):
        """Data-parallel form of ``_step_graphed`` (world > 1): the step is captured as TWO graphs with the collectives
        eager between them -- [forward, loss, decoder backward] | all-reduce of the post_quant + decoder region in
        ~bucket-sized asynchronous pieces | [encoder backward] | all-reduce of the encoder + quant region | join | Adam.
        The decoder region's exchange (3/4 of the gradient bytes of config A) runs under the second graph, as the eager
        callbacks arrange it; what the graphs remove is the eager step's ~5 ms of host enqueue per rank, which at N = 8
        is no longer hidden behind a single process's GPU time.  None when this call must run eagerly.
        EVERY RANK MUST TAKE THE SAME PATH on the same step (the two paths cut the arena into different buckets): the
        decision below depends only on state that is identical across ranks in a symmetric job -- the step count, the batch
        shape, ``step_graph`` -- plus ``ops.KERNEL_PROFILE``, which a caller that profiles one rank must pair with
        ``step_graph = False`` on all of them (bench.py does)."""
        key = ("dp", tuple(images.shape), self.l2, self.kl_weight, self.third_mode, self.eng.wgrad_stream is None,
               self.eng.wgrad_batch_max)
        st = self._graphs.get(key)
        if st is None:
            if self._eager_steps < 2 or len(self._graphs) >= 4 or ops.KERNEL_PROFILE is not None:
                return None
            net = self.net
            down = 2 ** (len(net.channels) - 1)
            gx = images.detach().float().contiguous().clone()
            geps = torch.zeros(images.shape[0], net.latent_channels, images.shape[2] // down, images.shape[3] // down,
                               dtype=torch.float32, device=images.device)
            torch.cuda.synchronize()
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            self.eng.packed_version = -1          # the re-pack launches are part of the first graph (see _step_graphed)
            self.eng.grad_ready_cb = None
            try:
                with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                    out2, carry = self._plain_fwd_dec_bwd(gx, geps, join=True)
                with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
                    self._plain_enc_bwd(carry)
            except Exception as ex:
                import warnings
                warnings.warn(f"HIP-graph capture of the data-parallel step failed ({ex!r}); continuing with eager launches")
                self.step_graph = False
                torch.cuda.synchronize()
                return None
            st = self._graphs[key] = (ga, gb, gx, geps, out2, carry)     # `carry` keeps the tensors crossing the seam alive
        ga, gb, gx, geps, out2, _ = st
        gx.copy_(images)
        if eps is None:
            geps.normal_(generator=self.gen)
        else:
            geps.copy_(eps)
        red = self.reducer
        (e_lo, e_hi), (d_lo, d_hi) = self.net.arena_regions()
        red.begin_step()
        ga.replay()
        red.reduce_range(d_lo, d_hi)          # asynchronous: runs under the second graph
        gb.replay()
        red.reduce_range(e_lo, e_hi)
        red.finish()
        return out2.clone()

    def step(self, images: torch.Tensor, eps: torch.Tensor | None = None, attributes: dict | None = None,
             adversarial: bool = False):
        """One optimiser step on ``images`` [B,C,H,W] fp32 (already on the device).  Returns a dict of
        DEVICE scalars {"loss", "recon", "kl"} -- plus {"ar", "ar_per_attr", "ar_pairs"} (names: ``self.ar.names``) when
        the AR-VAE term is on, which needs ``attributes`` = {name: [B] tensor} -- with no host sync with THIS step
        (see ``max_steps_in_flight``).  ``adversarial`` (the reference's ``adv_enabled and epoch > 5``): the generator
        loss gains adv_weight * the least-squares term through the discriminator, then the discriminator takes its
        own optimiser step; adds {"adv_gen", "adv_disc"} (unweighted, as the reference logs them before weighting).
        With a perceptual module and a non-zero ``perceptual_weight`` the result also holds {"perceptual"}."""
        if adversarial and self.disc is None:
            raise ValueError("step(adversarial=True) needs VAETrainer(discriminator=...)")
        net, eng, red = self.net, self.eng, self.reducer
        while len(self._step_done) >= max(1, self.max_steps_in_flight):
            self._step_done.popleft().synchronize()
        plain = (self.world == 1 and self.ar is None and not adversarial
                 and not (self.perceptual is not None and self.perceptual_weight != 0.0))
        use_graph = self.step_graph is True or (self.step_graph == "auto" and images.shape[0] * images.shape[2] * images.shape[3] <= 6 * 65536)
        plain_dp = (self.world > 1 and self.ar is None and not adversarial
                    and not (self.perceptual is not None and self.perceptual_weight != 0.0))
        if use_graph and (plain or plain_dp) and ops.KERNEL_PROFILE is None:
            out2 = self._step_graphed(images, eps) if plain else self._step_graphed_dp(images, eps)
            if out2 is not None:
                self.opt.step(grad_scale=1.0 / self.world)
                done = torch.cuda.Event()
                done.record()
                self._step_done.append(done)
                return {"loss": out2[0] + self.kl_weight * out2[1], "recon": out2[0], "kl": out2[1]}
        self._eager_steps += 1
        net.grad_arena.zero_()
        red.begin_step()
        eng.grad_ready_cb = red.ready if self.world > 1 else None
        try:
            p_target = None
            if self.perceptual is not None and self.perceptual_weight != 0.0:
                p_target = self._perceptual_target_taps(images)     # side stream, under the VAE forward
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
            ar_out = None
            if self.ar is not None:   # + gamma * AR-VAE(z_mu.mean(h, w)): its gradient goes straight into d_mu
                ar_out = self._ar_term(mu, attributes, d_mu)
            p_loss = None
            with_p = self.perceptual is not None and self.perceptual_weight != 0.0
            adv_ctx = adv_disc = adv_done = None
            if adversarial and with_p and self._adv_stream is not None:
                # both optional terms depend on the reconstruction only: the generator's adversarial term runs on the
                # discriminator's stream (followed there by the discriminator's own step) while the perceptual term
                # runs here; the gradients join in the serial schedule's order, (VAE + perceptual) + adversarial
                main, advs = torch.cuda.current_stream(), self._adv_stream
                advs.wait_stream(main)
                with torch.cuda.stream(advs):
                    d_adv = torch.empty_like(d_recon)
                    adv_gen, adv_ctx = self._adv_generator_term(recon, d_adv, accumulate=False)
                    g_done = torch.cuda.Event()
                    g_done.record(advs)
                    adv_disc = self._adv_discriminator_losses(adv_ctx, images, train=True)
                    adv_done = torch.cuda.Event()
                    adv_done.record(advs)
                for t in (d_adv, adv_gen, adv_disc):
                    t.record_stream(main)
                p_loss = self._perceptual_term(recon, images, d_recon, p_target)
                main.wait_event(g_done)
                d_recon.add_(d_adv)
            else:
                if with_p:
                    p_loss = self._perceptual_term(recon, images, d_recon, p_target)
                if adversarial:       # + adv_weight * generator term: its gradient w.r.t. the reconstruction joins d_recon
                    adv_gen, adv_ctx = self._adv_generator_term(recon, d_recon)
                    if self._adv_stream is not None:     # the discriminator's step, on its stream, under the VAE backward
                        main = torch.cuda.current_stream()
                        self._adv_stream.wait_stream(main)
                        with torch.cuda.stream(self._adv_stream):
                            adv_disc = self._adv_discriminator_losses(adv_ctx, images, train=True)
                            adv_done = torch.cuda.Event()
                            adv_done.record(self._adv_stream)
                        adv_disc.record_stream(main)
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
        if adversarial:
            if adv_done is not None:
                torch.cuda.current_stream().wait_event(adv_done)   # its loss and the updated discriminator, before returning
            else:
                adv_disc = self._adv_discriminator_losses(adv_ctx, images, train=True)
        done = torch.cuda.Event()
        done.record()
        self._step_done.append(done)
        res = {"loss": out2[0] + self.kl_weight * out2[1], "recon": out2[0], "kl": out2[1]}
        if ar_out is not None:
            res["ar"], res["ar_per_attr"], res["ar_pairs"] = ar_out
            res["loss"] = res["loss"] + self.ar.gamma * ar_out[0]
        if p_loss is not None:
            res["perceptual"] = p_loss
            res["loss"] = res["loss"] + self.perceptual_weight * p_loss
        if adversarial:
            res["adv_gen"], res["adv_disc"] = adv_gen[0], adv_disc[0]
            res["loss"] = res["loss"] + self.adv_weight * adv_gen[0]
        return res

    @torch.no_grad()
    def eval_losses(self, images: torch.Tensor, attributes: dict | None = None, adversarial: bool = False):
        """Validation forward as the reference does it (``validate``: SAMPLED forward under no_grad,
        train_vae.py:555-560; adversarial terms :564-571; AR-VAE term :573-589).  Returns device scalars {"recon",
        "kl"[, "ar", ...][, "adv_gen", "adv_disc"]} and the reconstruction."""
        mu, sigma, _ = self.eng.encode_forward(images, save=False)
        eps = torch.randn(sigma.shape, generator=self.gen, device=sigma.device, dtype=sigma.dtype)
        recon, _ = self.eng.decode_forward(torch.addcmul(mu, eps, sigma), save=False)
        third = sigma if self.net.third_output == "sigma" else 2.0 * torch.log(sigma)
        out2 = torch.zeros(2, dtype=torch.float32, device=recon.device)
        ops.vae_loss(recon, images.contiguous().float(), mu, third, out2, None, None, None, l2=self.l2,
                     third_mode=self.third_mode, kl_weight=self.kl_weight)
        res = {"recon": out2[0], "kl": out2[1]}
        if self.ar is not None:
            res["ar"], res["ar_per_attr"], res["ar_pairs"] = self._ar_term(mu, attributes, None)
        if self.perceptual is not None and self.perceptual_weight != 0.0:
            res["perceptual"] = self._perceptual_term(recon, images, None)
        if adversarial:
            if self.disc is None:
                raise ValueError("eval_losses(adversarial=True) needs VAETrainer(discriminator=...)")
            adv_gen, ctx = self._adv_generator_term(recon, None)
            res["adv_gen"], res["adv_disc"] = adv_gen[0], self._adv_discriminator_losses(ctx, images, train=False)[0]
        return res, recon
