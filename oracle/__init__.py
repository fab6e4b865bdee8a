"""TEST INFRASTRUCTURE ONLY — CPU fp32 oracle for the VAE training hot path.

Nothing under ``oracle/`` may be imported by the product package
(``pti_ldm_vae_amd``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

Parity status: the encoder/decoder arithmetic of the reference lives in MONAI
1.5.1 (``uv.lock:859-860``), which is absent from the reference tree and from
this image, and the reference has no tests: **parity unpinned** for the
encoder/decoder (restated from SURVEY.md Appendix A).  The loss functions
(``src/pti_ldm_vae/models/losses.py``) ARE pinned: ``oracle/make_golden.py``
imports that file by path in the build container and commits its outputs as
fixtures under ``tests/golden/``.  ``oracle/perceptual.py`` (LPIPS / SqueezeNet-1.1 term, the checker of the
perceptual tests since round 3) and ``oracle/patch_discriminator.py`` are **parity unpinned** for the same reason
as the encoder/decoder (lpips / torchvision / MONAI absent, no weights, no reference output).
"""
