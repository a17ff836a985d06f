"""CPU oracle for the gyre diffusion generation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gyre_amd/`` may import this package:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker / reported baseline.

What is pinned and what is not:

* Reference-owned arithmetic (per-image RNG, DPM-Solver++(2M), CFG combine,
  sigma tables, sigma<->t, Txt2img latent crop/pad, latent mask pooling) is
  PINNED against golden vectors produced by importing the reference itself in
  the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
* The UNet / VAE forward lives in ``diffusers ~= 0.16.0`` (pyproject.toml:22 of
  the reference), which is neither vendored nor installed; the reference's own
  tests hold no tensors for it.  ``models_ref.py`` restates the published
  architecture from ``gyre/ldm_config/v1-inference.yaml:29-64`` and the
  diffusers state-dict key space.  PARITY UNPINNED at that boundary.
"""
