"""CPU oracle for the BrepGen denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``brepgen_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.

Pinning status
--------------
* ``oracle.denoisers`` -- PINNED: validated against the reference's own ``network.py`` classes (imported in the build
  container with the missing ``diffusers`` package stubbed out, see ``tests/golden/gen_golden.py``) and against the golden
  vectors that script wrote to ``tests/golden/``.
* ``oracle.ref_formulation`` -- PINNED the same way (``tests/test_oracle_golden.py``): the reference's formulation rebuilt
  from stock ``torch.nn`` (``nn.TransformerEncoder``, seq-first); the like-for-like ``torch.autocast`` parity target on the
  GPU box and the formulation ``bench.py``'s CPU leg times.
* ``oracle.dedup`` -- the reference's numpy de-dup loops (sample.py:159-183, 242-261), restated line for line; checker of
  the device kernels.
* ``oracle.philox`` -- PINNED to the published Philox4x32-10 known-answer vectors (``tests/test_oracle_philox.py``).
* ``oracle.schedulers`` -- PARITY UNPINNED against ``diffusers==0.27`` itself (``requirements.txt:5`` of the reference; not
  vendored under /root/reference, not installable offline) but pinned to the PUBLISHED math in fp64 over the full
  schedules: DDPM posterior of Ho et al. eq. 6-7/15, PNDM transfer / pseudo-RK / PLMS of Liu et al. eq. 11-13
  (``tests/test_oracle_pins.py``), plus the known-answer constants of SURVEY.md App. B.4.  ``tests/golden/pin_diffusers.py``
  diffs it against upstream the moment diffusers is importable.
* ``oracle.vae`` -- the diffusers VAE blocks, restated from the published 0.27 code and anchored on the reference's wiring
  (``network.py:30-299``) and the parameter counts of SURVEY.md App. C.  The 2-D (surface) decoder and encoder are PINNED to
  an independent third-party implementation of the same published latent-diffusion auto-encoder that ships in this image:
  Hugging Face ``transformers``' ``JanusVQVAEDecoder`` / ``JanusVQVAEEncoder`` with BrepGen's hyper-parameters
  (``tests/test_oracle_vae_pin.py``, fp32 round-off).  The 1-D (edge) VAE uses diffusers' dance-diffusion blocks, for which
  no second implementation is available offline: PARITY UNPINNED against diffusers itself.
* ``oracle.joint_opt`` -- the Chamfer offset fit of ``utils.py:746-772``: loss semantics PARITY UNPINNED (``chamferdist``,
  unversioned third-party CUDA package), optimiser + gradient PINNED against ``torch.optim.AdamW`` + autograd
  (``tests/test_oracle_joint_opt.py``).
"""
