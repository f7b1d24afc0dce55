"""CPU oracle for the BrepGen denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``brepgen_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.

Pinning status
--------------
* ``oracle.denoisers`` — PINNED: validated against the reference's own
  ``network.py`` classes (imported in the build container with the missing
  ``diffusers`` package stubbed out, see ``tests/golden/gen_golden.py``) and against
  the golden vectors that script wrote to ``tests/golden/``.
* ``oracle.schedulers`` — PARITY UNPINNED: the arithmetic lives in the
  third-party ``diffusers==0.27`` package (``requirements.txt:5`` of the
  reference), which is neither vendored under /root/reference nor installable
  offline.  The restatement follows the published 0.27 algorithm and the
  reference's call sites (``sample.py:101-117,128-153``); it is checked by
  self-consistency tests and the known-answer constants in SURVEY.md App. B.4.
* ``oracle.vae`` — PARITY UNPINNED for the same reason (the decoder / encoder blocks are diffusers code); restated from
  the published 0.27 blocks, anchored on the reference's wiring (``network.py:30-299``) and on the parameter counts
  SURVEY.md App. C records (asserted in ``tests/test_oracle_vae.py``).
* ``oracle.joint_opt`` — the Chamfer offset fit of ``utils.py:746-772``: loss semantics PARITY UNPINNED (``chamferdist``,
  unversioned third-party CUDA package), optimiser + gradient PINNED against ``torch.optim.AdamW`` + autograd
  (``tests/test_oracle_joint_opt.py``).
"""
