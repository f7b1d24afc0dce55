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
"""
