"""CPU oracle for the trajectory-diffusion sampling path.

TEST INFRASTRUCTURE ONLY.  This package is an independent fp32 CPU restatement
(torch CPU tensors / NumPy float64 for the schedules) of the reference
algorithm.  It is imported by ``tests/``, by ``__graft_entry__.smoke()`` and by
the ``cpu_baseline`` leg of ``bench.py`` -- only as the checker or as the
reported CPU baseline, never by the product package ``autodiffusion_amd``.

Parity status: PINNED.  Every function is checked against golden vectors
recorded from the reference itself (``tests/golden/*.npz``, produced by
``tests/golden/make_golden.py``), except the diffusers-style scheduler steps in
``oracle.sampling`` (third-party, absent, unpinned version): those are marked
"parity unpinned" and are covered by analytic known-answer tests only.
"""
