"""CPU oracle for the HRNet -> decode -> EPnP/RANSAC hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / reported baseline.
The product path (``spacecraft-pose-estimation_amd``) never imports this package
and fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * hrnet_ref   -- pinned against the reference ``lib/models/pose_hrnet.py``
                   imported in the build container (tests/golden/make_golden.py).
  * decode_ref  -- pinned against the reference ``lib/core/inference.py`` imported
                   under a cv2 stub (same script).
  * pnp_ref     -- PARITY UNPINNED: cv2 (opencv-python 3.4.11.41) is a third-party
                   dependency absent from /root/reference and from this image; the
                   restatement follows OpenCV 3.4's published algorithm and is
                   anchored on analytic known-answer cases only.
"""
