"""ORACLE -- test infrastructure only.

CPU restatement of the reference's forward hot path, used as the parity checker by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.  The
product package (``seam-match-rcnn_amd``) never imports anything from here.

  heads.py      in-tree heads (models/match_head.py, models/nlb.py) -- PINNED by
                fixtures captured from the imported reference (tests/golden/).
  detection.py  torchvision-owned stages (transform, ResNet-50-FPN, RPN, RoIAlign,
                box/mask heads, NMS) restated from torchvision's published
                semantics.  torchvision is NOT installed in
                this image and its version is unpinned upstream: PARITY UNPINNED
                against the real library; pinned only by analytic known-answer tests.
  model.py      VideoMatchRCNN / MatchRCNN eval forward orchestration
                (models/video_matchrcnn.py:207-316, models/matchrcnn.py:333-472).
"""
