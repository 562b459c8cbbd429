"""Host-side mirror of the reference's ``models/`` package (same module, class and
function names, signatures and state-dict keys) with the arithmetic dispatched to the
gfx950 kernels behind ``include/seam_hip.h``.

  nlb.py             <-> reference models/nlb.py
  match_head.py      <-> reference models/match_head.py   (heads only; losses are callers)
  video_matchrcnn.py <-> reference models/video_matchrcnn.py
  matchrcnn.py       <-> reference models/matchrcnn.py
  detection.py       the torchvision-owned stages the reference only configures
"""
