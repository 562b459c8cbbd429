"""seam-match-rcnn_amd: MI355X-native (gfx950) forward hot path of SEAM Match-RCNN.

Layout (only what the path needs):
  csrc/      hand-written HIP kernels + the C ABI (``include/seam_hip.h``)
  _native.py ctypes loader for ``lib/libseam_hip.so`` (fails loudly when missing)
  ops.py     thin launch wrappers (raw device pointers + current HIP stream)
  models/    host-side mirror of the reference's ``models/`` interface
  retrieval.py  clip sharding + RCCL all-gather of the product bank + match
  synth.py   deterministic synthetic weights / inputs
"""
__version__ = "0.1.0"
