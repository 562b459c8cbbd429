"""Import shim: the product package lives in the directory ``seam-match-rcnn_amd/``
(a hyphen is not importable), so this module turns itself into that package:
``import seam_match_rcnn_amd`` / ``from seam_match_rcnn_amd.models import ...`` resolve
into ``seam-match-rcnn_amd/``."""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "seam-match-rcnn_amd")
__path__ = [_PKG_DIR]
__file__ = _os.path.join(_PKG_DIR, "__init__.py")
with open(__file__, "r", encoding="utf-8") as _f:
    exec(compile(_f.read(), __file__, "exec"))
