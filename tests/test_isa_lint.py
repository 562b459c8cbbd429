"""ISA lint of the whole library (no GPU needed: hipcc cross-compiles gfx950): no wide VMEM store with an SGPR offset may be
followed at once by a vector-ALU write to its data registers -- hipcc 7.2 omits the wait state in that form, and the store then
writes the NEXT value (seam_pwpc.hip round 5: wrong fourth channels; seam_pwh.hip round 6: NaNs; eight latent instances found in
seam_pw.hip by this lint).  tools/isa_store_hazard.py documents the rule."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_store_data_hazard_in_the_library_isa():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_store_hazard.py")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert r.stdout.count("wide stores with an SGPR offset checked") >= 15
