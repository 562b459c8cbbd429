export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 python -m pytest tests/test_gpu_wino.py -q -x 2>&1 | tail -5
echo "######## NT=2 persistent"; SEAM_W24_NT=2 timeout 300 python tools/w24_ab.py 2>/dev/null | tail -n +3
echo "######## NT=2 one-shot"; SEAM_W24_PERSIST=0 SEAM_W24_NT=2 timeout 300 python tools/w24_ab.py 2>/dev/null | tail -n +3
echo "######## NT=1"; SEAM_W24_NT=1 timeout 300 python tools/w24_ab.py 2>/dev/null | tail -n +3
