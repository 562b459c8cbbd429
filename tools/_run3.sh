export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
SEAM_W24_NT=1 python tools/w24_ab.py > $O/r02d_w24_nt1.txt 2>&1
SEAM_W24_NT=2 python tools/w24_ab.py > $O/r02d_w24_nt2.txt 2>&1
python tools/w24_ab.py > $O/r02d_w24_auto.txt 2>&1
python -m pytest tests/test_gpu_wino.py -q 2>&1 | tail -3
cat $O/r02d_w24_nt1.txt; cat $O/r02d_w24_nt2.txt; tail -1 $O/r02d_w24_auto.txt
