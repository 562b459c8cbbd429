export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
SH="80,200,200,256,256,1 2560,14,14,256,256,1 80,100,100,256,256,1 80,50,50,256,256,1 80,25,25,512,512,1 2560,14,14,256,256,0"
for nt in 1 2; do for ns in 1 2 4; do
  echo "######## NT=$nt NSPLIT=$ns"
  SEAM_W24_NT=$nt SEAM_W24_NSPLIT=$ns python tools/w24_ab.py $SH 2>/dev/null | tail -n +3
done; done > $O/r02f_w24_nsplit.txt 2>&1
cat $O/r02f_w24_nsplit.txt
