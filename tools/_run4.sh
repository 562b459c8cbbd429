export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for nt in 1 2; do
  echo "######## NT=$nt"
  SEAM_W24_NT=$nt bash tools/experiments/wino24_abl.sh run 80,200,200,256,256,1 2560,14,14,256,256,1 80,50,50,256,256,1
done > $O/r02e_w24_abl.txt 2>&1
cat $O/r02e_w24_abl.txt
