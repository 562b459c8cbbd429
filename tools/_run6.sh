export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for nt in 1 2; do
  echo "######## NT=$nt scalar-lane VALU (new)"; SEAM_W24_NT=$nt python tools/w24_ab.py 2>/dev/null | tail -n +3
  echo "######## NT=$nt packed VALU (old)"; SEAM_LIB_PATH=$R/tools/experiments/_lib/libseam_pk.so SEAM_W24_NT=$nt python tools/w24_ab.py 2>/dev/null | tail -n +3
done > $O/r02g_w24_scalar_valu.txt 2>&1
cat $O/r02g_w24_scalar_valu.txt
