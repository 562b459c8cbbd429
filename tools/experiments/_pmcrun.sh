S=80,200,200,256,256,1
C1="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
L=$PWD/tools/experiments/_lib
bash tools/pmc_w24.sh $S $C1
for f in $(ls $L/libseam_*.so | sort -V); do SEAM_LIB_PATH=$f bash tools/pmc_w24.sh $S $C1; done
