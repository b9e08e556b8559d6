set -e
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "32 32 256 32 0" "96 32 256 32 64" "64 64 128 32 0"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d gpurun_out/r4/pmc_blk_$tag -o p --output-format csv -- python tools/gpu_block.py $cfg 5 1 0 0 0 > gpurun_out/r4/pmc_blk_$tag.log 2>&1
  rocprofv3 --kernel-trace --stats -d gpurun_out/r4/kt_blk_$tag -o k --output-format csv -- python tools/gpu_block.py $cfg 5 1 0 0 0 > gpurun_out/r4/kt_blk_$tag.log 2>&1
done
ls gpurun_out/r4/pmc_blk_*/ | head
