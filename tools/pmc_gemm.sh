#!/bin/bash
# PMC passes over single pointwise-GEMM shapes (GPU box): tools/pmc_gemm.sh "<shape idx> <knob>" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2/pmc_gemm; mkdir -p $O
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS -d $O/a_$tag -o p --output-format csv -- python $R/tools/gpu_tune.py one $cfg > $O/a_$tag.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $O/b_$tag -o p --output-format csv -- python $R/tools/gpu_tune.py one $cfg > $O/b_$tag.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS SQ_LDS_MEM_VIOLATIONS -d $O/c_$tag -o p --output-format csv -- python $R/tools/gpu_tune.py one $cfg > $O/c_$tag.log 2>&1 || echo "pass c failed (counter names?)"
  (echo "== $cfg"; python $R/tools/pmc_raw.py $O/a_$tag pw_gemm; python $R/tools/pmc_raw.py $O/b_$tag pw_gemm; python $R/tools/pmc_raw.py $O/c_$tag pw_gemm) > $O/$tag.txt 2>&1
  rm -rf $O/a_$tag $O/b_$tag $O/c_$tag
  cat $O/$tag.txt
done
