#!/bin/bash
# round 6 call 32: PMC of the halo-tile conv after the instruction diet (two counter sets, separate passes; tools/halo_pmc.py, tools/pmc_fold.py) - the
# counterpart of profiles/r05_halo_pmc_lean_epilogue.json
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/r06_halo_pmc/a -- python $R/tools/halo_pmc.py > $OUT/r06_halo_pmc_a.log 2>&1
echo "a rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/r06_halo_pmc/b -- python $R/tools/halo_pmc.py > $OUT/r06_halo_pmc_b.log 2>&1
echo "b rc=$?"
cd $R
python tools/pmc_fold.py $OUT/r06_halo_pmc $OUT/r06_halo_pmc.json > /dev/null 2>&1; echo "fold rc=$?"
find $OUT/r06_halo_pmc -type f -size +1M -delete 2>/dev/null
head -c 3500 $OUT/r06_halo_pmc.json
