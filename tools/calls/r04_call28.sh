#!/bin/bash
# attention PMC refresh (two --pmc passes, kernel-trace only)
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/r04_attn_pmc/a -- python $R/tools/attn_pmc.py > $OUT/r04_attn_pmc_a.log 2>&1
echo "pass a rc=$?"
timeout 150 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/r04_attn_pmc/b -- python $R/tools/attn_pmc.py > $OUT/r04_attn_pmc_b.log 2>&1
echo "pass b rc=$?"
cd $R
python tools/attn_pmc.py --summarise $OUT/r04_attn_pmc > $OUT/r04_attention_pmc.json 2> $OUT/r04_attn_pmc_sum.err
find $OUT/r04_attn_pmc -type f -size +1M -delete
head -c 1500 $OUT/r04_attention_pmc.json
