#!/bin/bash
# round 6 call 6: PMC of the streaming / ping-pong attention kernels, K-group GEMM tests, in-program knob A/B
R=$PWD
OUT=$R/gpurun_out
export LB_SYNTH_CACHE=/tmp TMPDIR=/tmp
mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "glds_variant or layernorm_fused or 192x128" > $OUT/r06_gemm_tests6.txt 2>&1
echo "tests rc=$?"; tail -3 $OUT/r06_gemm_tests6.txt
cd /tmp
export LB_ATTN_SELF_ONLY=1
for F in 0 513 514; do
  export LB_ATTN_FORCE=$F
  timeout 150 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/r06_attn_pmc/f$F/a -- python $R/tools/attn_pmc.py > $OUT/r06_attn_pmc_a$F.log 2>&1
  echo "pass a force=$F rc=$?"
  timeout 150 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/r06_attn_pmc/f$F/b -- python $R/tools/attn_pmc.py > $OUT/r06_attn_pmc_b$F.log 2>&1
  echo "pass b force=$F rc=$?"
done
unset LB_ATTN_FORCE LB_ATTN_SELF_ONLY
cd $R
for F in 0 513 514; do python tools/attn_pmc.py --summarise $OUT/r06_attn_pmc/f$F > $OUT/r06_attention_pmc_f$F.json 2> /dev/null; done
find $OUT/r06_attn_pmc -type f -size +1M -delete
head -c 2500 $OUT/r06_attention_pmc_f0.json
timeout 900 python tools/unet_knob_ab.py > $OUT/r06_unet_knob_ab6.txt 2>&1
echo "knob rc=$?"; grep -E "^B=" $OUT/r06_unet_knob_ab6.txt
