#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4_t; mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_encoder_fused.py -q -k "fp32_autograd" -s > $OUT/dw.log 2>&1; echo "dw rc=$?"; grep -E "worst|passed|failed" $OUT/dw.log | cut -c1-250
TABGNN_NO_DW_FFN=1 timeout -k 10 600 python -m pytest tests/test_gpu_encoder_fused.py -q -k "fp32_autograd" -s > $OUT/nodw.log 2>&1; echo "nodw rc=$?"; grep -E "worst|passed|failed" $OUT/nodw.log | cut -c1-250
