#!/bin/bash
# batched products: third / second form (default) against the second form alone (PKV_GEMM_NO_DEEP=1) and the first (PKV_GEMM_V1=1), same shapes
out=gpurun_out/r03/gemm_ab.txt
echo "== default (third form for f16 / packed q4_K below 256 x 256 tiles, second form otherwise)" > $out
python3 probes/gemm_bench.py >> $out 2>&1
echo "== second form only (PKV_GEMM_NO_DEEP=1)" >> $out
PKV_GEMM_NO_DEEP=1 python3 probes/gemm_bench.py >> $out 2>&1
echo "== first form (PKV_GEMM_V1=1)" >> $out
PKV_GEMM_V1=1 python3 probes/gemm_bench.py >> $out 2>&1
