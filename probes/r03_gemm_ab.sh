#!/bin/bash
# batched products: second form (gemm2_kernel) against the first (PKV_GEMM_V1=1), same process image, same shapes
out=gpurun_out/r03/gemm_ab.txt
echo "== second form (default)" > $out
python3 probes/gemm_bench.py >> $out 2>&1
echo "== first form (PKV_GEMM_V1=1)" >> $out
PKV_GEMM_V1=1 python3 probes/gemm_bench.py >> $out 2>&1
