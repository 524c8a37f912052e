#!/bin/bash
O=gpurun_out/p_phase; mkdir -p $O
for v in 0 1 0 1; do for c in 4 5 0; do echo -n "vlate $v: "; PKV_LIB_PATH=probes/bin/libpkv_vlate_$v.so PKV_PREFILL_CASE=$c timeout -k 10 120 python probes/bench_prefill.py 2>&1 | grep n_seq; done; done | tee $O/vlate.txt
