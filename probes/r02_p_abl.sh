#!/bin/bash
# timing-only ablations of the prefill loop (probes/build_prefill_ablations.sh builds probes/bin/libpkv_abl_N.so); cases: pp4096 x 1, pp512 x 32
O=${O_ABL:-gpurun_out/p_abl}; mkdir -p $O
names=("0:complete kernel" "1:no tile barrier" "2:no softmax arithmetic" "4:no K/V prefetch, no LDS stores" "64:no K/V prefetch loads (stores kept)" "256:loads waited for, LDS stores skipped" "8:no QK^T MFMAs" "16:no PV MFMAs" "24:no MFMAs" "6:no softmax, no staging")
for e in "${names[@]}"; do
  ab=${e%%:*}; [ -f probes/bin/libpkv_abl_$ab.so ] || continue
  for c in 4 0; do
    echo -n "ablate $ab (${e#*:}): "; PKV_LIB_PATH=probes/bin/libpkv_abl_$ab.so PKV_PREFILL_CASE=$c timeout -k 10 120 python3 probes/bench_prefill.py 2>&1 | grep n_seq
  done
done | tee $O/ablate.txt
