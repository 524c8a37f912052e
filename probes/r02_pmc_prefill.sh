#!/bin/bash
# PMC passes over the prefill kernel (one case per run: PKV_PREFILL_CASE), counters in their own runs (no sys/hip trace domains)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-pmc_prefill}; rm -rf $O; mkdir -p $O
for c in ${CASES:-4 0}; do
  export PKV_PREFILL_CASE=$c
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
             "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/c${c}_p$i -- python3 probes/bench_prefill.py > $O/c${c}_p$i.log 2>&1 || { echo "pass $i case $c failed"; tail -5 $O/c${c}_p$i.log; }
  done
done
find $O -name "*_agent_info.csv" -delete
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("O_DIR") or sorted(glob.glob("gpurun_out/*pmc*"))[-1]
for c in sorted({os.path.basename(p).split("_")[0] for p in glob.glob(O + "/c*_p*") if os.path.isdir(p)}):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/{c}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_prefill_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(c, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
