cd $GRAFT_REPO_ROOT
for mode in "" "--two-call"; do
timeout -k 10 300 python bench.py --no-cpu-baseline --n-seq 32 $mode | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('n32 $mode', d['value'], d['roofline']['frac'], d['roofline']['kernel_us'])"
done
