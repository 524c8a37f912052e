cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_decode.py tests/test_gpu_decode_step.py -m gpu -x -q 2>&1 | tail -2
for kv in q8_0 q4_0; do
python3 bench.py --no-cpu-baseline --n-seq 32 --kv-type $kv > gpurun_out/b_$kv.json
python3 -c "import json; d=json.load(open('gpurun_out/b_$kv.json')); print('$kv', d['value'], 'tok/s', d['roofline']['kernel_us']['mean'], 'us', d['roofline']['frac'], d['roofline']['algorithmic_bytes_per_launch'])"
done
python3 bench.py --no-cpu-baseline --kv-type q8_0 > gpurun_out/b_q8_1.json
python3 -c "import json; d=json.load(open('gpurun_out/b_q8_1.json')); print('q8_0 n1', d['value'], 'tok/s', d['roofline']['kernel_us']['mean'], 'us')"
