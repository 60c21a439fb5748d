export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "split3" 2>&1 | tail -5
timeout 300 python bench.py --arith fp32_split3 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/bench_split3.log 2>&1
grep '^{' gpurun_out/bench_split3.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('stage_ms'), d.get('parity_check'))"
tail -3 gpurun_out/bench_split3.log | cut -c1-400
