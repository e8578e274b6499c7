#!/bin/bash
# round-2 job 3: new parity tests (instances + full size), then the whole GPU suite
set -u
O=gpurun_out/r02_job3; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_instances.py tests/test_gpu_fullsize.py -m gpu -x -q --durations=15 2>&1 | tail -40 | tee $O/pytest_new.txt
GPMPC_PAIR_TB=2 python bench.py --config C4 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4 TB=2', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))" | tee $O/c4_tb2.txt
python bench.py --config C4 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4 TB=1', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))" | tee -a $O/c4_tb2.txt
