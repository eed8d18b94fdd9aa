#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_upper.py -x -q 2>&1 | tail -1
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['phase_ms_per_iter_1gpu'].items()}, {k[:20]: round(v,4) for k,v in d['kernel_ms'].items()})"
done
python scripts/sweep_chunk.py 1000000 0 2>&1 | tail -1
bash scripts/kernel_times.sh kb python scripts/profile_iter.py 1000000 2>&1 | grep -E "k_chunk_backward|k_fused"
