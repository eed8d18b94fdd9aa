#!/bin/bash
# parity subset on the new library, then A/B ld1 vs fd1
mkdir -p gpurun_out/r6j
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows_kernel.py tests/test_gpu_pending.py tests/test_gpu_irows.py tests/test_gpu_vw.py tests/test_gpu_configs.py tests/test_gpu_projection.py -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r6j/parity_subset.txt
for rep in 1 2; do
  for L in lib_ld1.so lib_fd1.so; do
    echo "$L bench: $(GPSLAM_LIB=$PWD/build_ab/$L python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("ms_per_step", round(d["ms_per_step"],4), "l0_in_iter_ms", round(d["roofline"]["avg_launch_ms"],4), "frac", round(d["roofline"]["frac"],3))')"
    echo "$L 1e6: $(GPSLAM_LIB=$PWD/build_ab/$L python scripts/sweep_chunk.py 1000000 0 2>&1 | tail -1 | tr '\n' ' ')"
    echo "$L c5b: $(GPSLAM_LIB=$PWD/build_ab/$L python scripts/profile_cfg.py c5b 1000000 2>&1 | tail -2 | tr '\n' ' ')"
  done
done 2>&1 | tee gpurun_out/r6j/ab_fd.txt
