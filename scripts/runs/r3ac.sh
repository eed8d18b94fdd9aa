#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_segmented.py -x -q -k "register_resident" 2>&1 | tail -5
# the driver's multi-GPU launch line at world size 1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -2 | cut -c1-600
