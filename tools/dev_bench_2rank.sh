#!/bin/bash
# rehearsal: bench.py under torch.distributed.run with 2 ranks on the ONE GPU of the test box (RCCL refuses the duplicate
# device, so this exercises the launcher plumbing and the relay fallback)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
SBO_BENCH_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 3 --warmup 1
