#!/bin/bash
# round 3, robustness: GPU suite repeated + open/close determinism loop (now through the device k-means), and bench.py --gpus 2 on a
# one-GPU box (both ranks on device 0): the real engine through the launcher, file barriers, weak and strong scaling
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
bash scripts/stress_gpu.sh > $O/stress.log 2>&1; cat $O/stress.log | cut -c1-300
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu --no-secondary > $O/bench_2ranks_1gpu_weak.json 2> $O/bench_2ranks_weak.err; cut -c1-400 $O/bench_2ranks_1gpu_weak.json
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu --no-secondary --scaling strong > $O/bench_2ranks_1gpu_strong.json 2> $O/bench_2ranks_strong.err; cut -c1-400 $O/bench_2ranks_1gpu_strong.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu --no-secondary > $O/bench_torchrun_2ranks.json 2> $O/bench_torchrun.err; cut -c1-300 $O/bench_torchrun_2ranks.json; tail -3 $O/bench_torchrun.err
