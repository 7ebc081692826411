#!/bin/bash
# The N > 1 path of bench.py rehearsed on a one-GPU box: two ranks share GPU 0 and exchange their gradient buckets over gloo
# (RCCL refuses two ranks on one device): spawner, shard ranges, global-batch loss denominators, GradBucket launch order, max-over-ranks timing.
mkdir -p gpurun_out/run
SR_FORCE_DEVICE=0 SR_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --global-batch 1536 --steps 4 --warmup 2 --no-roofline > gpurun_out/run/gloo2.json 2> gpurun_out/run/gloo2.err; echo "rc=$?"
tail -3 gpurun_out/run/gloo2.err; cut -c1-300 gpurun_out/run/gloo2.json
# the same with the gradient buckets going through the C ABI's communicator (bench.py --comm abi).  RCCL refuses two ranks on one
# device, so on a one-GPU box this form runs at world size 1: communicator, bucket plumbing and the exchange timing are exercised, the
# collective itself is a no-op; the two-rank line above is torch.distributed's path (--comm torch).
timeout -k 10 300 python bench.py --gpus 1 --comm abi --global-batch 768 --steps 4 --warmup 2 --no-roofline --no-cpu-baseline > gpurun_out/run/abi1.json 2> gpurun_out/run/abi1.err; echo "rc=$?"
tail -2 gpurun_out/run/abi1.err; cut -c1-300 gpurun_out/run/abi1.json
