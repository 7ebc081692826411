mkdir -p gpurun_out/r4
timeout -k 10 400 python -m pytest tests/test_model_gpu.py tests/test_packed_roles_gpu.py tests/test_parallel_gpu.py -x -q 2>&1 | tail -3
for v in 0 1 0 1; do SR_OVERLAP_PRED=$v timeout -k 10 200 python bench.py --global-batch 768 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pred-branch on side stream=$v', d['ms_per_step'], d['config']['final_loss'])"; done
for v in 0 1; do SR_OVERLAP_PRED=$v timeout -k 10 200 python bench.py --global-batch 1536 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b1536 pred-branch on side stream=$v', d['ms_per_step'], d['config']['final_loss'])"; done
