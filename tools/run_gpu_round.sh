./tools/ubench/mfma_f8_layout
SR_FORCE_DEVICE=0 SR_DIST_BACKEND=gloo timeout -k 10 200 python bench.py --gpus 2 --global-batch 512 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_gloo2.json 2> gpurun_out/r2_bench_gloo2.err; echo "gloo2 rc=$?"; python3 -c "
import json
d=json.load(open('gpurun_out/r2_bench_gloo2.json')); print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['final_loss'], d['roofline']['frac'])"
