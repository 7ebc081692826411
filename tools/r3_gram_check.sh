set -e
python -m pytest tests/test_kernels_gpu.py tests/test_production_shapes_gpu.py tests/test_race_screen_gpu.py "tests/test_full_configs_gpu.py::test_gram_statistics_route_matches_statistics_only_launch_in_the_backbone" tests/test_model_gpu.py -q -m gpu -x > gpurun_out/r3_t3.log 2>&1 || { tail -40 gpurun_out/r3_t3.log; exit 1; }
tail -3 gpurun_out/r3_t3.log
python tools/race_screen.py 40 > gpurun_out/r3_race.txt 2>&1 || { cat gpurun_out/r3_race.txt; exit 1; }
tail -4 gpurun_out/r3_race.txt
python tools/pmc_r3.py all 2>/dev/null | tee gpurun_out/r3_gram_times.txt
python tools/bench_gram.py 2>/dev/null | tee gpurun_out/r3_bench_gram.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/r3_bench_a.json; python -c "
import json; d=json.load(open('gpurun_out/r3_bench_a.json')); print(d['ms_per_step'], d['value']); 
for e in d['roofline']['by_kernel']: print(e['kernel'][:50], e['ms_per_step'], e['achieved'], e['unit'], e['frac'])"
