set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_fp8_gpu.py -m gpu -q -s --timeout 300 > gpurun_out/r2_tests_fp8.log 2>&1; rc=$?
echo "pytest rc=$rc"; grep -E "fp8 backbone|passed|failed" gpurun_out/r2_tests_fp8.log
cat > /tmp/f8bench.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
from situation_recognition_amd import ops
from bench_layers import timeit, line
F8 = torch.float8_e4m3fn
for (B, H, C) in ((6144, 14, 256), (6144, 28, 128), (6144, 7, 512)):
    M = B * H * H
    x = torch.randn(B, H, H, C, device='cuda').relu_()
    xq = (x * 16).clamp(max=448).to(F8).view(torch.uint8)
    xb = x.to(torch.bfloat16)
    w = torch.randn(C, 9 * C, device='cuda') * (9 * C) ** -0.5
    wq = (w * 100).clamp(-448, 448).to(F8).view(torch.uint8)
    wb = w.to(torch.bfloat16)
    dq = torch.full((C,), 1 / 1600.0, device='cuda')
    fl = 2.0 * M * C * 9 * C
    line("fp8  3x3 %d->%d @%d +stats" % (C, C, H), timeit(lambda: ops.conv3x3_fp8(xq, wq, dq, C, want_stats=True)), fl, M * C * 3)
    line("bf16 3x3 %d->%d @%d +stats" % (C, C, H), timeit(lambda: ops.conv2d(xb, wb, C, 3, 1, 1, want_stats=True)), fl, M * C * 4)
    sc, sh = torch.rand(C, device='cuda'), torch.rand(C, device='cuda')
    line("bn_apply -> fp8 C=%d" % C, timeit(lambda: ops.quantize_fp8(xb, 16.0, sc, sh, relu=True)), 0, M * C * 3)
PY
timeout -k 10 200 python /tmp/f8bench.py 2>&1 | grep -v amdgpu
for f in "" "--fp8"; do
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --T 8 --global-batch 8192 --no-cpu-baseline --no-roofline $f > gpurun_out/r2_bench_c5$f.json 2> gpurun_out/r2_bench_c5$f.err; echo "config5 $f $(python3 -c "
import json
d=json.load(open('gpurun_out/r2_bench_c5$f.json')); print(d['value'], d['ms_per_step'], d['config']['final_loss'])")"
done
