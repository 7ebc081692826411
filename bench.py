#!/usr/bin/env python3
"""Headline benchmark: training images/sec (fwd+bwd) of ResNet-152 + 6-role GGNN (T=5) at GLOBAL batch 6144,
bf16, on N MI355X (BASELINE.json metric; config.workload names the configuration).

One "step" = zero_grad -> FCGGNN.forward(img, gt_verb) -> verb_loss + nouns_loss -> backward -> gradient
all-reduce (N > 1) -> clip_grad_norm_(1) -> Adamax step  (reference order: sr.py:63-83), on synthetic inputs
already resident in HBM.  Scorer / data loading / .item() logging are outside the step.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: one rank per GPU -- under torch.distributed.run when
                                                                  launched by it, otherwise bench.py starts the ranks itself)

Prints ONE JSON line on rank 0, including
  roofline     -- the dominant kernel family (backbone implicit-GEMM convolution): algorithmic FLOPs per launch / average
                  launch duration, measured with HIP events on the launch stream, against the dense bf16 MFMA peak, plus
                  `by_kernel`: every kernel family of the step on the roofline that bounds it;
  cpu_baseline -- the CPU oracle (oracle/, "port") timed on the host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0                    # HBM3E spec (MI355X_MICROARCH.md: 8 TB/s peak, ~6.3 achievable)
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_FP8_TFLOPS = 5000.0           # dense fp8 (block-scaled MFMA, K = 128)
RESNET_GFLOP = {18: 3.627, 50: 8.174, 152: 23.023}     # per image per pass at 224x224


def synthetic_batch(enc, B, res, device, seed_shift=0):
    """img ~ N(0,1) clamped to the ImageNet-normalised range, uniform verbs, labels uniform in
    [0,L) on real roles and L (= ignore) on padded roles."""
    g = torch.Generator(device="cpu").manual_seed(1234 + seed_shift)
    V, L, R = enc.get_num_verbs(), enc.get_num_labels(), enc.get_max_role_count()
    verb = torch.randint(0, V, (B,), generator=g)
    nouns = torch.randint(0, L, (B, 3, R), generator=g)
    counts = enc.role_counts[verb]
    nouns[(torch.arange(R)[None, :] >= counts[:, None])[:, None, :].expand(B, 3, R)] = L
    gd = torch.Generator(device=device).manual_seed(1234 + seed_shift)
    img = torch.empty((B, 3, res, res), device=device, dtype=torch.float32)
    chunk = 512
    for i in range(0, B, chunk):
        img[i:i + chunk] = torch.randn((min(chunk, B - i), 3, res, res), device=device, generator=gd).clamp_(-2.2, 2.7)
    return img, verb.to(device), nouns.to(device)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def host_cores():
    """CPU cores this process may actually use: the cgroup quota / affinity mask, not the machine's core count
    (the GPU box exposes 256 logical CPUs but grants a 1-GPU job a share of them)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return int(os.environ.get("SR_CPU_THREADS", min(n, 64)))


def cpu_baseline(args):
    """The oracle's training step (fp32, CPU) on a bounded sample of the same workload."""
    from oracle.ref_encoder import SyntheticEncoder
    from oracle.ref_model import RefBackbone, RefFCGGNN, train_step
    torch.manual_seed(1238)
    cores = host_cores()
    torch.set_num_threads(cores)
    enc = SyntheticEncoder()
    D = 2048 if args.backbone >= 50 else 512
    net = RefFCGGNN(enc, D, steps=args.T, backbone_factory=lambda: RefBackbone(args.backbone))
    net.train()
    opt = torch.optim.Adamax([p for p in net.parameters() if p.requires_grad], lr=0.002)
    def sample(B, steps, warm):
        g = torch.Generator().manual_seed(7)
        img = torch.randn(B, 3, args.res, args.res, generator=g).clamp_(-2.2, 2.7)
        verb = torch.randint(0, enc.get_num_verbs(), (B,), generator=g)
        nouns = torch.randint(0, enc.get_num_labels(), (B, 3, enc.get_max_role_count()), generator=g)
        for _ in range(warm):
            log("cpu_baseline: warm-up step (batch %d, %d threads)" % (B, cores))
            train_step(net, opt, img, verb, nouns)
        t0 = time.perf_counter()
        for i in range(steps):
            train_step(net, opt, img, verb, nouns)
            log("cpu_baseline: batch %d, timed step %d/%d done" % (B, i + 1, steps))
        dt = time.perf_counter() - t0
        return B * steps / dt, dt

    B = args.cpu_batch
    steps = args.cpu_steps
    rate, dt = sample(B, steps, 1)
    out = {"value": round(rate, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "oracle (fp32 CPU restatement of reference model.py) full training step, ResNet-%d + 6-role GGNN T=%d, "
                     "batch %d, %d timed steps after 1 warm-up (%.1f s)" % (args.backbone, args.T, B, steps, dt)}
    if args.cpu_batch2 > 0:
        # the second size of SURVEY 8(d) / BASELINE.md (batch 256): ONE timed step after one warm-up step (the first step at a new
        # size pays for the allocator's growth and oneDNN's primitive creation: 58 s against 23 s on 16 cores)
        rate2, dt2 = sample(args.cpu_batch2, 1, 1)
        out["value_batch%d" % args.cpu_batch2] = round(rate2, 3)
        out["sample"] += "; value_batch%d: the same step at batch %d, 1 timed step after 1 warm-up (%.1f s)" % (args.cpu_batch2, args.cpu_batch2, dt2)
    else:
        out["sample"] += "; the second size of SURVEY 8(d), batch 256, is measured in the same run with --cpu-batch2 256 (tools/collect_profiles.sh)"
    return out


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves -- as a CHILD process, before
    this process has touched the GPU (never re-exec a process that initialised HIP) -- and relay rank 0's JSON line."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    log("spawning %d ranks: %s" % (args.gpus, " ".join(cmd)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in p.stdout:                      # ONE JSON line on stdout: anything else the ranks print (backend banners) goes to stderr
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return p.wait()


def profile_step(step_fn, ops, torch):
    """One extra training step with every libsrhip launch bracketed by HIP events on its launch stream.
    Returns {tag: (launches, seconds, algorithmic flops, algorithmic bytes)}."""
    ops.PROFILE = []
    step_fn()
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    agg = {}
    for tag, e0, e1, fl, by in prof:
        n, t, f, b = agg.get(tag, (0, 0.0, 0.0, 0.0))
        agg[tag] = (n + 1, t + e0.elapsed_time(e1) * 1e-3, f + fl, b + by)
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--global-batch", type=int, default=6144)
    ap.add_argument("--graphs", default="off", choices=["auto", "on", "off"],
                    help="replay the backbones' train-mode passes from hipGraphs (auto: per-GPU batch <= 1536); measured: no gain --\n"
                         "the gaps between a pass's launches are the GPU's dispatch latency, not the host's (82.3 vs 82.5 ms at batch 768)")
    ap.add_argument("--backbone", type=int, default=152)
    ap.add_argument("--T", type=int, default=5)
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--cpu-batch", type=int, default=64)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--cpu-batch2", type=int, default=0,
                    help="second CPU sample size (SURVEY 8d: 256; one timed step after one warm-up: 2 x 60 s on a 16-core box, so it is "
                         "off in the default run and on in tools/collect_profiles.sh, whose line is committed under profiles/)")
    ap.add_argument("--comm", default="torch", choices=["torch", "abi"],
                    help="gradient exchange of the N > 1 path: torch.distributed's all_reduce (RCCL under the nccl backend; the default) or "
                         "the C ABI's sr_allreduce_sum (parallel.HipComm: RCCL bound by libsrhip itself, SURVEY 8b's export)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--fp8", action="store_true",
                    help="BASELINE config 5: the bottlenecks' 3x3 convolutions on the 2x-rate fp8 (e4m3) MFMA, everything else bf16 "
                         "(quote it with --T 8 --global-batch 8192).  Not the headline configuration.")
    ap.add_argument("--shared-backbone", action="store_true",
                    help="both backbones start from the SAME weights (what the reference's two pretrained=True loads give, "
                         "model.py:16,100-101): FCGGNN then runs one train-mode pass for both.  Not the headline configuration.")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args, sys.argv[1:]))

    from situation_recognition_amd import ops, parallel
    from situation_recognition_amd.imsitu_encoder import imsitu_encoder
    from situation_recognition_amd.model import FCGGNN

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if os.environ.get("SR_FORCE_DEVICE"):        # rehearsal of the N>1 path on a one-GPU box (with SR_DIST_BACKEND=gloo)
        local = int(os.environ["SR_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ops.lib()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    enc = imsitu_encoder.synthetic()                              # V=504, 190 roles, L=2001, R=6
    torch.manual_seed(1238)                                       # identical replicas on every rank
    D = 2048 if args.backbone >= 50 else 512
    net = FCGGNN(enc, D, steps=args.T, backbone=args.backbone, dtype=dtype, fp8=args.fp8)
    if args.shared_backbone:
        net.convnet_nouns.load_state_dict(net.convnet_verbs.state_dict())
    net = net.to(dev)
    net.drop_seed_base += rank
    net.train()
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.Adamax(params, lr=0.002)
    comm = parallel.HipComm() if args.comm == "abi" else None     # (world 1: communicator and bucket plumbing run, nothing is exchanged)
    bucket = parallel.GradBucket(params, comm=comm) if (world > 1 or comm is not None) else None
    exch = []                                                     # per step: (event before, event after) bucket.finish() on the main stream

    lo, hi = parallel.shard_range(args.global_batch, rank, world)
    B = hi - lo
    # optional: the frozen backbones' train-mode passes replayed from captured hipGraphs (the first warm-up step runs eagerly and
    # captures).  Off by default: it removes the host's launch work, which is not what spaces the launches of a pass.
    use_graphs = args.graphs == "on" or (args.graphs == "auto" and B <= 1536 and not args.shared_backbone)
    if use_graphs:
        net.enable_graphs(True, train=True)
    img, verb, nouns = synthetic_batch(enc, B, args.res, dev, seed_shift=rank)

    def step():
        role_rows["last"] = []
        if bucket is None:
            opt.zero_grad(set_to_none=True)
        else:
            bucket.zero()
        pv, pn, pg = net(img, verb)
        if bucket is None:
            loss = net.verb_loss(pv, verb) + net.nouns_loss(pn, nouns)
            loss.backward()
        else:
            # loss means over the GLOBAL batch (sr.py:67-76 after DataParallel's gather): per-rank sums over all-reduced
            # denominators; the gradient buckets are then summed over the ranks, launched from autograd hooks during backward
            loss = parallel.global_batch_loss(net, pv, pn, verb, nouns)[0]
            loss.backward()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            bucket.finish()
            e1.record()
            exch.append((e0, e1))
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        return loss

    # packed role rows (FCGGNN._use_packed): how many rows the two noun branches of a step really push through the GGNN and the
    # classifier -- the predicted-verb branch of an untrained head collapses onto very few verbs, so its row count is whatever those
    # verbs' role counts are; printed in `config` so that two runs are comparable
    role_rows = {}
    plan0 = net._pack_plan

    def counting_plan(verbs, B_, R_):
        out_ = plan0(verbs, B_, R_)
        role_rows["last"] = role_rows.get("last", []) + [int(out_[1]) + 1]
        return out_

    net._pack_plan = counting_plan
    log("model + %d synthetic images per rank resident; warm-up" % B)
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log("warm-up step %d/%d done" % (i + 1, args.warmup))
    parallel.barrier()
    torch.cuda.synchronize()
    del exch[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = time.perf_counter() - t0
    final_loss = loss.detach().clone()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
        torch.distributed.all_reduce(final_loss)                  # shares of the global-batch loss -> the loss
    final_loss = float(final_loss)
    log("timed region done: %.1f ms/step" % (1000.0 * elapsed / args.steps))

    out = {
        "metric": "training images/sec (fwd+bwd) at batch 6144, 1/2/4/8 MI355X",
        "value": round(args.global_batch * args.steps / elapsed, 2),
        "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": args.dtype + ("+fp8(e4m3) 3x3 convolutions" if args.fp8 else ""), "data": "synthetic",
        "config": {"workload": "ResNet-%d backbone (x2, frozen, train-mode BN) + 6-role GGNN T=%d + verb/noun classifiers, "
                               "full training step, global batch %d, %dx%d synthetic images, imSitu-sized vocabulary "
                               "(504 verbs / 190 roles / 2001 labels)" % (args.backbone, args.T, args.global_batch, args.res, args.res),
                   "global_batch": args.global_batch, "per_gpu_batch": B, "parallelism": "dp%d" % world,
                   "backbone_weights": "shared (one pass serves both)" if args.shared_backbone else "two distinct backbones",
                   "backbone_launch": "hipGraph replay" if use_graphs else "eager",
                   "role_rows_executed": ({"predicted_verb_branch": role_rows["last"][0], "ground_truth_branch": role_rows["last"][1],
                                           "of_full_form": B * enc.get_max_role_count()} if len(role_rows.get("last", [])) == 2
                                          else {"predicted_verb_branch": B * enc.get_max_role_count(),
                                                "ground_truth_branch": B * enc.get_max_role_count(), "of_full_form": B * enc.get_max_role_count()}),
                   "final_loss": round(final_loss, 4)},
    }
    if bucket is not None:
        # what the gradient exchange costs the step on the main stream: HIP events around bucket.finish() -- the launch of the
        # buckets autograd's hooks had not sent yet plus the wait for all of them (the buckets launched during backward overlap it)
        ex_ms = [a.elapsed_time(b) for a, b in exch]
        out["config"]["gradient_exchange"] = {
            "via": "sr_allreduce_sum (C ABI, RCCL bound by libsrhip: parallel.HipComm)" if comm is not None else
                   "torch.distributed.all_reduce (%s backend)" % (torch.distributed.get_backend() if torch.distributed.is_initialized() else "none"),
            "buckets": len(bucket.buckets), "bytes": bucket.nbytes,
            "exposed_ms_per_step": round(sum(ex_ms) / max(len(ex_ms), 1), 3)}

    if not args.no_roofline:                  # (every rank runs the profiled step -- it contains the collectives -- rank 0 reports it)
        # One extra training step, every libsrhip launch bracketed by HIP events on its launch stream (single stream: the
        # two-stream overlap of small batches is switched off so that a launch's duration is its own).
        # Headline = the dominant kernel family (the backbone convolutions: gemm.hip, expand.hip, stem.hip) on the MFMA roofline,
        # as SURVEY 8(d) bounds it: achieved = sum of ALGORITHMIC FLOPs (2*M*N*K of each convolution once = 23.023 GFLOP per
        # image and pass for ResNet-152) / sum of the launches' durations -- statistics-only launches of the two-launch
        # BatchNorm scheme add their time but no FLOPs.  (The Gram-matrix statistics sweeps of the expansion convs are NOT convolution
        # launches: they are fused with the BatchNorm-apply sweep in front of them and reported as their own by_kernel entry.)  by_kernel: the same per kernel family, each on the roofline that
        # bounds it (3x3 and stem convolutions: MFMA; 1x1 convolutions: HBM; GGNN gate GEMMs: MFMA; the GGNN step's non-GEMM
        # kernels -- aggregate, GRU backward halves -- HBM, bytes as SURVEY 8(d) counts them).
        keep = net.overlap_backbones
        net.overlap_backbones = False
        agg = profile_step(step, ops, torch)
        net.overlap_backbones = keep
    if rank == 0 and not args.no_roofline:
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS

        def entry(tags, bound, kernel):
            n = sum(agg[t][0] for t in tags if t in agg)
            if n == 0:
                return None
            tsum = sum(agg[t][1] for t in tags if t in agg)
            fsum = sum(agg[t][2] for t in tags if t in agg)
            bsum = sum(agg[t][3] for t in tags if t in agg)
            e = {"kernel": kernel, "launches_per_step": n, "avg_launch_ms": round(1e3 * tsum / n, 4), "ms_per_step": round(1e3 * tsum, 2)}
            if bound == "mfma":
                pk = PEAK_FP8_TFLOPS if tags == ["conv3x3_fp8"] else peak
                e.update(bound="mfma", achieved=round(fsum / tsum / 1e12, 2), peak=pk, unit="TFLOP/s", frac=round(fsum / tsum / 1e12 / pk, 4),
                         alg_gflop_per_launch=round(fsum / n / 1e9, 3))
            else:
                e.update(bound="hbm", achieved=round(bsum / tsum / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(bsum / tsum / 1e9 / PEAK_HBM_GBS, 4),
                         alg_bytes_per_launch=round(bsum / n))
            return e

        conv_tags = ["conv1x1", "conv1x1_pair", "conv3x3", "conv7x7", "conv3x3_fp8"]
        head = entry(conv_tags, "mfma", "backbone convolutions, both passes, incl. statistics-only launches: conv_igemm_v3_kernel (reduce 1x1, downsample, stride-2 and layer4 3x3), conv3x3_c64_kernel / conv3x3_slices_kernel (layer1 / layer2 + layer3 stride-1 3x3, direct, BatchNorm of their input applied on load), conv1x1_pair_kernel (expansion 1x1 fused with the next block's reduce 1x1, layers 1-3), conv1x1_ws_kernel (the other output-heavy 1x1), stem_conv_kernel + stem_pool_kernel (7x7 stem)")
        hbm_view = entry(conv_tags, "hbm", "backbone convolutions")
        # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, run separately on this exact
        # command: profiles/conv_traffic.json records them with the gfx950 corrections); null for any other configuration
        traffic, traffic_source = None, None
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "conv_traffic.json")))
            c = t["config"]
            if (c["backbone"], c["per_gpu_batch"], c["dtype"], c["res"]) == (args.backbone, B, args.dtype, args.res) and not args.fp8:
                traffic = round(t["hbm_bytes_per_launch"])
                traffic_source = ("RECORDED constant, not measured in this run: profiles/conv_traffic.json = HBM bytes per convolution launch from "
                                  "separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this command (FETCH_SIZE x2 on gfx950), "
                                  "collected by tools/collect_profiles.sh; " + str(t.get("source", "")))
        except (OSError, KeyError, ValueError):
            pass
        by = [entry(["conv3x3"], "mfma", "3x3 convolutions: conv3x3_slices_kernel (layer3, layer2) + conv3x3_c64_kernel (layer1): direct, input BatchNorm on load; conv_igemm_v3_kernel (stride-2 layers, layer4)"),
              entry(["conv3x3_fp8"], "mfma", "conv3x3_fp8_kernel (e4m3 x e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4; peak = dense fp8)"), 
              entry(["conv1x1_pair"], "hbm", "conv1x1_pair_kernel: a bottleneck's expansion conv (+ BN + residual + ReLU) fused with the next block's reduce conv (+ statistics), the block output written once and not read back; 44 pairs per backbone pass"),
              entry(["conv1x1"], "hbm", "the other 1x1 convolutions: conv_igemm_v3_kernel (reduce convs of each layer's first block, downsample) + conv1x1_ws_kernel (expansion + BN + residual + ReLU of each layer's last block, layer4)"),
              entry(["conv7x7"], "mfma", "7x7 stem: stem_conv_kernel (statistics pass) + stem_pool_kernel (conv + BN + ReLU + max-pool); FLOPs counted once"),
              entry(["gram"], "hbm", "gram_kernel (Gram-matrix statistics of the expansion convs, fused with the preceding BN-apply)"),
              entry(["bn_apply"], "hbm", "bn_apply_kernel"), entry(["maxpool"], "hbm", "maxpool_kernel"),
              entry(["gemm_gate"], "mfma", "gemm_nt_v3_kernel with GRU gate epilogues (GGNN step: z, r & r*h, candidate & blend)"),
              entry(["gemm"], "mfma", "gemm_nt_v3_kernel linear (W_p, classifiers, backward data gradients)"),
              entry(["gemm_tn"], "mfma", "gram_kernel<256,true,true> TN weight gradients"),
              entry(["aggregate", "gru_bwd"], "hbm", "GGNN step non-GEMM kernels: aggregate_kernel, gru_bwd1/2_kernel (SURVEY 8d: 11*M*D*s per step "
                                                      "if the forward gates were standalone; they are fused into the gate GEMMs)")]
        out["roofline"] = dict(head, traffic=traffic, traffic_source=traffic_source, alg_bytes_per_launch=hbm_view["alg_bytes_per_launch"],
                               hbm_view={k: hbm_view[k] for k in ("achieved", "peak", "unit", "frac")},
                               by_kernel=[e for e in by if e is not None])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        parallel.barrier()                      # rank 0 may still be in its roofline leg
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
