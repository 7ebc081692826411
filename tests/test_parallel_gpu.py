"""A14 (reference sr.py:467-470, `nn.DataParallel`) on the GPU: the REAL FCGGNN -- HIP backbones, hand-written GGNN backward,
GGNN weights shared by the verb and noun paths, `_Shadow` weight copies -- under two data-parallel ranks with
`parallel.GradBucket` + `parallel.global_batch_loss`.

The ranks are two child processes (started with subprocess; never an exec of this GPU-initialised process) that share GPU 0 and
exchange gradients over gloo (RCCL refuses two ranks on one device; the collective itself is covered by the world-size-1 test of
the C ABI below and by tests/test_parallel_gloo.py).  Unequal shards (7 + 4 images) with unequal numbers of valid roles, two
steps.

  frozen_bn   backbones in eval mode (images independent): gradients, clipped norm, losses and post-Adamax parameters must equal
              (a) this process running the GLOBAL batch of 11 in one piece and (b) the CPU oracle doing the same -- the reference
              computes its loss means after DataParallel's gather (sr.py:67-81).
  train_bn    train-mode BatchNorm, statistics per rank exactly as under DataParallel (no SyncBN): must equal this process
              running the two shards one after the other with the global loss denominators and summing the gradients.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run_two_ranks(tmp_path, mode):
    port = _free_port()
    prefix = str(tmp_path / ("dp_" + mode))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SR_FORCE_DEVICE="0", SR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_gpu_worker.py"), prefix, mode], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out[-4000:])
    return [torch.load("%s.rank%d.pt" % (prefix, r), weights_only=True) for r in range(2)]


def _close(a, b, rel=2e-4):
    scale = max(1e-3, float(b.abs().max()))
    return float((a - b).abs().max()) <= rel * scale


@pytest.mark.parametrize("mode", ["frozen_bn", "train_bn"])
def test_two_rank_step_of_the_real_model(tmp_path, mode):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, HERE)
    import dp_gpu_worker as W
    r0, r1 = _run_two_ranks(tmp_path, mode)
    assert r0["nbuckets"] > 2
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)                                           # replicas stay identical
    for s in range(2):
        assert abs(r0["hist"][s]["grad_norm"] - r1["hist"][s]["grad_norm"]) < 1e-6 * max(1.0, r0["hist"][s]["grad_norm"])
        for a, b in zip(r0["hist"][s]["grads"], r1["hist"][s]["grads"]):
            assert torch.equal(a, b)                                       # the all-reduced gradient is the same tensor on both ranks

    # ---- single process (this one), same model / data
    net, enc, chw = W.build(mode)
    img, verb, nouns = (t.cuda() for t in W.data(enc, chw))
    params = [p for p in net.parameters() if p.requires_grad]
    assert [k for k, p in net.named_parameters() if p.requires_grad] == r0["names"]
    opt = torch.optim.Adamax(params, lr=0.002)
    L = enc.get_num_labels()
    for s in range(2):
        opt.zero_grad(set_to_none=True)
        if mode == "frozen_bn":                                            # the global batch in one piece, the reference's own means
            pv, pn, pg = net(img, verb)
            vl, nl = net.verb_loss(pv, verb), net.nouns_loss(pn, nouns)
            (vl + nl).backward()
            got_pv = torch.cat([r0["hist"][s]["pred_verb"], r1["hist"][s]["pred_verb"]])
            assert _close(got_pv, pv.detach().cpu(), 1e-4)
        else:                                                              # per-replica BatchNorm: shard by shard, global denominators
            denom_b = torch.tensor(float(W.SPLIT[-1]), device="cuda")
            denoms = (nouns != L).sum(dim=(0, 2)).float()
            vl = nl = 0.0
            for lo, hi in zip(W.SPLIT[:-1], W.SPLIT[1:]):
                pv, pn, pg = net(img[lo:hi], verb[lo:hi])
                v = net.verb_loss(pv, verb[lo:hi], denom=denom_b)
                n = net.nouns_loss(pn, nouns[lo:hi], denoms=denoms)
                (v + n).backward()
                vl, nl = vl + v.detach(), nl + n.detach()
        h = r0["hist"][s]
        assert abs(h["losses"][0] - float(vl)) < 2e-5 * max(1.0, float(vl)) and abs(h["losses"][1] - float(nl)) < 2e-5 * max(1.0, float(nl))
        for k, p, g in zip(r0["names"], params, h["grads"]):
            assert _close(g, p.grad.detach().cpu()), (mode, s, k, float((g - p.grad.cpu()).abs().max()), float(p.grad.abs().max()))
        gn = torch.nn.utils.clip_grad_norm_(params, 1.0)
        assert abs(float(gn) - h["grad_norm"]) < 2e-4 * float(gn)
        opt.step()
    for k, p, q in zip(r0["names"], params, r0["params"]):
        assert float((p.detach().cpu() - q).abs().max()) < 1e-5 * max(1.0, float(q.abs().max())), k

    if mode != "frozen_bn":
        return
    # ---- the CPU oracle's global-batch step (reference semantics: loss means over the gathered batch, sr.py:67-81)
    from golden_util import load, oracle_fcggnn
    ora, oenc, _ = oracle_fcggnn(load("g3_fcggnn_bottleneck.npz"))
    ora.train()
    ora.convnet_verbs.eval(); ora.convnet_nouns.eval()
    ora.verb_classifier[0].p = 0.0
    ora.nouns_classifier[0].p = 0.0
    oimg, overb, onouns = W.data(enc, chw)
    oparams = [p for p in ora.parameters() if p.requires_grad]
    oopt = torch.optim.Adamax(oparams, lr=0.002)
    names = [k for k, p in ora.named_parameters() if p.requires_grad]
    assert names == r0["names"]
    for s in range(2):
        oopt.zero_grad()
        pv, pn, pg = ora(oimg, overb)
        vl, nl = ora.verb_loss(pv, overb), ora.nouns_loss(pn, onouns)
        (vl + nl).backward()
        h = r0["hist"][s]
        assert abs(h["losses"][0] - float(vl)) < 1e-3 and abs(h["losses"][1] - float(nl)) < 1e-3
        for k, p, g in zip(names, oparams, h["grads"]):
            ref = p.grad if p.grad is not None else torch.zeros_like(p)
            assert float((g - ref).abs().max()) <= 1e-3 * max(1e-2, float(ref.abs().max())), (s, k)
        gn = torch.nn.utils.clip_grad_norm_(oparams, 1.0)
        assert abs(float(gn) - h["grad_norm"]) < 1e-3 * float(gn)
        oopt.step()
    for k, p, q in zip(names, oparams, r0["params"]):
        assert float((p.detach() - q).abs().max()) <= 1e-3 * max(1.0, float(q.abs().max())), k


def test_c_abi_allreduce_world_size_one():
    """`sr_comm_unique_id` / `sr_comm_init` / `sr_allreduce_sum` / `sr_comm_destroy` (include/srhip.h): RCCL bound through
    the C ABI.  One rank is all a one-GPU box allows (RCCL refuses two ranks on one device): the communicator must come up, report
    world 1, sum fp32 and bf16 buffers in place on a side stream (identity at world 1) and drive GradBucket's `comm=` path."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from situation_recognition_amd import _lib, parallel
    comm = parallel.HipComm(rank=0, world=1)
    assert _lib.lib().sr_comm_world(comm._h) == 1
    for dt in (torch.float32, torch.bfloat16):
        t = torch.randn(1 << 20, device="cuda").to(dt)
        keep = t.clone()
        ev = comm.all_reduce_sum_(t)
        torch.cuda.current_stream().wait_event(ev)
        torch.cuda.synchronize()
        assert torch.equal(t, keep)
    lin = torch.nn.Linear(64, 64).cuda()
    bucket = parallel.GradBucket(list(lin.parameters()), comm=comm, min_bucket_bytes=64)
    bucket.zero()
    lin(torch.randn(8, 64, device="cuda")).sum().backward()
    bucket.finish()
    assert all(bucket._launched) and float(lin.weight.grad.abs().max()) > 0
    with pytest.raises(RuntimeError):                      # a second backward without zero(): its gradients would miss the all-reduce
        lin(torch.randn(8, 64, device="cuda")).sum().backward()
    bucket.close()
    comm.close()
    assert _lib.lib().sr_allreduce_sum(None, None, 0, 0, None) == -1
