"""world_size=2 gloo (CPU) tests of the data-parallel step: shard ranges, the flat gradient bucket whose slices are the
parameters' .grad tensors, hook-launched bucketed all-reduce, and EXACT loss semantics -- an FCGGNN-shaped loss
(verb cross-entropy + three noun cross-entropies with `ignore_index`, reference model.py:183-201) on UNEQUAL shards with
unequal numbers of valid targets must give the gradient the single process gets on the whole batch (the reference computes
its loss means after DataParallel's gather, sr.py:67-81), the same clipped norm and the same parameters after Adamax."""
import os
import socket
import types

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

V, R, L, D, B = 7, 4, 9, 12, 11
SPLIT = (0, 7, 11)                  # rank 0: 7 samples, rank 1: 4


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _Head(torch.nn.Module):
    """CPU stand-in with FCGGNN's outputs and FCGGNN's OWN loss methods: [B,D] features -> verb logits [B,V] and noun logits
    [B,R,L] through a shared matrix (as the GGNN is shared by both paths), two embeddings-like tables and two classifiers.
    One parameter (`unused`) never receives a gradient, one module is frozen (the backbones)."""

    def __init__(self):
        super().__init__()
        from situation_recognition_amd.model import FCGGNN
        self.encoder = types.SimpleNamespace(get_num_labels=lambda: L)
        self.shared = torch.nn.Linear(D, D)
        self.role = torch.nn.Parameter(torch.randn(R, D))
        self.verb_classifier = torch.nn.Linear(D, V)
        self.nouns_classifier = torch.nn.Linear(D, L)
        self.unused = torch.nn.Parameter(torch.zeros(3))
        self.frozen = torch.nn.Linear(3, 3)
        for p in self.frozen.parameters():
            p.requires_grad = False
        self.verb_loss = types.MethodType(FCGGNN.verb_loss, self)
        self.nouns_loss = types.MethodType(FCGGNN.nouns_loss, self)

    def forward(self, x, nouns_first=False):
        """`nouns_first`: build the noun branch's graph before the verb branch's, as FCGGNN.forward does when it queues the
        ground-truth noun branch on the side stream (model.py `overlap_gt_branch`): autograd then completes the parameters'
        gradients in a different order."""
        h = torch.tanh(self.shared(x))
        if nouns_first:
            pn = self.nouns_classifier(torch.tanh(self.shared(h[:, None, :] * self.role[None])))
            pv = self.verb_classifier(torch.tanh(self.shared(x)))
            return pv, pn
        pv = self.verb_classifier(h)
        pn = self.nouns_classifier(torch.tanh(self.shared(h[:, None, :] * self.role[None])))
        return pv, pn


def _data():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, D, generator=g)
    verb = torch.randint(0, V, (B,), generator=g)
    nouns = torch.randint(0, L, (B, 3, R), generator=g)
    nroles = torch.tensor([4, 1, 2, 4, 3, 1, 1, 4, 4, 3, 4])            # rank 0 holds 16 valid role slots, rank 1 holds 15 of 16
    nouns[(torch.arange(R)[None, :] >= nroles[:, None])[:, None, :].expand(B, 3, R)] = L
    nouns[9, 1, 0] = L                                                   # annotators differ: per-annotator denominators
    return x, verb, nouns


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from situation_recognition_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    net = _Head()
    params = [p for p in net.parameters() if p.requires_grad]
    bucket = parallel.GradBucket(params, min_bucket_bytes=256)          # several buckets even at toy sizes
    assert bucket.flat.numel() == sum(p.numel() for p in params) and len(bucket.buckets) > 2
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
    x, verb, nouns = _data()
    lo, hi = SPLIT[rank], SPLIT[rank + 1]
    opt = torch.optim.Adamax(params, lr=0.01)
    hist = []
    for step in range(2):                                               # two steps: zero() must re-arm hooks and views
        bucket.zero()
        if step == 1:
            net.shared.weight.grad = None                               # a caller that reset a gradient: the hook adopts the new tensor
        # The ranks build their graphs in DIFFERENT orders (FCGGNN picks its packed-role / side-stream forms from the LOCAL batch,
        # and shards straddle the threshold: 1024 vs 1023 images): the collectives must still go out in one order on every rank.
        pv, pn = net(x[lo:hi], nouns_first=(rank == 1))
        loss, vl, nl, _ = parallel.global_batch_loss(net, pv, pn, verb[lo:hi], nouns[lo:hi])
        loss.backward()
        bucket.finish()
        assert all(bucket._launched) and float(net.unused.grad.abs().max()) == 0.0
        assert bucket.launch_order == list(range(len(bucket.buckets))), bucket.launch_order
        share = torch.stack([vl.detach(), nl.detach()])
        dist.all_reduce(share)
        gn = torch.nn.utils.clip_grad_norm_(params, 1.0)
        hist.append((float(gn), share.tolist(), [p.grad.clone() for p in params]))
        opt.step()
    parallel.barrier()
    out[rank] = (hist, [p.detach().clone() for p in params])
    dist.destroy_process_group()


def test_two_rank_step_equals_global_batch_step():
    from situation_recognition_amd import parallel
    assert [parallel.shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert parallel.shard_range(6144, 7, 8) == (5376, 6144)
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    (h0, p0), (h1, p1) = out[0], out[1]
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)                                        # replicas stay identical
    # single process on the whole batch with the reference's own loss (means over the gathered batch)
    torch.manual_seed(0)
    net = _Head()
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.Adamax(params, lr=0.01)
    x, verb, nouns = _data()
    for step in range(2):
        opt.zero_grad()
        pv, pn = net(x)
        vl, nl = net.verb_loss(pv, verb), net.nouns_loss(pn, nouns)
        (vl + nl).backward()
        grads = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in params]
        gn = torch.nn.utils.clip_grad_norm_(params, 1.0)
        g0, share, dp_grads = h0[step]
        assert abs(g0 - h1[step][0]) < 1e-7
        assert abs(float(gn) - g0) < 1e-5 * float(gn)
        assert abs(share[0] - float(vl)) < 1e-5 and abs(share[1] - float(nl)) < 1e-5
        for a, b in zip(grads, dp_grads):                               # (dp_grads were cloned after clipping: compare clipped)
            scale = min(1.0, 1.0 / (float(gn) + 1e-6))
            assert (a * scale - b).abs().max() < 2e-6, float((a * scale - b).abs().max())
        opt.step()
    for a, b in zip(params, p0):
        assert (a - b).abs().max() < 1e-6
    # and the rank-mean shortcut would NOT have matched: the shards hold different numbers of valid targets
    torch.manual_seed(0)
    net = _Head()
    means = []
    for lo, hi in ((0, 7), (7, 11)):
        pv, pn = net(x[lo:hi])
        means.append(float(net.verb_loss(pv, verb[lo:hi]) + net.nouns_loss(pn, nouns[lo:hi])))
    pv, pn = net(x)
    assert abs(sum(means) / 2 - float(net.verb_loss(pv, verb) + net.nouns_loss(pn, nouns))) > 1e-3


def _eval_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from situation_recognition_amd import parallel
    from situation_recognition_amd.imsitu_encoder import imsitu_encoder
    from situation_recognition_amd.imsitu_scorer import imsitu_scorer
    from situation_recognition_amd.sr import _EvalShard
    parallel.init_from_env(backend="gloo")
    enc = imsitu_encoder.synthetic(V=10, NR=8, L=20, R=4, seed=2)
    g = torch.Generator().manual_seed(3)
    N = 23                                                               # not a multiple of the world size
    verbs = torch.randint(0, 10, (N,), generator=g)
    gold = torch.randint(0, 20, (N, 3, 4), generator=g)
    pv, pn, pg = torch.randn(N, 10, generator=g), torch.randn(N, 4, 20, generator=g), torch.randn(N, 4, 20, generator=g)
    pv[torch.arange(0, N, 2), verbs[::2]] += 5
    ids = torch.tensor(list(_EvalShard(N, rank, world)))
    res = {}
    for k in (1, 5):
        sc = imsitu_scorer(enc, k, 3)
        sc.add_point_both(pv[ids], verbs[ids], pn[ids], gold[ids], pg[ids])
        sc.all_reduce_()
        full = imsitu_scorer(enc, k, 3)
        full.add_point_both(pv, verbs, pn, gold, pg)
        res[k] = (sc.get_average_results_both(), full.get_average_results_both(), len(ids))
    out[rank] = res
    dist.destroy_process_group()


def test_sharded_evaluation_reduces_to_the_whole_set():
    """sr.eval with several ranks: unpadded shards (every sample exactly once) + all-reduced card sums and counts give the
    metric of the whole set on every rank (the reference evaluates the whole dev set in one process, sr.py:165-231)."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_eval_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0][1][2] + out[1][1][2] == 23
    for r in (0, 1):
        for k in (1, 5):
            got, want, _ = out[r][k]
            assert got.keys() == want.keys()
            for key in want:
                assert abs(got[key] - want[key]) < 1e-12, (r, k, key)
