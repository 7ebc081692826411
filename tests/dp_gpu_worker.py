"""One rank of the two-rank data-parallel GPU test (tests/test_parallel_gpu.py starts two of these as child processes; not a test
module itself).  Both ranks share GPU 0 of the one-GPU box and exchange gradients over gloo -- the rehearsal path
SR_FORCE_DEVICE=0 SR_DIST_BACKEND=gloo (RCCL refuses two ranks on one device); everything else is the product's multi-rank
path: parallel.init_from_env, the real FCGGNN with its hand-written backward, parallel.GradBucket's hook-launched bucketed
all-reduce, parallel.global_batch_loss, clip_grad_norm_, Adamax (reference sr.py:63-83 under DataParallel, sr.py:467-470).

usage: dp_gpu_worker.py OUT_PREFIX MODE      MODE = frozen_bn | train_bn
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SPLIT = (0, 7, 11)                  # rank 0: 7 images, rank 1: 4 (unequal shards, unequal numbers of valid roles)


def build(mode):
    """The G3 'bottleneck' model (weights written by the reference itself, tests/golden) on the reference's 5-image vocabulary."""
    from golden_util import load, overfitting_json, sub
    from situation_recognition_amd.imsitu_encoder import imsitu_encoder
    from situation_recognition_amd.model import FCGGNN
    g3 = load("g3_fcggnn_bottleneck.npz")
    enc = imsitu_encoder(overfitting_json(), quiet=True)
    net = FCGGNN(enc, int(g3["D"]), steps=4, backbone=int(g3["cfg_depth"]), dtype=torch.float32,
                 width=int(g3["cfg_width"]), blocks=tuple(int(b) for b in g3["cfg_blocks"]))
    net.load_state_dict(sub(g3, "state/"), strict=True)
    net.cuda()
    net.train()
    if mode == "frozen_bn":                               # images independent: the two-rank step must equal the global-batch step
        net.convnet_verbs.eval()
        net.convnet_nouns.eval()
    net.verb_classifier[0].p = 0.0                        # Dropout masks are per rank: identity, as in the G5 golden
    net.nouns_classifier[0].p = 0.0
    return net, enc, tuple(g3["img"].shape[1:])


def data(enc, chw):
    g = torch.Generator().manual_seed(77)
    B = SPLIT[-1]
    img = torch.randn((B,) + chw, generator=g).clamp_(-2.2, 2.7)
    V, L, R = enc.get_num_verbs(), enc.get_num_labels(), enc.get_max_role_count()
    verb = torch.randint(0, V, (B,), generator=g)
    nouns = torch.randint(0, L, (B, 3, R), generator=g)
    counts = torch.tensor([enc.get_role_count(int(v)) for v in verb])
    nouns[(torch.arange(R)[None, :] >= counts[:, None])[:, None, :].expand(B, 3, R)] = L
    nouns[9, 1, 0] = L                                    # annotators differ: three different denominators
    return img, verb, nouns


def main():
    out_prefix, mode = sys.argv[1], sys.argv[2]
    from situation_recognition_amd import parallel
    rank, world, _ = parallel.init_from_env()
    torch.cuda.set_device(int(os.environ.get("SR_FORCE_DEVICE", "0")))
    net, enc, chw = build(mode)
    img, verb, nouns = data(enc, chw)
    lo, hi = SPLIT[rank], SPLIT[rank + 1]
    img, verb, nouns = img[lo:hi].cuda(), verb[lo:hi].cuda(), nouns[lo:hi].cuda()
    params = [p for p in net.parameters() if p.requires_grad]
    bucket = parallel.GradBucket(params, min_bucket_bytes=1 << 16)          # several buckets at this model size
    opt = torch.optim.Adamax(params, lr=0.002)
    hist = []
    for step in range(2):                                 # two steps: _Shadow refresh after the optimizer step, bucket.zero() re-arming
        bucket.zero()
        pv, pn, pg = net(img, verb)
        loss, vl, nl, _ = parallel.global_batch_loss(net, pv, pn, verb, nouns)
        loss.backward()
        bucket.finish()
        assert all(bucket._launched)
        share = torch.stack([vl.detach(), nl.detach()]).cpu()
        torch.distributed.all_reduce(share)
        grads = [p.grad.detach().clone().cpu() for p in params]
        gn = torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        hist.append(dict(grad_norm=float(gn), losses=share.tolist(), grads=grads,
                         pred_verb=pv.detach().cpu(), pred_nouns=pn.detach().cpu(), gt_pred_nouns=pg.detach().cpu()))
    torch.cuda.synchronize()
    torch.save(dict(hist=hist, params=[p.detach().cpu() for p in params], nbuckets=len(bucket.buckets),
                    names=[k for k, p in net.named_parameters() if p.requires_grad]), "%s.rank%d.pt" % (out_prefix, rank))
    parallel.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
