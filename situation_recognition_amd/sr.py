#!/usr/bin/env python3
"""Training / evaluation / single-image inference driver with the reference's command line (reference sr.py:383-539)
on top of the MI355X hot path.

    python -m situation_recognition_amd.sr --dataset_folder imSitu --imgset_dir resized_256 [--train_file overfitting.json]
    python -m torch.distributed.run --nproc-per-node 8 -m situation_recognition_amd.sr ...        # data parallel

Same flags, same printed lines, same checkpoint keys (epoch, avg_scores, verb_losses, nouns_losses, val_*,
model_state_dict, optimizer_state_dict).  Differences: one process per GPU + one RCCL gradient all-reduce instead of
nn.DataParallel (sr.py:467-470); bf16 storage instead of fp16 autocast + GradScaler (no loss scaling needed: bf16 has
fp32's exponent range); the encoder cache is JSON, not a pickled object; extra knobs --backbone/--steps/--dtype.
"""
import json
import os
from argparse import ArgumentParser
from pathlib import Path

import torch

from . import imsitu_loader, parallel, utils
from .imsitu_encoder import imsitu_encoder
from .imsitu_scorer import imsitu_scorer
from .model import FCGGNN


def _mean8(top1_a, top5_a):
    s = top1_a['verb'] + top1_a['value'] + top1_a['value-all'] + top5_a['verb'] + top5_a['value'] + top5_a['value-all'] + \
        top1_a['gt-value'] + top1_a['gt-value-all']
    return s / 8 * 100                                                   # sr.py:96-100


def _print_scores(prefix, losses, top1_a, top5_a, avg):
    print('{} = [v: {:.2f}, n: {:.2f}, gt: {:.2f}]'.format(prefix, *losses))
    gt = {k: top1_a[k] for k in ('gt-value', 'gt-value-all')}
    one = {k: top1_a[k] for k in ('verb', 'value', 'value-all')}
    print('{}\n{}\n{}, mean = {:.2f}'.format(utils.format_dict(one, '{:.2f}', '1-'), utils.format_dict(top5_a, '{:.2f}', '5-'),
                                             utils.format_dict(gt, '{:.2f}', ''), avg))


class _EvalShard(torch.utils.data.Sampler):
    """Evaluation shard of a rank: samples rank, rank+world, ... -- no padding, no duplicates (DistributedSampler pads by
    repeating samples, which would count them twice in the metric)."""

    def __init__(self, n, rank, world):
        self.ids = range(rank, n, world)

    def __iter__(self):
        return iter(self.ids)

    def __len__(self):
        return len(self.ids)


def eval(model, loader, encoder, logging=False):                         # noqa: A001  (reference name, sr.py:165)
    """reference sr.py:165-232.  With several ranks each one scores its shard of the set and the score-card sums and sample
    counts are all-reduced, so every rank returns (and rank 0 prints / checkpoints) the scorer metrics of the WHOLE set, exactly
    as the reference's single process does.  The three validation LOSSES are, as in the reference (sr.py:199-214), the mean over
    batches of per-batch means -- here over every rank's batches: the same estimator on a different partition of the set into
    batches (batch_size // world per rank, ragged last batches), so their last digits depend on the world size, as the
    reference's depend on its batch size."""
    model.eval()
    dev = next(model.parameters()).device
    top1, top5 = imsitu_scorer(encoder, 1, 3), imsitu_scorer(encoder, 5, 3)
    sums, n = [0.0, 0.0, 0.0], 0
    with torch.no_grad():
        for _, img, verb, nouns in loader:
            img, verb, nouns = img.to(dev), verb.to(dev), nouns.to(dev)
            pv, pn, pg = model(img, verb)
            top1.add_point_both(pv, verb, pn, nouns, pg)
            top5.add_point_both(pv, verb, pn, nouns, pg)
            for i, l in enumerate((model.verb_loss(pv, verb), model.nouns_loss(pn, nouns), model.nouns_loss(pg, nouns))):
                sums[i] += l.item()
            n += 1
    if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        t = torch.tensor(sums + [float(n)], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t)
        sums, n = t[:3].tolist(), int(t[3])
        top1.all_reduce_()
        top5.all_reduce_()
    losses = [s / max(n, 1) for s in sums]
    val_losses = {'verb_loss': losses[0], 'nouns_loss': losses[1], 'gt_loss': losses[2]}
    avg = 0
    if logging:
        t1, t5 = top1.get_average_results_both(), top5.get_average_results_both()
        avg = _mean8(t1, t5)
        _print_scores('val losses', losses, t1, t5, avg)
        print()
    return top1, top5, val_losses, avg


def train(model, train_loader, dev_loader, optimizer, max_epoch, encoder, model_saving_name, folder, checkpoint=None):
    """reference sr.py:15-162."""
    rank = int(os.environ.get("RANK", "0"))
    dev = next(model.parameters()).device
    hist = {k: [] for k in ('avg_scores', 'verb_losses', 'nouns_losses', 'val_avg_scores', 'val_verb_losses', 'val_nouns_losses')}
    epoch = 0
    if checkpoint is not None:
        epoch = checkpoint['epoch']
        for k in hist:
            hist[k] = checkpoint[k]
        model.load_state_dict(checkpoint['model_state_dict'])
        optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
    params = [p for p in model.parameters() if p.requires_grad]
    bucket = parallel.GradBucket(params) if torch.distributed.is_initialized() else None
    model.train()
    for e in range(epoch, max_epoch):
        for ld in (train_loader, getattr(train_loader, "sampler", None)):   # a new permutation every epoch (DistributedSampler
            if hasattr(ld, "set_epoch"):                                     # replays epoch 0's until told otherwise)
                ld.set_epoch(e)
        if rank == 0:
            print('Epoch-{}, lr: {:.4f}'.format(e, optimizer.param_groups[0]['lr']))
        top1, top5 = imsitu_scorer(encoder, 1, 3), imsitu_scorer(encoder, 5, 3)
        acc = [0.0, 0.0, 0.0]
        for _, img, verb, nouns in train_loader:
            img, verb, nouns = img.to(dev), verb.to(dev), nouns.to(dev)
            if bucket is None:
                optimizer.zero_grad()
            else:
                bucket.zero()                                            # .grad tensors are views of the flat bucket
            pv, pn, pg = model(img, verb)
            if bucket is None:
                vl, nl = model.verb_loss(pv, verb), model.nouns_loss(pn, nouns)
                gl = model.nouns_loss(pg, nouns)
                (vl + nl).backward()                                     # sr.py:76-79 (gt loss is logged only)
            else:
                # loss means over the GLOBAL batch (the reference computes them after DataParallel's gather, sr.py:67-76):
                # per-rank sums over global denominators, then a plain SUM of the gradients, launched bucket by bucket
                # from autograd hooks while the backward is still running
                loss, vl, nl, (_, n_glob) = parallel.global_batch_loss(model, pv, pn, verb, nouns)
                with torch.no_grad():
                    gl = model.nouns_loss(pg, nouns, denoms=n_glob)
                loss.backward()
                bucket.finish()
                logged = torch.stack([vl.detach(), nl.detach(), gl.detach()])
                torch.distributed.all_reduce(logged)                     # shares -> the global-batch values, for the log line
                vl, nl, gl = logged[0], logged[1], logged[2]
            torch.nn.utils.clip_grad_norm_(params, 1)                    # sr.py:81
            optimizer.step()
            top1.add_point_both(pv, verb, pn, nouns, pg)
            top5.add_point_both(pv, verb, pn, nouns, pg)
            for i, l in enumerate((vl, nl, gl)):
                acc[i] += l.item()
        nb = max(len(train_loader), 1)
        if bucket is not None:                                           # the epoch's training metric over every rank's samples
            top1.all_reduce_()
            top5.all_reduce_()
        t1, t5 = top1.get_average_results_both(), top5.get_average_results_both()
        avg = _mean8(t1, t5)
        hist['avg_scores'].append(avg)
        hist['verb_losses'].append(acc[0] / nb)
        hist['nouns_losses'].append(acc[1] / nb)
        if rank == 0:
            _print_scores('training losses', [a / nb for a in acc], t1, t5, avg)
            print('-' * 50)
        _, _, val_losses, val_avg = eval(model, dev_loader, encoder, logging=(rank == 0))
        model.train()
        hist['val_avg_scores'].append(val_avg)
        hist['val_verb_losses'].append(val_losses['verb_loss'])
        hist['val_nouns_losses'].append(val_losses['nouns_loss'])
        if rank == 0:
            _plot(hist, os.path.join(folder, model_saving_name + '.png'))
            ck = dict(hist, epoch=e + 1, model_state_dict=model.state_dict(), optimizer_state_dict=optimizer.state_dict())
            torch.save(ck, os.path.join(folder, model_saving_name))


def _plot(hist, path):                                                   # sr.py:132-142, optional dependency
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        return
    for k, lab, st in (('verb_losses', 'verb losses', '-'), ('nouns_losses', 'nouns losses', '-'), ('avg_scores', 'accuracy mean', '-'),
                       ('val_verb_losses', 'val verb losses', '-.'), ('val_nouns_losses', 'val nouns losses', '-.'),
                       ('val_avg_scores', 'val accuracy mean', '-.')):
        plt.plot(hist[k], st, label=lab)
    plt.grid(); plt.legend(); plt.savefig(path); plt.clf()


def results(model, image, encoder, gt_verb, space_json=os.path.join("imSitu", "imsitu_space.json")):
    """Single-image inference, reference sr.py:235-281 (incl. its softmax over ROLES, dim=0, at line 264)."""
    from PIL import Image
    model.eval()
    dev = next(model.parameters()).device
    with open(space_json) as f:
        space = json.load(f)
    nouns_space, verbs_space = space["nouns"], space["verbs"]
    img = encoder.dev_transform(Image.open(image).convert('RGB')).unsqueeze(0).to(dev)
    with torch.no_grad():
        if gt_verb and gt_verb in encoder.verb_list:
            verb_tensor, verb_prob = torch.tensor([encoder.verb_list.index(gt_verb)], device=dev), 100
        else:
            print("No ground truth verb found, calculating by myself...")
            logits = model.predict_verb(img, 1)
            verb_tensor = torch.argmax(logits, 1)
            verb_prob = torch.max(torch.softmax(logits.float(), dim=1)).item() * 100
        logits = model.predict_nouns(img, verb_tensor, 1).squeeze(0).float()
    nouns_tensor = torch.argmax(logits, 1)
    labels_prob = [p.item() * 100 for p in torch.max(torch.softmax(logits, dim=0), 1)[0]]
    verb_name = encoder.verb_list[int(verb_tensor)]
    roles = list(verbs_space[verb_name]["roles"].keys())
    labels = {}
    for count, i in enumerate(nouns_tensor[:len(roles)].tolist()):
        lab = encoder.label_list[i]
        labels[roles[count]] = '-' if lab in ('', 'UNK') else nouns_space[lab]['gloss'][0]
    return verb_name, verb_prob, labels, labels_prob


def build_parser():
    p = ArgumentParser(description='Situation recognition with GNN (MI355X hot path).')
    p.add_argument('--resume_model', type=str, default='')
    p.add_argument('--evaluate_dev', action='store_true')
    p.add_argument('--evaluate_test', action='store_true')
    p.add_argument('--test_img', type=str, default='')
    p.add_argument('--verb', type=str, default='')
    p.add_argument('--subset', type=int, default=0)
    p.add_argument('--model_saving_name', type=str, default='sr')
    p.add_argument('--saving_folder', type=str, default='checkpoints')
    p.add_argument('--imgset_dir', type=str, default='resized_256')
    p.add_argument('--dataset_folder', type=str, default='imSitu')
    p.add_argument('--train_file', type=str, default='train.json')
    p.add_argument('--dev_file', type=str, default='dev.json')
    p.add_argument('--test_file', type=str, default='test.json')
    p.add_argument('--batch_size', type=int, default=6144)
    p.add_argument('--num_workers', type=int, default=10)
    p.add_argument('--epochs', type=int, default=1000)
    p.add_argument('--lr', type=float, default=0.002)
    # knobs the reference hard-codes
    p.add_argument('--backbone', type=int, default=152, choices=[18, 34, 50, 101, 152])
    p.add_argument('--steps', type=int, default=4, help='GGNN message-passing steps (reference: 4)')
    p.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    p.add_argument('--encoder_file', type=str, default='', help='annotation file the vocabulary is built from (default train.json)')
    p.add_argument('--shards', type=str, default='', help='directory of pre-decoded uint8 shards (imsitu_shards): one sub-directory '
                   'per annotation file; decode + resize happen once, crop / flip / normalise on the GPU')
    p.add_argument('--make_shards', action='store_true', help='write the shards of --train_file/--dev_file/--test_file under --shards and exit')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.make_shards:                                  # host-only: decode + resize every image once (no GPU needed)
        from .imsitu_shards import write_shards
        for f in (args.train_file, args.dev_file, args.test_file):
            path = os.path.join(args.dataset_folder, f)
            if os.path.isfile(path):
                n = write_shards(args.imgset_dir, list(json.load(open(path))), _shard_dir(args, f), quiet=False)
                print('{}: {} shards under {}'.format(f, n, _shard_dir(args, f)))
        return
    if args.subset:
        # reference sr.py:284-380,529 (`analize_subset`): an IPython notebook pretty-printer over random dev images -- outside the
        # hot path.  Refuse instead of falling through to training (which would overwrite the checkpoint).
        raise SystemExit("--subset (analize_subset) is not implemented in this build; use --test_img for single images")
    rank, world, local = parallel.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("situation_recognition_amd.sr needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    Path(args.saving_folder).mkdir(exist_ok=True)
    load = lambda f: json.load(open(os.path.join(args.dataset_folder, f)))
    cache = os.path.join(args.saving_folder, 'encoder.json')
    if os.path.isfile(cache):
        print("Loading encoder file")
        encoder = imsitu_encoder.from_state(json.load(open(cache)))
    else:
        encoder = imsitu_encoder(load(args.encoder_file or 'train.json'), quiet=(rank != 0))
        if rank == 0:
            json.dump(encoder.state(), open(cache, 'w'))
    mk = lambda f, tf, shuffle: _loader(args, load(f), encoder, tf, shuffle, rank, world, fname=f)
    D = 2048 if args.backbone >= 50 else 512
    model = FCGGNN(encoder, D, steps=args.steps, backbone=args.backbone,
                   dtype=torch.bfloat16 if args.dtype == 'bf16' else torch.float32).to(dev)
    model.drop_seed_base += rank
    if rank == 0:
        print('Using', world, 'GPUs!')
    optimizer = torch.optim.Adamax([p for p in model.parameters() if p.requires_grad], lr=args.lr)
    checkpoint = None
    if len(args.resume_model) > 1:
        print('Resume training from: {}'.format(args.resume_model))
        path = os.path.join(args.saving_folder, args.resume_model)
        checkpoint = torch.load(path, map_location=dev, weights_only=True)
        utils.load_net(path, [model])
        args.model_saving_name = args.resume_model
    if args.evaluate_dev:
        print('=> evaluating model with dev-set...')
        eval(model, mk(args.dev_file, encoder.dev_transform, False), encoder, logging=True)
    elif args.evaluate_test:
        print('=> evaluating model with test-set...')
        eval(model, mk(args.test_file, encoder.dev_transform, False), encoder, logging=True)
    elif args.test_img:
        verb, verb_prob, labels, labels_prob = results(model, args.test_img, encoder, args.verb)
        print('&' * 50); print('Analizing: ', args.test_img); print('&' * 50)
        print('action ({:.2f}%): {}'.format(verb_prob, verb))
        for c, (k, v) in enumerate(labels.items()):
            print('{} ({:.2f}%): {}'.format(k, labels_prob[c], v))
    else:
        print('Model training started!')
        train(model, mk(args.train_file, encoder.train_transform, True), mk(args.dev_file, encoder.dev_transform, False),
              optimizer, args.epochs, encoder, args.model_saving_name, folder=args.saving_folder, checkpoint=checkpoint)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def _shard_dir(args, fname):
    return os.path.join(args.shards, os.path.splitext(os.path.basename(fname))[0])


def _loader(args, ann, encoder, transform, shuffle, rank, world, fname=None):
    if args.shards and fname is not None and os.path.isfile(os.path.join(_shard_dir(args, fname), "index.json")):
        from .imsitu_shards import ShardLoader
        local = int(os.environ.get("LOCAL_RANK", "0"))
        return ShardLoader(_shard_dir(args, fname), ann, encoder, max(1, args.batch_size // world), torch.device("cuda", local),
                           train=shuffle, rank=rank, world=world)
    ds = imsitu_loader.imsitu_loader(args.imgset_dir, ann, encoder, transform)
    sampler = None
    if world > 1:      # training: equal (padded) shards -- every batch ends in an all-reduce; evaluation: every sample once
        sampler = (torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True) if shuffle
                   else _EvalShard(len(ds), rank, world))
    return torch.utils.data.DataLoader(ds, pin_memory=True, batch_size=max(1, args.batch_size // world),
                                       shuffle=(shuffle and sampler is None), sampler=sampler, num_workers=args.num_workers)


if __name__ == '__main__':
    main()
