"""Data parallelism for the training step: one process per GPU, weights resident on every rank, each
rank reads its own shard of the batch, and the trainable gradients (35.9 M fp32 = 143.7 MB) are summed over the ranks
with RCCL over xGMI (torch.distributed's "nccl" backend; "gloo" on CPU for tests) -- in buckets, one per large
parameter tensor (each 2048 x 2048 GGNN matrix is its own 16.8 MB bucket), launched from autograd hooks the moment a
bucket's gradients exist, on the collective's own stream, so the exchange runs behind the rest of the backward.

Replaces the reference's `torch.nn.DataParallel` (sr.py:467-470), which every step scatters the input
from GPU 0, re-broadcasts all weights, gathers the logits and reduces the gradients onto GPU 0
(SURVEY 2a).  The frozen backbones (2 x 58 M parameters) are never communicated; BatchNorm statistics
stay per rank, exactly as under DataParallel (no SyncBN).

Exact loss semantics.  The reference computes its cross-entropy means on GPU 0 over the GATHERED global batch
(sr.py:67-76 after DataParallel's gather): verb_loss divides by the global batch size, and each of the three noun terms
(model.py:196-199, `ignore_index`) by the GLOBAL number of non-ignored targets.  A mean of per-rank means is a different
number whenever ranks hold different counts, so here every rank divides its loss SUMS by the global denominators
(`loss_denominators`: one all-reduce of four integers before backward) and the gradient exchange is a plain SUM --
which is then exactly the gradient of the reference's global-batch loss.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank).  World size 1 needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("SR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _world(group=None):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def loss_denominators(gt_verb, gt_nouns, ignore_index, group=None):
    """Global denominators of the reference's loss means: (B_global, n_global[3]) as fp32 tensors on the labels' device,
    where n[i] = number of targets of annotator i that are not `ignore_index`.  One all-reduce of 4 integers."""
    counts = torch.cat([torch.tensor([gt_verb.shape[0]], device=gt_nouns.device, dtype=torch.int64),
                        (gt_nouns != ignore_index).sum(dim=(0, 2)).to(torch.int64)])
    if _world(group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    counts = counts.to(torch.float32)
    return counts[0], counts[1:]


def global_batch_loss(model, pred_verb, pred_nouns, gt_verb, gt_nouns, group=None):
    """This rank's share of the reference's training loss `verb_loss + nouns_loss` (sr.py:67-76) over the GLOBAL batch:
    summing the returned values over the ranks gives the loss a single replica would compute on the gathered batch, and
    summing the gradients gives its gradient.  Returns (loss, verb_share, nouns_share, (B_global, n_global[3]))."""
    L = model.encoder.get_num_labels()
    b_glob, n_glob = loss_denominators(gt_verb, gt_nouns, L, group)
    vl = model.verb_loss(pred_verb, gt_verb, denom=b_glob)
    nl = model.nouns_loss(pred_nouns, gt_nouns, denoms=n_glob)
    return vl + nl, vl, nl, (b_glob, n_glob)


class HipComm:
    """RCCL communicator owned through the C ABI (`sr_comm_*` / `sr_allreduce_sum`, include/srhip.h): the exchange step of the
    data-parallel path without torch.distributed's collectives.  The ncclUniqueId travels from rank 0 to the others over the
    already-initialised torch.distributed process group (host-side rendezvous only); every rank must have selected its GPU.
    World size 1 needs no process group."""

    def __init__(self, rank=None, world=None, group=None):
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib, C
        self.world = _world(group) if world is None else world
        self.rank = (dist.get_rank(group) if dist.is_initialized() else 0) if rank is None else rank
        ident = (C.c_ubyte * 128)()
        if self.rank == 0:
            _lib.check(_lib.lib().sr_comm_unique_id(ident), "sr_comm_unique_id")
        if self.world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = (C.c_ubyte * 128).from_buffer_copy(box[0])
        handle = C.c_void_p()
        _lib.check(_lib.lib().sr_comm_init(ident, self.rank, self.world, C.byref(handle)), "sr_comm_init")
        self._h = handle
        self.stream = torch.cuda.Stream()          # the collective's own stream: it runs beside the rest of the backward

    def all_reduce_sum_(self, t, stream=None, after=None):
        """In-place sum of the contiguous fp32 / bf16 CUDA tensor `t` over the ranks, ordered after the work already enqueued on
        the current stream (and after the event `after`, if given: the producer of `t` may have run on another stream); returns an
        event the consumer waits on."""
        if not t.is_cuda or not t.is_contiguous():
            raise self._lib.SrError("all_reduce_sum_: contiguous CUDA tensor expected")
        s = stream or self.stream
        s.wait_stream(torch.cuda.current_stream())
        if after is not None:
            s.wait_event(after)
        self._lib.check(self._lib.lib().sr_allreduce_sum(self._h, t.data_ptr(), t.numel(), self._lib.dtype_code(t.dtype), s.cuda_stream),
                        "sr_allreduce_sum")
        t.record_stream(s)
        ev = torch.cuda.Event()
        ev.record(s)
        return ev

    def close(self):
        if self._h is not None and self._h.value:
            self._lib.lib().sr_comm_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _EventWork:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class GradBucket:
    """Flat fp32 buffer over the trainable parameters whose slices ARE the parameters' `.grad` tensors, cut into buckets.

        bucket.zero()              instead of optimizer.zero_grad()   (keeps .grad pointing into the buffer)
        loss.backward()            autograd accumulates into the views; when the last gradient of a bucket has been
                                   written AND every earlier bucket has been launched, its hook launches an asynchronous
                                   all-reduce(SUM) of that slice (same order of collectives on every rank, whatever order
                                   autograd finishes them in)
        bucket.finish()            launches what is left (parameters that received no gradient stay zero) and waits

    Afterwards every rank holds the SUM of the ranks' gradients (`average=True`: the mean, for losses that are per-rank
    means).  With the loss of `global_batch_loss` the sum is the global-batch gradient, so `clip_grad_norm_` after
    `finish()` sees what the reference's single replica sees (sr.py:79-81).
    """

    def __init__(self, params, group=None, average=False, min_bucket_bytes=4 << 20, comm=None):
        """`comm`: a HipComm -- the buckets are then summed by `sr_allreduce_sum` (RCCL through the C ABI) instead of
        torch.distributed's all_reduce (which is RCCL as well under the "nccl" backend, gloo in the CPU tests)."""
        self.params = [p for p in params if p.requires_grad]
        self.group, self.average, self.comm = group, average, comm
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.views, self.spans, o = [], [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            self.spans.append((o, o + p.numel()))
            o += p.numel()
        # buckets in REVERSE parameter order (the order gradients become available in); a tensor of >= min_bucket_bytes
        # closes the bucket it falls in, smaller ones (biases, small embeddings) ride along with their neighbours
        self.buckets, cur = [], []
        for i in reversed(range(len(self.params))):
            cur.append(i)
            if sum(self.params[j].numel() for j in cur) * 4 >= min_bucket_bytes:
                self.buckets.append(cur)
                cur = []
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {}
        for b, idxs in enumerate(self.buckets):
            for i in idxs:
                self._bucket_of[i] = b
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._pending = [len(b) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._done_ev = [None] * len(self.buckets)   # recorded on the stream of the hook that completed the bucket (CUDA only)
        self._next = 0                 # buckets [0, _next) have been launched: launches go out in bucket order on EVERY rank
        self.launch_order = []         # bucket indices in the order their collectives were issued since the last zero() (tests)
        self._works = []
        self._hooks = []
        for p in self.params:
            p.grad = self.views[self._index[id(p)]]
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def close(self):
        """Detach from the parameters: a second GradBucket over the same parameters must not find this one's hooks alive."""
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def zero(self):
        self.flat.zero_()
        for i, p in enumerate(self.params):
            if p.grad is None or p.grad.data_ptr() != self.views[i].data_ptr():
                p.grad = self.views[i]
        self._pending = [len(b) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._done_ev = [None] * len(self.buckets)
        self._next = 0
        self.launch_order = []
        self._works = []

    def _span(self, b):
        idxs = self.buckets[b]
        return min(self.spans[i][0] for i in idxs), max(self.spans[i][1] for i in idxs)

    def _launch(self, b):
        self._launched[b] = True
        self.launch_order.append(b)
        lo, hi = self._span(b)
        # In-order launching means bucket b may go out from the hook of a parameter of ANOTHER bucket, on whatever stream that hook
        # runs: the collective is ordered behind the event recorded when b's last gradient was accumulated (FCGGNN runs one noun
        # branch's backward on a side stream), not only behind the launching hook's stream.
        ev = self._done_ev[b]
        if self.comm is not None:
            if self.comm.world > 1:
                self._works.append(_EventWork(self.comm.all_reduce_sum_(self.flat[lo:hi], after=ev)))
        elif _world(self.group) > 1:
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
            self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_grad(self, p):
        i = self._index[id(p)]
        if p.grad.data_ptr() != self.views[i].data_ptr():      # .grad was reset to None before backward: adopt the new tensor
            self.views[i].copy_(p.grad)
            p.grad = self.views[i]
        b = self._bucket_of[i]
        if self._launched[b]:
            # a second backward() between two zero() calls: its gradients would miss the all-reduce that has already run
            raise RuntimeError("GradBucket: a gradient arrived for a bucket whose all-reduce was already launched -- "
                               "call zero() before every backward (gradient accumulation over several backwards is not supported)")
        self._pending[b] -= 1
        if self._pending[b] == 0 and self.flat.is_cuda:
            self._done_ev[b] = torch.cuda.Event()
            self._done_ev[b].record(torch.cuda.current_stream())
        # Rank-invariant launch order.  The order in which autograd completes the buckets depends on how the forward was BUILT, and
        # that can differ between ranks of one step (FCGGNN picks the packed-role / side-stream forms from its LOCAL batch, and
        # shard sizes differ by one): ranks issuing differently sized collectives in different orders hang RCCL or sum the wrong
        # slices.  So bucket b goes out only once buckets 0..b-1 have; a bucket that completes early waits for its predecessors
        # (finish() launches the rest, in order).
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def finish(self):
        for b in range(self._next, len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        self._next = len(self.buckets)
        for w in self._works:
            w.wait()
        self._works = []
        n = self.comm.world if self.comm is not None else _world(self.group)
        if self.average and n > 1:
            self.flat.div_(n)

    def reduce(self):
        """One-call form for code that does not use the hooks' overlap: everything not yet launched goes now."""
        self.finish()


def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of `total` samples for `rank` (sizes differ by at most 1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def barrier():
    if dist.is_initialized():
        dist.barrier()
