"""Data-parallel plumbing for the hot path: one process per GPU, graphs sharded across ranks, ONE data-path-free
collective per step -- a sum all-reduce of flat gradient buckets (torch.distributed backend "nccl" == RCCL over xGMI on
ROCm; "gloo" in the CPU tests).

The reference is single-device; a batch is a disjoint union of independent graphs, so forward and
backward need no exchange.  Gradients of this path are a few MB, i.e. latency-bound on xGMI, so they
travel as one flattened bucket per (dtype, device) whose slices ARE the parameters' .grad tensors (no pack/unpack copies).

The GC models normalise with BatchNorm1d over the NODES of the batch (gconv.py:187-194, rgconv.py:85-93): under data
parallelism each replica would see only its shard's statistics.  `SyncBatchNorm1d` restores the single-process result:
per-rank (count, mean, M2) are gathered in one small collective and merged (Chan's parallel-variance update -- the
numerically safe form of the [sum, sum-of-squares] reduction), and the backward all-reduces [sum dy, sum dy*xhat]
(SURVEY.md 8e caveat 1).  `convert_sync_batchnorm(model)` swaps every BatchNorm1d in place, keeping parameter and buffer
names, so reference state_dicts still load.
"""
import torch
import torch.distributed as dist


# With a world of one the collectives are identities and are skipped -- unless this is set (tests: it lets a single-GPU box run
# the RCCL all-reduce / all-gather calls the multi-GPU job makes).
RUN_COLLECTIVES_IN_A_WORLD_OF_ONE = False


def _dist_on():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or RUN_COLLECTIVES_IN_A_WORLD_OF_ONE)


def shard_graphs(batch_num_nodes, batch_num_edges, world_size):
    """Contiguous graph ranges per rank, balanced by nodes + edges.  Returns [(g_begin, g_end)] * world_size."""
    w = (torch.as_tensor(batch_num_nodes).long() + torch.as_tensor(batch_num_edges).long()).cpu()
    G = int(w.numel())
    csum = torch.cumsum(w, 0)
    total = int(csum[-1]) if G else 0
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        # first graph whose cumulative weight reaches the target closes the shard
        g = int(torch.searchsorted(csum.double(), torch.tensor([target], dtype=torch.float64), right=False)) + 1 if G else 0
        bounds.append(min(max(g, bounds[-1]), G))
    bounds.append(G)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def dp_loss_scale(local_graphs, global_graphs, world_size=None):
    """Factor for a rank's batch-MEAN loss (main.py:41 F.nll_loss, train.py:623-627) so that the AVERAGED gradient bucket
    equals the gradient of the global-batch mean when shards hold different numbers of graphs: B_r * W / B (1 when equal)."""
    if world_size is None:
        world_size = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return float(local_graphs) * world_size / float(global_graphs)


class FlatGradBucket:
    """All parameters' gradients live in flat buffers (one per dtype and device); all_reduce() sums each across ranks in
    one collective.

    The parameters' .grad tensors are views of the flat buffers, and that aliasing is RE-ESTABLISHED on every zero() and
    all_reduce(): the reference's loops call optimizer.zero_grad() (train.py:836, main.py:38), which on torch >= 2.0 sets
    .grad to None, after which autograd allocates fresh gradient tensors -- those are copied into their slice and .grad is
    pointed back at it, so the collective always carries the step's gradients (never a stale buffer)."""

    def __init__(self, params, average=True):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self.average = average
        self._groups = {}                               # (dtype, device) -> [flat, [(param, offset)]]
        for p in self.params:
            g = self._groups.setdefault((p.dtype, p.device), [None, []])
            off = sum(q.numel() for q, _ in g[1])
            g[1].append((p, off))
        for key, g in self._groups.items():
            n = sum(q.numel() for q, _ in g[1])
            g[0] = torch.zeros(n, dtype=key[0], device=key[1])
        self._meta = [(p.dtype, p.device, p.numel()) for p in self.params]
        self._bind(copy_existing=False)

    # the single-bucket accessors of round 1 (bench.py, tests): the first (usually only) flat buffer
    @property
    def flat(self):
        return next(iter(self._groups.values()))[0]

    @property
    def numel(self):
        return sum(g[0].numel() for g in self._groups.values())

    def buckets(self):
        return [g[0] for g in self._groups.values()]

    def _check(self):
        for p, (dt, dev, n) in zip(self.params, self._meta):
            if p.dtype != dt or p.device != dev or p.numel() != n:
                raise RuntimeError("FlatGradBucket: a parameter changed dtype / device / size after the bucket was built "
                                   "(%s %s -> %s %s); build the bucket after .to(...)" % (dt, dev, p.dtype, p.device))

    def _bind(self, copy_existing):
        """Point every p.grad at its slice.  copy_existing: gradients that live elsewhere (fresh tensors autograd made after
        zero_grad(set_to_none=True)) are copied in first -- all of a bucket's strays in ONE multi-tensor launch; a missing
        gradient (unused parameter) becomes zeros."""
        self._check()
        for flat, entries in self._groups.values():
            dsts, srcs = [], []
            for p, off in entries:
                view = flat[off:off + p.numel()].view_as(p)
                g = p.grad
                if g is not None and g.data_ptr() == view.data_ptr() and g.shape == view.shape and g.dtype == view.dtype:
                    continue
                if copy_existing:
                    if g is None:
                        view.zero_()
                    else:
                        if g.dtype != view.dtype or g.device != view.device:
                            raise RuntimeError("FlatGradBucket: gradient dtype / device differs from its parameter's")
                        dsts.append(view)
                        srcs.append(g.detach())
                p.grad = view
            if len(dsts) == 1:
                dsts[0].copy_(srcs[0])
            elif dsts and len(dsts) == len(entries):
                # the usual case after zero_grad(set_to_none=True): every gradient of the bucket is a fresh tensor, and the slices
                # tile the flat buffer in order -- ONE concatenation kernel (a multi-tensor copy of 7 tensors costs 4x as much)
                torch.cat([g_.reshape(-1) for g_ in srcs], out=flat)
            elif dsts:
                torch._foreach_copy_(dsts, srcs)

    def zero(self, set_to_none=False):
        """Zero all gradients and (re)alias them to the buckets: use INSTEAD of, or right after, optimizer.zero_grad().
        set_to_none=True is optimizer.zero_grad()'s own default: every .grad becomes None, autograd then WRITES the step's
        gradients (no zero fill, no accumulate launch per parameter) and pack() / all_reduce() gathers them into the bucket."""
        if set_to_none:
            for p in self.params:
                p.grad = None
            return
        for flat, _ in self._groups.values():
            flat.zero_()
        self._bind(copy_existing=False)

    def pack(self):
        """Gather the step's gradients into the flat buffers (one launch per bucket) and alias .grad to them; all_reduce() does
        this itself -- call it separately to keep the copy inside a captured step while the collective stays outside."""
        self._bind(copy_existing=True)

    def all_reduce(self, async_op=False):
        """Call after loss.backward().  Returns the list of work handles (async_op) or None."""
        self._bind(copy_existing=True)
        if not _dist_on():
            return None
        works = []
        # RCCL averages inside the collective (one launch less per bucket and step); gloo has no AVG: scale first
        avg_op = self.average and dist.get_backend() == "nccl"
        for flat, _ in self._groups.values():
            if self.average and not avg_op:
                flat.div_(dist.get_world_size())
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.AVG if avg_op else dist.ReduceOp.SUM, async_op=async_op))
        return works if async_op else None

    def bytes(self):
        return sum(g[0].numel() * g[0].element_size() for g in self._groups.values())


class OverlappedGradReducer:
    """Gradient all-reduce that runs UNDER the backward pass: one FlatGradBucket per parameter group (normally one per layer, in
    forward order), each reduced asynchronously the moment its last gradient has been accumulated -- while autograd is still
    working on the layers below it.  The communication stream (RCCL's own, or gloo's worker thread) waits only for the work
    queued so far, so the collective of layer L overlaps the backward launches of layers L-1, L-2, ...

        reducer = OverlappedGradReducer([layer.parameters() for layer in model.layers])
        for batch in loader:
            optimizer.zero_grad()              # set_to_none=True (the default) is fine: see FlatGradBucket
            loss(model(batch)).backward()      # hooks: pack + async all-reduce per group, as the groups complete
            reducer.finish()                   # wait for the collectives (and reduce groups that never completed)
            optimizer.step()

    Within ONE RGIN layer nothing can be overlapped: the conv's weight gradient needs the per-graph gradient sums that the
    input-gradient pass produces as a by-product (the absorbed fold), so it is the last launch of the layer's backward; the
    overlap is between layers (the reference stacks 3, config.py `rgin_num_layers`).  `launched` logs, per step, (group index,
    number of parameters of LATER-completing groups that had no gradient yet) -- what the tests use to show the collective left
    before backward ended."""

    def __init__(self, param_groups, average=True):
        self.buckets = [FlatGradBucket(list(g), average=average) for g in param_groups]
        if not self.buckets:
            raise ValueError("no parameter groups")
        self._pending = [set() for _ in self.buckets]
        self._sync, self._poisoned = True, False
        self._works = [None] * len(self.buckets)
        self._done = [False] * len(self.buckets)
        self.launched = []
        self._handles = []
        for gi, b in enumerate(self.buckets):
            for p in b.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(gi)))
        self._arm()

    # The ORDER of the collectives is the same on every rank whatever each rank's data does: groups leave in reverse forward
    # order (last group first -- the order backward completes them in), and a group whose gradients are all there leaves only
    # once every group behind it in that order has left.  A rank on which some parameter of a later group got no gradient (a
    # data-dependent branch, an empty shard) therefore holds the earlier groups back until finish() instead of enqueuing its
    # all-reduces in another order than its peers (a hang -- or, with equally sized per-layer buckets, layer A averaged with
    # layer B).
    def _arm(self):
        for gi, b in enumerate(self.buckets):
            self._pending[gi] = {id(p) for p in b.params}
            self._works[gi], self._done[gi] = None, False
        self._next = len(self.buckets) - 1              # the group whose collective must be issued next
        self.launched = []

    def _make_hook(self, gi):
        def hook(param):
            if not self._sync:                          # no_sync(): gradients accumulate locally, nothing is counted or sent
                return
            if self._poisoned:
                raise RuntimeError("OverlappedGradReducer: a previous backward() broke the step protocol; call reset() "
                                   "(and zero the gradients) before the next step")
            if self._done[gi]:
                self._poisoned = True                   # the flat buffer holds AVERAGED gradients + a local one: unusable
                raise RuntimeError("OverlappedGradReducer: a gradient of group %d arrived after the group's all-reduce was "
                                   "issued (a second backward() without finish(): the flat buffer already holds the AVERAGED "
                                   "gradients). Call finish() after every backward(); accumulate micro-batches under "
                                   "no_sync() and let the last backward() reduce. The reducer stays unusable until reset()." % gi)
            self._pending[gi].discard(id(param))
            self._drain()
        return hook

    def _drain(self):
        while self._next >= 0 and not self._pending[self._next]:
            self._launch(self._next)

    def _launch(self, gi):
        assert gi == self._next and not self._done[gi]
        waiting = sum(1 for gj, b in enumerate(self.buckets) if not self._done[gj] and gj != gi
                      for p in b.params if id(p) in self._pending[gj])
        self.launched.append((gi, waiting))
        self._done[gi] = True
        self._next = gi - 1
        self._works[gi] = self.buckets[gi].all_reduce(async_op=True)

    def no_sync(self):
        """Context manager for gradient accumulation (torch DDP's no_sync): backward() calls inside it only ACCUMULATE into the
        buckets' .grad views -- no hook counts, no collective.  The first backward() after the block reduces the sum:

            with reducer.no_sync():
                for mb in micro_batches[:-1]:
                    loss(model(mb)).backward()
            loss(model(micro_batches[-1])).backward(); reducer.finish()

        Every rank must run the same number of backward() calls outside no_sync() (as with DDP)."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            if any(self._done):
                raise RuntimeError("OverlappedGradReducer.no_sync(): a step is in flight (call finish() first)")
            prev, self._sync = self._sync, False
            try:
                yield self
            finally:
                self._sync = prev
                self._arm()                             # the next backward() starts a fresh count
        return _cm()

    def reset(self):
        """Re-arm WITHOUT communicating: waits for collectives already issued (their results are the caller's to discard, e.g.
        with zero_grad()) and clears the poisoned state.  Call it on EVERY rank -- a rank that skips a collective its peers
        issue would hang them."""
        for works in self._works:
            for w in works or ():
                w.wait()
        self._poisoned = False
        self._arm()

    def finish(self):
        """Call after backward(): reduces the groups that are still held back (a parameter without a gradient, in them or in a
        group behind them) in the same fixed order, waits for every collective and re-arms the hooks for the next step."""
        if self._poisoned:
            raise RuntimeError("OverlappedGradReducer.finish(): the step protocol was broken (see the earlier error); call reset()")
        while self._next >= 0:
            self._launch(self._next)
        for works in self._works:
            for w in works or ():
                w.wait()
        self._arm()

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def bytes(self):
        return sum(b.bytes() for b in self.buckets)


# ---------------------------------------------------------------------------------------------------------------------
# BatchNorm over the nodes of a sharded batch
# ---------------------------------------------------------------------------------------------------------------------
def _merge_stats(cnt, mean, m2):
    """Chan et al.: merge per-rank (count [W], mean [W,C], M2 [W,C]) into the global (count, mean, M2)."""
    n = cnt.sum()
    w = (cnt / n.clamp(min=1.0)).view(-1, 1)
    gmean = (w * mean).sum(0)
    gm2 = m2.sum(0) + (cnt.view(-1, 1) * (mean - gmean).square()).sum(0)
    return n, gmean, gm2


class _SyncBNFunction(torch.autograd.Function):
    """Statistics and the normalisation run in fp32 (fp64 for fp64 inputs), whatever the storage type of x."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        ct = torch.float64 if x.dtype == torch.float64 else torch.float32
        n_local, C = x.shape
        xf = x.to(ct)
        if n_local > 0:
            mean = xf.mean(0)
            m2 = (xf - mean).square().sum(0)
        else:
            mean = torch.zeros(C, dtype=ct, device=x.device)
            m2 = torch.zeros_like(mean)
        packed = torch.cat([torch.full((1,), float(n_local), dtype=ct, device=x.device), mean, m2])
        W = dist.get_world_size(group)
        allp = torch.empty(W * packed.numel(), dtype=ct, device=x.device)
        dist.all_gather_into_tensor(allp, packed, group=group)
        allp = allp.view(W, packed.numel())
        n, gmean, gm2 = _merge_stats(allp[:, 0], allp[:, 1:1 + C], allp[:, 1 + C:])
        var = gm2 / n.clamp(min=1.0)                                  # biased, as F.batch_norm normalises
        rstd = torch.rsqrt(var + eps)
        xhat = (xf - gmean) * rstd
        y = xhat * weight.to(ct) + bias.to(ct) if weight is not None else xhat
        ctx.save_for_backward(xhat, weight, rstd, n)
        ctx.group = group
        ctx.mark_non_differentiable(gmean, var, n)
        return y.to(x.dtype), gmean, var, n

    @staticmethod
    def backward(ctx, dy, _gm, _gv, _gn):
        xhat, weight, rstd, n = ctx.saved_tensors
        ct = xhat.dtype
        dyf = dy.to(ct)
        C = xhat.shape[1]
        s = torch.cat([dyf.sum(0), (dyf * xhat).sum(0)])              # local sums: also dbeta / dgamma (the gradient
        dbeta, dgamma = s[:C].clone(), s[C:].clone()                  # all-reduce of the step sums those across ranks)
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=ctx.group)
        g = weight.to(ct) if weight is not None else torch.ones(C, dtype=ct, device=dy.device)
        dx = (g * rstd) * (dyf - s[:C] / n - xhat * (s[C:] / n))
        if weight is None:
            return dx.to(dy.dtype), None, None, None, None
        return dx.to(dy.dtype), dgamma.to(weight.dtype), dbeta.to(weight.dtype), None, None


class SyncBatchNorm1d(torch.nn.BatchNorm1d):
    """BatchNorm1d over [rows, C] whose training statistics cover the rows of ALL ranks (same parameters, buffers and
    state_dict names as torch.nn.BatchNorm1d; identical to it when torch.distributed is not initialised or in eval mode).
    Works on any backend (one all_gather of 2C+1 floats forward, one all_reduce of 2C floats backward)."""

    def __init__(self, *a, process_group=None, **k):
        super().__init__(*a, **k)
        self.process_group = process_group

    fuse_relu = False                                # set by convert_sync_batchnorm when the module it replaces applied the ReLU too

    def forward(self, x):
        if not (self.training and _dist_on()) or x.dim() != 2:
            if x.is_cuda:                                        # one process: the HIP BatchNorm launches (ReLU included)
                from .graph_classification.models import hip_batch_norm_forward
                return hip_batch_norm_forward(self, x, self.fuse_relu)
            y = super().forward(x)
            return torch.relu(y) if self.fuse_relu else y
        y = self._forward(x)
        return torch.relu(y) if self.fuse_relu else y

    def _forward(self, x):
        y, mean, var, n = _SyncBNFunction.apply(x, self.weight, self.bias, self.eps, self.process_group)
        if self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                mom = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                unbiased = var * (n / (n - 1.0).clamp(min=1.0))
                self.running_mean.mul_(1.0 - mom).add_(mean.to(self.running_mean.dtype), alpha=mom)
                self.running_var.mul_(1.0 - mom).add_(unbiased.to(self.running_var.dtype), alpha=mom)
        return y


def convert_sync_batchnorm(module, process_group=None):
    """Replace every torch.nn.BatchNorm1d below `module` by SyncBatchNorm1d sharing the same Parameters and buffers
    (names unchanged: reference state_dicts keep loading).  Returns `module` (or its replacement if it is one itself)."""
    if isinstance(module, torch.nn.BatchNorm1d) and not isinstance(module, SyncBatchNorm1d):
        new = SyncBatchNorm1d(module.num_features, module.eps, module.momentum, module.affine, module.track_running_stats,
                              process_group=process_group)
        if module.affine:
            new.weight, new.bias = module.weight, module.bias
        if module.track_running_stats:
            new.running_mean, new.running_var = module.running_mean, module.running_var
            new.num_batches_tracked = module.num_batches_tracked
        new.training = module.training
        new.fuse_relu = bool(getattr(module, "fuse_relu", False))     # graph_classification.models.HipBatchNorm1d(fuse_relu=True)
        return new
    for name, child in list(module.named_children()):
        new = convert_sync_batchnorm(child, process_group)
        if new is not child:
            # a module registered under several names (GIN: nns.i IS convs.i.nn) is replaced once per name below; the
            # shared CHILD objects (the Sequential) stay shared because only their BatchNorm entries are swapped in place
            setattr(module, name, new)
    return module
