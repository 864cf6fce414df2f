"""Training loop pieces for the hot path, MI355X-first (SURVEY 8e + tools/train.py, tools/train_utils/__init__.py):

  * FlatParams      every parameter (and its gradient) is a view into ONE contiguous fp32 buffer, so the DDP exchange
                    is a single RCCL all-reduce over the whole 21 MB model (8e: "bucket = whole model"), gradient
                    clipping is one reduction and the update one streaming kernel.
  * OneCycle        tools/train_utils/optimization/learning_schedules_fastai.py:9-78 (cosine lr / momentum phases).
  * AdamOneCycle    optimizer type 'adam_onecycle' (optimization/__init__.py:17-30; fastai_optim.py:101-148:
                    Adam(betas=(mom, 0.99)), true weight decay on every group incl. BatchNorm) + clip_grad_norm_
                    (train_utils/__init__.py:60) fused in sassd_grad_sumsq / sassd_adam_step.
  * GradSync        one process per GPU; parameters broadcast from rank 0 once, gradients all-reduced (sum; the 1/world
                    mean is folded into the update kernel's grad_scale).  BatchNorm stays per-GPU like the reference.
  * train_one_iter / checkpoint_state / save / load   train_utils/__init__.py:36-66,120-170.
"""
import math

import torch
import torch.distributed as dist

from . import kernels as K


class FlatParams:
    """Parameters are views of `data`.  Gradients: autograd hands every parameter a fresh tensor (zero_grad() resets
    .grad to None, so AccumulateGrad steals the incoming gradient instead of launching one add kernel per parameter
    into a zero-filled buffer); collect() then gathers them into the contiguous gradient buffer with one multi-tensor
    copy and re-points .grad at the views.  `grad` (the flat buffer) collects lazily, so readers always see the
    gradients of the last backward."""

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        self.offsets, off = [], 0
        for n in sizes:
            self.offsets.append(off)
            off += (n + 3) // 4 * 4                         # keep every view 16-byte aligned
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self._grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.gviews = []
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.data[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + n].view(p.shape)
            self.gviews.append(self._grad[o:o + n].view(p.shape))
            p.grad = self.gviews[-1]
        self._gptr = [v.data_ptr() for v in self.gviews]
        K.bump_weights_generation()       # every parameter just moved: no cached weight image may survive this

    @property
    def grad(self):
        self.collect()
        return self._grad

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def collect(self, lo=0, hi=None):
        """Gather the gradients of params[lo:hi] into the flat buffer (idempotent; a parameter that received no
        gradient gets zeros)."""
        hi = len(self.params) if hi is None else hi
        dst, src = [], []
        for i in range(lo, hi):
            p, v = self.params[i], self.gviews[i]
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != self._gptr[i]:
                dst.append(v)
                src.append(g.detach())
            else:
                continue
            p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)


class PackPlan:
    """Every kernel-layout weight image of the training step, refreshed by ONE gather launch per element type right
    after the optimizer step (sassd_gather_pack) instead of ~60 individual pack launches spread over forward and
    backward: the sparse convs' MFMA-fragment packs and their transposed twins (data gradient), the direct conv packs
    of the 1x1 head convs (forward + data gradient) and -- under set_bev_precision("bf16") -- the bf16 images of the 3x3
    BEV convs (forward + data gradient).  Each image is a permutation (+ zero padding) of the flat parameter buffer;
    the index maps are built once by pushing index-valued weights through the individual pack routines, so the layouts
    stay defined in one place.  Packs not covered here (Winograd transforms of the fp32 mode) keep packing lazily."""

    def __init__(self, model, flat):
        import numpy as np
        from . import autograd as AG, spconv as SP
        from .detector import _HipConv2d
        self.flat = flat
        dev = flat.data.device
        base = flat.data.data_ptr()
        f32_maps, bf_maps = [], []
        self._f32, self._bf = [], []                  # (offset, numel, install(view))

        def idx_like(p):
            """(flat index + 1) of every element of p, int32, shaped like p."""
            off = (p.data_ptr() - base) // 4
            assert 0 <= off and off + p.numel() <= flat.numel, "parameter outside the flat buffer"
            return (torch.arange(p.numel(), dtype=torch.int32, device=dev) + (off + 1)).view(p.shape)

        def bits(t):
            """int32 -> the same words typed fp32 (the pack kernels only move words, they never compute)."""
            return t.contiguous().view(torch.float32)

        def add(maps, sinks, m, install):
            sinks.append((sum(x.numel() for x in maps), m.numel(), install))
            maps.append(m.reshape(-1))

        def add_f32(packed_bits, install):
            add(f32_maps, self._f32, packed_bits.view(torch.int32) - 1, install)

        def bf16_map(off, cout, cin, transposed):
            i = off + np.arange(cout * cin * 9, dtype=np.int64).reshape(cout, cin, 9)
            if transposed:                            # weight.transpose(0, 1).flip(2, 3): [cin, cout, 8 - tap]
                i = i.transpose(1, 0, 2)[:, :, ::-1]
                cout, cin = cin, cout
            cp = (cin + 31) // 32 * 32
            co_p = cout if transposed else AG.bf16_cout_pad(cout)      # forward: zero rows up to a multiple of 32 output channels
            pad = np.full((co_p, cp, 9), -1, np.int64)
            pad[:cout, :cin] = i
            cout = co_p
            return torch.from_numpy(pad.reshape(cout, cp // 8, 8, 9).transpose(3, 1, 0, 2).reshape(-1).astype(np.int32)).to(dev)

        def c1_map(off, cout, cin, transposed):
            """gather map of sassd_conv1x1_bf16_pack_weight: element j of lane l of (group g, k-step ks, row tile a) =
            W[64 g + 32 a + (l & 31)][16 ks + 8 (l >> 5) + j] of the layer's [cout, cin] array or of its transpose."""
            co_n, ci_n = (cin, cout) if transposed else (cout, cin)
            ks_n = 1 if ci_n <= 16 else 2 if ci_n <= 32 else 4 if ci_n <= 64 else 8 if ci_n <= 128 else 16
            g, ks, a, l, j = np.meshgrid(np.arange((co_n + 63) // 64), np.arange(ks_n), np.arange(2), np.arange(64),
                                         np.arange(8), indexing="ij")
            co, ci = 64 * g + 32 * a + (l & 31), 16 * ks + 8 * (l >> 5) + j
            idx = off + (ci * cin + co if transposed else co * cin + ci)
            idx = np.where((co < co_n) & (ci < ci_n), idx, -1)
            from . import _C
            assert idx.size == _C.lib().sassd_conv1x1_bf16_packed_elems(ci_n, co_n)
            return torch.from_numpy(idx.reshape(-1).astype(np.int32)).to(dev)

        bf16 = AG.bev_precision() == "bf16"
        for m in model.modules():
            if isinstance(m, SP.SparseConvolution) and m.weight.requires_grad:
                k = int(np.prod(m.kernel_size))
                wi = bits(idx_like(m.weight).reshape(k, m.in_channels, m.out_channels))

                def inst(view, m=m):
                    m._packed, m._packed_version = view, K.weight_key(m.weight)
                add_f32(K.spconv_pack_weight(wi), inst)
                if m.in_channels >= 16:
                    key = (m.weight.data_ptr(), (k, m.in_channels, m.out_channels))

                    # submanifold layers differentiate on the forward rulebook with the offset-reversed image
                    rev = isinstance(m, SP.SubMConv3d) and k == 27 and AG.SparseConvFn.subm_on_forward_table
                    key = key + ((True,) if rev else ())

                    def inst_t(view, m=m, key=key):
                        AG._sp_t_packs[key] = (K.weight_key(m.weight), view, m.weight)
                    add_f32(K.spconv_pack_weight_t(wi.flip(0).contiguous() if rev else wi), inst_t)
            elif isinstance(m, _HipConv2d) and m.weight.requires_grad:
                cout, cin, ks = m.out_channels, m.in_channels, m.kernel_size[0]
                off = (m.weight.data_ptr() - base) // 4
                wkey = (m.weight.data_ptr(), tuple(m.weight.shape))
                direct_fwd = ks == 1
                hw_any = 4                                   # (the shape test below does not depend on the map size)
                c1_fwd = ks == 1 and bf16 and K.conv1x1_bf16_supported(cin, cout, hw_any)
                c1_bwd = ks == 1 and bf16 and K.conv1x1_bf16_supported(cout, cin, hw_any)
                if c1_fwd:
                    def inst_c(view, m=m, wkey=wkey):
                        AG._bf16_1x1_packs[wkey + (False,)] = (K.weight_key(m.weight), view, m.weight)
                    add(bf_maps, self._bf, c1_map(off, cout, cin, False), inst_c)
                if c1_bwd:
                    def inst_ct(view, m=m, wkey=wkey):
                        AG._bf16_1x1_packs[wkey + (True,)] = (K.weight_key(m.weight), view, m.weight)
                    add(bf_maps, self._bf, c1_map(off, cout, cin, True), inst_ct)
                if ks == 3 and bf16:
                    def inst_b(view, m=m, wkey=wkey):
                        AG._bf16_packs[wkey + (False,)] = (K.weight_key(m.weight), view, m.weight)
                    add(bf_maps, self._bf, bf16_map(off, cout, cin, False), inst_b)
                    if cin % 32 == 0:
                        def inst_bt(view, m=m, wkey=wkey):
                            AG._bf16_packs[wkey + (True,)] = (K.weight_key(m.weight), view, m.weight)
                        add(bf_maps, self._bf, bf16_map(off, cout, cin, True), inst_bt)
                if direct_fwd:
                    def inst_d(view, m=m):
                        m._pk, m._pkv = view, K.weight_key(m.weight)
                    add_f32(K.conv2d_pack_weight(bits(idx_like(m.weight))), inst_d)
                if ks == 1:
                    wt = bits(idx_like(m.weight).transpose(0, 1).flip(2, 3))

                    def inst_dt(view, m=m, wkey=wkey):
                        AG._dgrad_direct[wkey] = (K.weight_key(m.weight),
                                                  dict(packed=view, wt=m.weight.detach().transpose(0, 1)))
                    add_f32(K.conv2d_pack_weight(wt), inst_dt)
        self.f32_map = torch.cat(f32_maps) if f32_maps else None
        self.bf_map = torch.cat(bf_maps) if bf_maps else None
        self.f32_dst = torch.empty(self.f32_map.numel(), dtype=torch.float32, device=dev) if f32_maps else None
        self.bf_dst = torch.empty(self.bf_map.numel(), dtype=torch.int16, device=dev) if bf_maps else None
        self.run()

    def run(self):
        if self.f32_map is not None:
            K.gather_pack(self.flat.data, self.f32_map, self.f32_dst)
            for off, n, install in self._f32:
                install(self.f32_dst[off:off + n])
        if self.bf_map is not None:
            K.gather_pack(self.flat.data, self.bf_map, self.bf_dst)
            for off, n, install in self._bf:
                install(self.bf_dst[off:off + n])


def annealing_cos(start, end, pct):
    return end + (start - end) / 2 * (math.cos(math.pi * pct) + 1)


class OneCycle:
    """lr: lr_max/div -> lr_max over the first pct_start of the steps, then -> lr_max/div/1e4; momentum mirrors it
    between moms[0] and moms[1]."""

    def __init__(self, optimizer, total_step, lr_max, moms, div_factor, pct_start):
        self.optimizer, self.total_step = optimizer, total_step
        low = lr_max / div_factor
        cut = int(pct_start * total_step)
        self.lr_phases = [(0, cut, (low, lr_max)), (cut, total_step, (lr_max, low / 1e4))]
        self.mom_phases = [(0, cut, tuple(moms)), (cut, total_step, tuple(moms[::-1]))]
        optimizer.lr, optimizer.mom = low, moms[0]

    def values(self, step):
        lr, mom = self.optimizer.lr, self.optimizer.mom
        for (s, e, (a, b)), (_, _, (ma, mb)) in zip(self.lr_phases, self.mom_phases):
            if step >= s and e > s:                  # (a phase of zero length -- tiny total_step -- is skipped)
                lr, mom = annealing_cos(a, b, (step - s) / (e - s)), annealing_cos(ma, mb, (step - s) / (e - s))
        return lr, mom

    def step(self, step):
        self.optimizer.lr, self.optimizer.mom = self.values(step)


class CosineWarmupLR:
    """tools/train_utils/optimization/learning_schedules_fastai.py:78-87: lr = eta_min + (base_lr - eta_min) *
    (1 - cos(pi * it / T_max)) / 2 for the first T_max iterations (the runner switches to the main schedule after)."""

    def __init__(self, optimizer, T_max, eta_min=0.0):
        self.optimizer, self.T_max, self.eta_min = optimizer, int(T_max), float(eta_min)
        self.base_lr = optimizer.lr

    def step(self, step):
        self.optimizer.lr = self.eta_min + (self.base_lr - self.eta_min) * (1 - math.cos(math.pi * step / self.T_max)) / 2


class AdamOneCycle:
    def __init__(self, model, lr, weight_decay, beta2=0.99, eps=1e-8, grad_clip=None, world_size=1):
        self.flat = model if isinstance(model, FlatParams) else FlatParams(model)
        self.lr, self.mom, self.wd, self.beta2, self.eps = lr, 0.9, weight_decay, beta2, eps
        self.max_norm = float(grad_clip["max_norm"]) if grad_clip else 0.0
        if grad_clip and grad_clip.get("norm_type", 2) != 2:
            raise NotImplementedError("only the L2 gradient-norm clip of the reference configs")
        self.world_size = world_size
        self.exp_avg = torch.zeros_like(self.flat.data)
        self.exp_avg_sq = torch.zeros_like(self.flat.data)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=self.flat.data.device)
        self.steps = 0
        self.pack_plan = None                 # PackPlan(model, self.flat): all weight images re-packed in one launch

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self):
        self.steps += 1
        grad = self.flat.grad                                 # gathers this step's gradients (one multi-tensor copy)
        sumsq = K.grad_sumsq(grad, self.sumsq) if self.max_norm > 0 else None
        K.adam_step(self.flat.data, grad, self.exp_avg, self.exp_avg_sq, sumsq, self.lr, self.mom,
                    self.beta2, self.eps, self.wd, self.steps, self.max_norm, 1.0 / self.world_size)
        K.bump_weights_generation()          # raw-pointer update: parameter `_version`s did not move
        if self.pack_plan is not None:
            self.pack_plan.run()

    def state_dict(self):
        return dict(exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(), steps=self.steps, lr=self.lr,
                    mom=self.mom)

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.steps, self.lr, self.mom = sd["steps"], sd["lr"], sd["mom"]
        K.bump_weights_generation()


def build_optimizer(model, optim_cfg, world_size=1):
    if optim_cfg["type"] != "adam_onecycle":
        raise NotImplementedError("the SA-SSD configs train with 'adam_onecycle'")
    opt = AdamOneCycle(model, optim_cfg["lr"], optim_cfg["weight_decay"], grad_clip=optim_cfg.get("grad_clip"),
                       world_size=world_size)
    if opt.flat.data.is_cuda and optim_cfg.get("pack_plan", True):
        opt.pack_plan = PackPlan(model, opt.flat)
        opt.flat.pack_plan = opt.pack_plan        # GradSync re-runs it after the parameter broadcast
    return opt


def build_scheduler(optimizer, total_iters_each_epoch, total_epochs, optim_cfg, lr_cfg):
    if lr_cfg["policy"] != "onecycle":
        raise NotImplementedError("the SA-SSD configs use the 'onecycle' policy")
    return OneCycle(optimizer, total_iters_each_epoch * total_epochs, optim_cfg["lr"], list(lr_cfg["moms"]),
                    lr_cfg["div_factor"], lr_cfg["pct_start"])


def build_warmup_scheduler(optimizer, optim_cfg, lr_cfg):
    """optimization/__init__.py:57-62: a CosineWarmupLR when the lr config carries 'warmup', else None."""
    if "warmup" not in lr_cfg:
        return None
    return CosineWarmupLR(optimizer, T_max=lr_cfg["warmup_iters"], eta_min=optim_cfg["lr"] * lr_cfg["warmup_ratio"])


class GradSync:
    """Data-parallel exchange over the flat buffers (RCCL over xGMI with backend 'nccl', gloo in CPU tests).

    Parameters are broadcast once from rank 0 (every cached weight image is invalidated and the optimizer's PackPlan
    re-run afterwards: the broadcast writes the flat buffer without moving any autograd version).  Gradients: the flat
    gradient buffer is cut into `buckets` contiguous ranges on parameter boundaries; a post-accumulate hook on every
    parameter counts its bucket down, and buckets are launched as asynchronous all-reduces IN A FIXED ORDER -- last
    bucket first, the order in which backward completes them -- a bucket only once it is complete AND every bucket before
    it in that order has been launched.  RCCL / gloo match collectives by issue order per communicator, so the order
    must not depend on which parameters happened to receive a gradient on a rank (an empty sparse level skips its
    BatchNorm; a sample without boxes skips a head).  `all_reduce_grads()` launches what is still pending in the same
    order and waits.  The exchange of the layers whose backward ran first (heads, BEV stack) overlaps the backward of
    the sparse backbone.  With one bucket, or `overlap=False`, it is the single whole-model all-reduce of SURVEY 8e.
    The sum is left in the buffer; the 1/world mean is folded into the update kernel.  `comm_ms()` reports the time of
    the last exchange between the first launch and the last completion (stream events; CUDA/HIP only)."""

    def __init__(self, flat, buckets=4, overlap=True, time_comm=False, force=False):
        self.flat = flat
        # `force`: run the exchange on a one-rank communicator too (a sum over one rank; every collective still goes
        # through the backend on the device stream -- the only way the RCCL path executes on a single-GPU box)
        self.on = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or bool(force))
        self.ranges, self.works, self._hooks = [], [], []
        self.time_comm, self._ev = bool(time_comm), None
        if not self.on:
            return
        dist.broadcast(flat.data, src=0)
        K.bump_weights_generation()           # raw write into every parameter: no packed image may survive it
        plan = getattr(flat, "pack_plan", None)
        if plan is not None:
            plan.run()
        n = len(flat.params)
        buckets = max(1, min(int(buckets), n))
        target = flat.numel / buckets
        # contiguous parameter runs of ~equal size: [(first param, last param + 1, lo float, hi float)]
        start = 0
        for i in range(n):
            end_f = flat.offsets[i + 1] if i + 1 < n else flat.numel
            if end_f - flat.offsets[start] >= target or i == n - 1:
                self.ranges.append((start, i + 1, flat.offsets[start], end_f))
                start = i + 1
        self.overlap = bool(overlap) and len(self.ranges) > 1
        self.order = list(range(len(self.ranges) - 1, -1, -1))       # backward reaches the last parameters first
        self._reset_counters()
        if self.overlap:
            owner = {}
            for b, (lo, hi, _, _) in enumerate(self.ranges):
                for i in range(lo, hi):
                    owner[i] = b
            for i, p in enumerate(flat.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(owner[i])))

    def _reset_counters(self):
        self._left = [hi - lo for lo, hi, _, _ in self.ranges]
        self._launched = [False] * len(self.ranges)
        self._next = 0

    def reset(self):
        """Start of a step: finish whatever an aborted step left in flight, forget its bookkeeping."""
        if not self.on:
            return
        for w in self.works:
            w.wait()
        self.works = []
        self._reset_counters()

    def _make_hook(self, b):
        def hook(_param):
            self._left[b] -= 1
            self._advance()
        return hook

    def _advance(self, force=False):
        """Launch, in the fixed order, every bucket that is complete (all of them with `force`)."""
        while self._next < len(self.order):
            b = self.order[self._next]
            if not force and self._left[b] > 0:
                return
            self._launch(b)
            self._next += 1

    def _launch(self, b):
        if not self._launched[b]:
            plo, phi, lo, hi = self.ranges[b]
            self.flat.collect(plo, phi)                       # this bucket's gradients -> the flat buffer
            if self.time_comm and not self.works and self.flat._grad.is_cuda:
                self._ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
                self._ev[0].record()
            self.works.append(dist.all_reduce(self.flat._grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            self._launched[b] = True

    def all_reduce_grads(self):
        if not self.on:
            return
        self._advance(force=True)
        for w in self.works:
            w.wait()
        if self._ev is not None:
            self._ev[1].record()
        self.works = []
        self._reset_counters()

    def comm_ms(self):
        """First bucket launch -> all buckets complete on the compute stream, of the last timed exchange (None if
        timing is off / not yet run).  Synchronises on the end event."""
        if self._ev is None:
            return None
        self._ev[1].synchronize()
        return self._ev[0].elapsed_time(self._ev[1])


def _record_stream(obj, stream):
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)
    elif hasattr(obj, "buf") and torch.is_tensor(getattr(obj, "buf")):        # kernels.HashTable
        _record_stream(obj.buf, stream)


class SideStreamPrefetch:
    """Builds the next batch (device voxelizer, anchor masks, rulebooks: kernels plus a few host reads of row counts)
    on its own HIP stream.  The host reads then wait only for the data-preparation kernels, not for the training
    step queued on the main stream -- with everything on one stream each `.item()` drained the whole step and the GPU
    idled while the host prepared the batch and began queueing the backward pass.  The main stream waits for the side
    stream's event before it touches the batch; every tensor of the batch is recorded on the main stream so that the
    caching allocator does not hand its memory to the next side-stream batch while main-stream kernels still read it.
    Every build -- the first one included -- should go through the prefetcher: the cached kernel workspaces of the
    builder then belong to this object's workspace scope, are allocated under the side stream and never touched from
    another stream; the first build waits once for the main stream (inputs uploaded there)."""

    def __init__(self, build):
        self.build = build
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self._first = True

    def __call__(self, *args, **kw):
        if self.stream is None:
            return self.build(*args, **kw)
        main = torch.cuda.current_stream()
        if self._first:
            self.stream.wait_stream(main)
            self._first = False
        with torch.cuda.stream(self.stream), K.ws_scope(("prefetch", id(self))):
            batch = self.build(*args, **kw)
            done = self.stream.record_event()
        main.wait_event(done)
        _record_stream(batch, main)
        return batch


def parse_losses(losses):
    """train_utils/__init__.py:8-25 without the per-term .item() host syncs: (total loss tensor, detached terms)."""
    one = lambda t: t.reshape(()) if t.numel() == 1 else t.mean()       # noqa: E731  (a [1] tensor needs no reduction)
    terms = {k: (one(v) if torch.is_tensor(v) else sum(one(x) for x in v)) for k, v in losses.items()}
    parts = [v for k, v in terms.items() if "loss" in k]
    total = torch.stack(parts).sum() if len(parts) > 1 and all(torch.is_tensor(p) for p in parts) else sum(parts)
    return total, {k: v.detach() for k, v in terms.items()}


def train_one_iter(model, optimizer, scheduler, sync, batch, it, prefetch=None):
    """One iteration of train_one_epoch (train_utils/__init__.py:39-61).  `prefetch()` (optional) builds the NEXT batch
    between the forward and the backward pass: its host syncs (voxel / rulebook row counts) then wait only for this
    step's forward, and the next forward's launches queue up behind this step's GPU-bound backward instead of
    waiting for it.  Returns (loss, loss terms[, next batch])."""
    scheduler.step(it)
    model.train()
    if hasattr(sync, "reset"):
        sync.reset()
    optimizer.zero_grad()
    loss, terms = parse_losses(model(**batch))
    nxt = prefetch() if prefetch is not None else None
    if hasattr(getattr(model, "rpn_head", None), "poll_guided_capacity"):
        model.rpn_head.poll_guided_capacity()        # non-blocking: reports an overflow a step or two late
    loss.backward()
    sync.all_reduce_grads()
    optimizer.step()
    return (loss.detach(), terms) if prefetch is None else (loss.detach(), terms, nxt)


def checkpoint_state(model, optimizer, epoch, it):
    return dict(epoch=epoch, it=it, version="sassd", model_state={k: v.detach().cpu() for k, v in model.state_dict().items()},
                optimizer_state={k: (v.cpu() if torch.is_tensor(v) else v) for k, v in optimizer.state_dict().items()})


def save_checkpoint(state, filename):
    torch.save(state, filename if filename.endswith(".pth") else filename + ".pth")


def _copy_model_state(model, model_state):
    """In-place copy (parameters stay views of the flat buffer) of every tensor whose name -- with or without the
    (MM)DataParallel 'module.' prefix of the reference's checkpoints -- and shape match.  -> (loaded, skipped)."""
    sd, loaded, skipped = model.state_dict(), [], []
    with torch.no_grad():
        for key, val in model_state.items():
            k = key[len('module.'):] if key.startswith('module.') and key not in sd else key
            if k in sd and sd[k].shape == val.shape:
                sd[k].copy_(val)
                loaded.append(k)
            else:
                skipped.append(key)
    K.bump_weights_generation()
    return loaded, skipped


def load_checkpoint(model, optimizer, filename, map_location="cpu"):
    """tools/train_utils/__init__.py:120-150 (resume): model + optimizer state.  A reference checkpoint carries a torch
    Adam / OptimWrapper state dict, which has no meaning for the flat fused optimizer: it is skipped with a warning."""
    import warnings
    ck = torch.load(filename, map_location=map_location, weights_only=False)
    _, skipped = _copy_model_state(model, ck["model_state"])
    if skipped:
        warnings.warn("load_checkpoint: %d tensors of %s do not match the model: %s" % (len(skipped), filename,
                                                                                        skipped[:4]))
    ost = ck.get("optimizer_state")
    if optimizer is not None and ost:
        if isinstance(ost, dict) and "exp_avg" in ost and "exp_avg_sq" in ost:
            optimizer.load_state_dict(ost)
        else:
            warnings.warn("load_checkpoint: optimizer state of %s is not in the flat exp_avg / exp_avg_sq format "
                          "(a reference torch optimizer state?) -- optimizer starts fresh" % filename)
    return ck.get("epoch", 0), ck.get("it", 0)


def load_params_from_file(model, filename, to_cpu=False):
    """tools/train_utils/__init__.py:152-173: copy every tensor of the checkpoint's 'model_state' whose name and shape match
    into the model (in place, so FlatParams views stay valid).  Checkpoints written by the reference's (MM)DataParallel
    wrappers carry a 'module.' prefix on every key; it is accepted with or without.  -> (loaded, skipped) key lists."""
    import os
    if not os.path.isfile(filename):
        raise FileNotFoundError(filename)
    ck = torch.load(filename, map_location=torch.device('cpu') if to_cpu or not torch.cuda.is_available() else None,
                    weights_only=False)
    return _copy_model_state(model, ck['model_state'])


_ANCHORS_B = {}


def _batched_anchors(an, B):
    """[A, 7] anchors -> contiguous [B, A, 7], cached per (anchor tensor, B): anchors are constants of a training run"""
    key = (an.data_ptr(), an._version, tuple(an.shape), int(B))
    hit = _ANCHORS_B.get(key)
    if hit is None:
        if len(_ANCHORS_B) > 16:
            _ANCHORS_B.clear()
        hit = _ANCHORS_B[key] = (an, an.unsqueeze(0).repeat(B, 1, 1).contiguous())    # (pins `an`: its address stays its own)
    return hit[1]


def device_batch(points, gt_bboxes, gt_types, class_names, anchors, anchors_bv, voxel_size, pc_range,
                 max_points=5, max_voxels=20000, area_threshold=1, model=None, level_cap_factor=2):
    """What KittiLiDAR.prepare_train_img + collate produce (kitti.py:212-262,333-343), built on the device from raw
    points already in HBM: HIP voxelizer + HIP anchor mask.  points: list of [N,4] device tensors; gt_bboxes: list of
    [G,7] device tensors; anchors: {class: [A,7] device}; anchors_bv: {class: [A,4] device}.
    With `model`, the seven sparse-conv rulebooks of the batch are built here too (key 'sassd_rulebooks'), so the
    forward pass contains no host synchronisation before its guided-anchor selection.
    Returns the keyword arguments of SingleStageDetector.forward(return_loss=True).

    Two host synchronisations per batch, whatever its size: every sample is voxelized into ONE coordinate / payload buffer
    through the row-offset chain of sassd_voxelize (then one read of the B + 1 offsets: the reference API carries
    exact-size tensors), the anchor masks of all samples come from one sassd_anchor_mask_batch sequence per class, and
    the rulebooks from the fused pyramid (sassd_rulebook_pyramid, 11 launches, capacity-sized tables) followed by one
    read of the three down-sampled row counts + the status word.  The merged batch tensors ride along under
    'sassd_merged' so that merge_second_batch does not concatenate the per-sample views again."""
    vs, cr = list(voxel_size), list(pc_range)
    w0 = int(round((cr[3] - cr[0]) / vs[0]))
    h0 = int(round((cr[4] - cr[1]) / vs[1]))
    B = len(points)
    dev = points[0].device
    ndim = points[0].shape[1]
    cap0 = B * int(max_voxels)
    kw = dict(img=None, img_meta=[dict(sample_idx=i) for i in range(B)], return_loss=True, voxels=[],
              coordinates=[], num_points=[], anchors={c: [] for c in class_names},
              anchors_mask={c: [] for c in class_names}, gt_bboxes=list(gt_bboxes), gt_labels=[],
              gt_types=list(gt_types))
    voxels = torch.empty(cap0, max_points, ndim, dtype=torch.float32, device=dev)
    coors4 = torch.empty(cap0, 4, dtype=torch.int32, device=dev)
    nump = torch.empty(cap0, dtype=torch.int32, device=dev)
    # one zero fill for every small device counter of the batch: [row offsets B+1 | voxel counts B | level row counts 3 | status 1]
    # (the last four are contiguous: host sync 2 reads them as one slice, without a cat)
    small = torch.zeros(2 * B + 5, dtype=torch.int32, device=dev)
    row_off, vnum, n_dev, status = small[:B + 1], small[B + 1:2 * B + 1], small[2 * B + 1:2 * B + 4], small[2 * B + 4:]
    for b, p in enumerate(points):
        K.voxelize(p, vs, cr, max_points, max_voxels, batch_idx=b, coors_cols=4, want_voxels=True, want_mean=False,
                   out=dict(voxels=voxels, coors=coors4, num_points=nump, voxel_num=vnum[b:b + 1]),
                   row_offset=row_off[b:b + 2], status=status, cap=cap0)
    masks = {}
    for c in class_names:                                  # coordinate-only work, queued before the first host read
        masks[c] = torch.empty(B, anchors_bv[c].shape[0], dtype=torch.uint8, device=dev)
        K.anchor_mask_batch(coors4, row_off, B, h0, w0, anchors_bv[c], vs, cr, area_threshold, masks[c])
    offs = row_off.cpu().numpy()                           # host sync 1: the per-sample row ranges
    n0 = int(offs[B])
    for b in range(B):
        lo, hi = int(offs[b]), int(offs[b + 1])
        kw["voxels"].append(voxels[lo:hi])
        kw["coordinates"].append(coors4[lo:hi, 1:])
        kw["num_points"].append(nump[lo:hi])
        for c in class_names:
            kw["anchors"][c].append(anchors[c])
            kw["anchors_mask"][c].append(masks[c][b].view(torch.bool))     # (the kernel writes 0 / 1 bytes: a view, no launch)
        names = list(class_names)
        lab = [names.index(t) + 1 if t in names else 0 for t in gt_types[b]]
        kw["gt_labels"].append(torch.tensor(lab, dtype=torch.int64, device=dev))
    # the batch as merge_second_batch would build it, without its launches: voxel buffers are already one tensor; the anchors
    # of every sample are the same tensor (one cached [B, A, 7] copy per anchor set instead of a stack per step); the masks
    # are already [B, A]
    kw["sassd_merged"] = dict(voxels=voxels[:n0], num_points=nump[:n0], coordinates=coors4[:n0],
                              anchors={c: _batched_anchors(anchors[c], B) for c in class_names},
                              anchors_mask={c: masks[c].view(torch.bool) for c in class_names})
    if model is not None:
        shape0 = [int(v) for v in model.neck.sparse_shape]

        def build_pyramid(factor):
            # a strided level has at most 8 x the rows of the level above (isolated voxels); LiDAR frames sit near 1.1 x,
            # tiny or very sparse clouds well above 2 x: capacity = factor x n0 with a floor, never more than 8 x the level above
            caps = [max(n0, 1)]
            for _ in range(3):
                caps.append(max(min(8 * caps[-1], max(n0 * factor, n0 + 16384)), 1))
            idx = [coors4[:max(n0, 1)]] + [torch.empty(c, 4, dtype=torch.int32, device=dev) for c in caps[1:]]
            n_ptrs = [row_off[B:B + 1]] + [n_dev[i:i + 1] for i in range(3)]
            nbr_s = [torch.empty(c, 27, dtype=torch.int32, device=dev) for c in caps]
            nbr_d = [None] + [torch.empty(c, 27, dtype=torch.int32, device=dev) for c in caps[1:]]
            pyr = K.RulebookPyramid(idx, n_ptrs, caps, shape0, B, nbr_s, nbr_d, status)
            pyr.build()
            return idx, nbr_s, nbr_d, pyr, small[2 * B + 1:].cpu().numpy()            # host sync 2: row counts + flags

        idx, nbr_s, nbr_d, pyr, tail = build_pyramid(level_cap_factor)
        if int(tail[3]) != 0 and level_cap_factor < 8:
            small[2 * B + 1:].zero_()                      # capacity overflow: once more with the worst-case bound
            idx, nbr_s, nbr_d, pyr, tail = build_pyramid(8)
        if int(tail[3]) != 0:
            raise RuntimeError("device_batch: status 0x%x (voxel / rulebook capacity overflow: raise max_voxels)"
                               % int(tail[3]))
        n = [n0] + [int(v) for v in tail[:3]]
        shapes = [shape0]
        for _ in range(3):
            shapes.append(list(K.conv_out_shape(shapes[-1])))
        books = {}
        for l in range(4):
            il = idx[l][:n[l]]
            books["subm%d" % l] = (il, nbr_s[l][:max(n[l], 1)], shapes[l], None)
            if l > 0:
                books["down%d" % (l - 1)] = (il, nbr_d[l][:max(n[l], 1)], shapes[l], None)
        books["_keep"] = pyr                              # workspace + argument block live as long as the rulebooks
        kw["sassd_rulebooks"] = books
    return kw
