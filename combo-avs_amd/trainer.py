"""Data-parallel training step for the COMBO hot path on one MI355X node.

One process per GPU.  All trainable parameters live in ONE flat fp32 buffer and all gradients in ONE flat fp32
buffer (parameters / .grad are views), so a step issues exactly one RCCL all-reduce over xGMI (plus the 4-byte
`num_masks` all-reduce inside the criterion, criterion.py:263-265 of the reference), one global-norm clip and one
AdamW pass.  Semantics follow train_net.py:147-226 of the reference:
  * AdamW(lr = BASE_LR, weight_decay = WEIGHT_DECAY), lr x BACKBONE_MULTIPLIER for parameters of modules whose
    name contains "backbone" (this matches `pre_sam_backbone` too, train_net.py:182),
  * weight_decay = WEIGHT_DECAY_NORM (0) for norm-layer parameters, WEIGHT_DECAY_EMBED (0) for nn.Embedding,
  * full-model gradient-norm clipping to CLIP_VALUE (0.01) before the update (train_net.py:205-211),
  * gradients are averaged over ranks (DDP semantics).
Parameters are laid out grouped by (lr, weight_decay) so every group is one contiguous segment.
"""
import math

import os

import torch
import torch.distributed as dist
from torch import nn

NORM_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm, nn.GroupNorm, nn.InstanceNorm1d,
              nn.InstanceNorm2d, nn.InstanceNorm3d, nn.LayerNorm, nn.LocalResponseNorm)


def param_groups(model, base_lr, weight_decay, backbone_multiplier=0.1, weight_decay_norm=0.0, weight_decay_embed=0.0):
    """-> list of (param, name, lr, wd) following train_net.py:170-194 of the reference."""
    out, memo = [], set()
    for module_name, module in model.named_modules():
        for pname, p in module.named_parameters(recurse=False):
            if not p.requires_grad or p in memo:
                continue
            memo.add(p)
            lr, wd = base_lr, weight_decay
            if "backbone" in module_name:
                lr = lr * backbone_multiplier
            if "relative_position_bias_table" in pname or "absolute_pos_embed" in pname:
                wd = 0.0
            if isinstance(module, NORM_TYPES):
                wd = weight_decay_norm
            if isinstance(module, nn.Embedding):
                wd = weight_decay_embed
            out.append((p, f"{module_name}.{pname}" if module_name else pname, lr, wd))
    return out


class FlatAdamW:
    """Flat-buffer AdamW with full-model grad-norm clipping and a single gradient all-reduce."""

    def __init__(self, model, base_lr=1e-4, weight_decay=0.05, backbone_multiplier=0.1, clip_value=0.01,
                 betas=(0.9, 0.999), eps=1e-8, weight_decay_norm=0.0, weight_decay_embed=0.0, grad_dtype=torch.float32,
                 grad_comm_dtype=torch.float32, early=None):
        """early(name) -> True for the parameters whose gradients are COMPLETE once the backward pass has reached the head's
        inputs (everything downstream of the cut: `sem_seg_head.*`): they form the front region of the flat buffers, so that
        their all-reduce can start while the rest of the backward pass (the backbones) still runs - see backward_early /
        backward_late / all_reduce_grads(region).  None: one region (the whole buffer)."""
        entries = param_groups(model, base_lr, weight_decay, backbone_multiplier, weight_decay_norm, weight_decay_embed)
        is_early = (lambda n: True) if early is None else early
        ref_index = {id(e[0]): i for i, e in enumerate(entries)}  # position in the reference's optimizer (train_net.py:170-194)
        # stable: contiguous (lr, wd) segments; inside a segment the head's matrix weights come first - their gradients are
        # written in place by the grouped weight-gradient launches (ops.linear.grad_targets), the rest forms few long runs
        # for the concatenation that fills the flat gradient buffer
        entries.sort(key=lambda e: (0 if is_early(e[1]) else 1, e[2], e[3], 0 if (e[0].dim() == 2 and "sem_seg_head" in e[1]) else 1))
        self.n_early = sum(1 for e in entries if is_early(e[1]))
        self.ref_order = [ref_index[id(e[0])] for e in entries]
        self.entries = entries
        # every (lr, wd) segment starts on a 16-byte boundary (vectorised fused AdamW kernel)
        # ... and so does every parameter (the dense-layer kernels read weights with 16-byte loads)
        total = sum((p.numel() + 3) // 4 * 4 for p, _, _, _ in entries)
        dev = entries[0][0].device
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=grad_dtype, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.segments = []  # (start, end, lr, wd)
        self.grad_views = []
        self.offsets = []
        self.params = [e[0] for e in entries]
        off = 0
        self.flat_param.zero_()
        for p, name, lr, wd in entries:
            n = p.numel()
            off = (off + 3) // 4 * 4
            self.flat_param[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + n].view_as(p)
            self.grad_views.append(self.flat_grad[off:off + n].view_as(p))
            self.offsets.append(off)
            if self.segments and self.segments[-1][2] == lr and self.segments[-1][3] == wd:
                self.segments[-1][1] = off + n
            else:
                self.segments.append([off, off + n, lr, wd])
            off += n
            if len(self.offsets) == self.n_early:
                self.split = (off + 3) // 4 * 4  # end of the early region (16-byte aligned) = start of the late one
        # optionally transport the gradient in bf16 (half the xGMI bytes), fp32 master + fp32 optimiser
        self.grad_comm_dtype = grad_comm_dtype
        self.betas, self.eps, self.clip_value = betas, eps, clip_value
        self.step_count = 0
        self.lr_scale = 1.0  # WarmupPolyLR factor, set by the caller each iteration
        self.numel = total
        if self.n_early == 0:
            self.split = 0
        if self.n_early == len(entries):
            self.split = total
        self._comm_stream = None
        self.comm_events = None  # a list: all_reduce_grads appends (start, end) events around every collective (bench.py)
        self.unused = ()  # indices (into self.params) of parameters the last backward pass produced no gradient for

    def zero_grad(self):
        self.flat_grad.zero_()

    def backward(self, loss):
        """d loss / d params straight into the flat gradient buffer: `autograd.grad` (no per-parameter AccumulateGrad
        add kernels - 530 launches/step for this model); weight gradients computed by the head's own kernels are written in
        place (ops.linear.grad_targets), every maximal run of the others is filled by one concatenation."""
        grads = self._grad(loss, self.params, None)
        self.unused = tuple(i for i, g in enumerate(grads) if g is None)
        self._collect(grads, 0, len(self.params))

    def backward_early(self, loss, cut):
        """First half of a backward pass cut at the tensors `cut` (the head's inputs, meta_arch.MaskFormer._head_inputs):
        gradients of the EARLY parameters (complete now: their region may be all-reduced) -> returns d loss / d cut."""
        cut = [t for t in cut if t.requires_grad]
        n = self.n_early
        grads = self._grad(loss, self.params[:n] + cut, None)
        self._unused_early = tuple(i for i, g in enumerate(grads[:n]) if g is None)
        self._collect(grads[:n], 0, n)
        self._cut = cut
        return list(grads[n:])

    def backward_late(self, cut_grads):
        """Second half: the parameters upstream of the cut from the gradients backward_early returned."""
        n = self.n_early
        pairs = [(t, g) for t, g in zip(self._cut, cut_grads) if g is not None]
        grads = self._grad([t for t, _ in pairs], self.params[n:], [g for _, g in pairs]) if pairs else [None] * (len(self.params) - n)
        self.unused = self._unused_early + tuple(n + i for i, g in enumerate(grads) if g is None)
        self._collect(grads, n, len(self.params))
        self._cut = None
        del pairs, grads

    def _grad(self, outputs, inputs, grad_outputs):
        from .ops.linear import deferred_dw, grad_targets
        in_place = self.flat_grad.dtype == torch.float32 and self.flat_grad.is_cuda
        # matrices (dense layers) and 4-D convolution weights (1x1 layers run as dense layers; the backbones' FrozenBN-folded
        # weights resolve to their parameters through ops.linear.register_grad_aliases)
        targets = {p.data_ptr(): v for p, v in zip(self.params, self.grad_views) if p.dim() in (2, 4)} if in_place else {}
        with grad_targets(targets), deferred_dw():  # latency-bound decoder weight gradients: one grouped launch when the context closes
            return torch.autograd.grad(outputs, inputs, grad_outputs, allow_unused=True)

    def _collect(self, grads, lo, hi):
        """gradients of self.params[lo:hi] -> their range of the flat buffer"""
        # torch.optim.AdamW (the reference's optimizer) skips parameters whose gradient is None - no decay, no moment update:
        # `step` leaves them out (a static property of the model / recipe, so it is safe inside a captured graph)
        start = self.offsets[lo] if lo < len(self.offsets) else self.numel
        if lo > 0:
            start = self.split
        end = self.split if hi == self.n_early and hi < len(self.params) else (self.numel if hi == len(self.params) else self.offsets[hi])
        if hi <= lo:
            return
        if self.flat_grad.dtype != torch.float32 or any(g is not None and g.dtype != torch.float32 for g in grads):
            dst = [v for v, g in zip(self.grad_views[lo:hi], grads) if g is not None]
            src = [g for g in grads if g is not None]
            if len(src) != len(grads):
                self.flat_grad[start:end].zero_()
            torch._foreach_copy_(dst, src)
            return
        # gradients that were written in place (they ARE their flat-buffer views) stay; every maximal run of the others is
        # filled by one concatenation (alignment gaps and parameters without a gradient: cached zeros)
        pieces, run_start, cursor = [], start, start

        def flush(end_):
            nonlocal pieces, run_start
            if pieces:
                torch.cat(pieces, out=self.flat_grad[run_start:end_])
            pieces, run_start = [], end_
        for g, p, off, view in zip(grads, self.params[lo:hi], self.offsets[lo:hi], self.grad_views[lo:hi]):
            n = p.numel()
            if off > cursor:
                pieces.append(self._zeros(off - cursor))
            if g is not None and g.data_ptr() == view.data_ptr() and g.is_contiguous():
                flush(off)
                run_start = off + n
            else:
                pieces.append(g.reshape(-1) if g is not None else self._zeros(n))
            cursor = off + n
        if end > cursor:
            pieces.append(self._zeros(end - cursor))
        flush(end)

    def _zeros(self, n):
        """cached zero filler (alignment gaps of the flat layout, parameters without a gradient)"""
        cache = self.__dict__.setdefault("_zero_cache", {})
        if n not in cache:
            cache[n] = torch.zeros(n, dtype=torch.float32, device=self.flat_grad.device)
        return cache[n]

    def _dist_on(self):
        # COMBO_FORCE_PG=1 (bench.py's one-rank process group on a 1-GPU box) also runs the collective at world size 1,
        # so that the RCCL all-reduce of the flat gradient buffer is exercised on the GPU
        forced = os.environ.get("COMBO_FORCE_PG") == "1"
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced)

    def all_reduce_grads(self, region=None, overlap=False):
        """Sum over ranks / world (DDP semantics).  region: None = the whole flat buffer in ONE collective (RCCL over xGMI on GPU;
        gloo in the CPU tests), "early" / "late" = the two regions of a cut backward pass.  overlap: issue the collective on a
        side stream that waits for the work enqueued so far (the caller goes on enqueueing the late backward on the main
        stream) - finish with wait_comm()."""
        if not self._dist_on():
            return
        lo, hi = {None: (0, self.numel), "early": (0, self.split), "late": (self.split, self.numel)}[region]
        if hi <= lo:
            return
        world = dist.get_world_size()
        g = self.flat_grad[lo:hi]

        def run():
            ev = None
            if self.comm_events is not None and g.is_cuda:  # bench.py: events on the stream the collective runs on
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if self.grad_comm_dtype != g.dtype:
                buf = (g / world).to(self.grad_comm_dtype)  # pre-divide: keeps the sum inside bf16's range
                dist.all_reduce(buf)
                g.copy_(buf)
            else:
                dist.all_reduce(g)
                g.div_(world)
            if ev is not None:
                ev[1].record()
                self.comm_events.append(ev)
        if overlap and g.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=g.device)
            self._comm_stream.wait_stream(torch.cuda.current_stream(g.device))
            with torch.cuda.stream(self._comm_stream):
                run()
        else:
            run()

    def wait_comm(self):
        """the main stream waits for the collectives issued with overlap=True"""
        if self._comm_stream is not None:
            torch.cuda.current_stream(self.flat_grad.device).wait_stream(self._comm_stream)

    @torch.no_grad()
    def step(self):
        from .ops import optim
        self.step_count += 1
        g = self.flat_grad
        if self.clip_value and self.clip_value > 0:
            total_norm = torch.linalg.vector_norm(g.float())  # clip_grad_norm_(all params, 0.01)
            clip_coef = torch.clamp(self.clip_value / (total_norm + 1e-6), max=1.0)
        else:
            clip_coef = torch.ones((), device=g.device)
        b1, b2 = self.betas
        bc1 = 1 - b1 ** self.step_count
        bc2 = 1 - b2 ** self.step_count
        for s, e, lr, wd in self._active_segments():
            optim.adamw_segment(self.flat_param[s:e], g[s:e], self.exp_avg[s:e], self.exp_avg_sq[s:e], clip_coef,
                                lr * self.lr_scale, wd, b1, b2, self.eps, bc1, bc2)

    def _active_segments(self):
        """the (lr, wd) segments with the ranges of the unused parameters cut out (every parameter starts 16-byte aligned)"""
        if not self.unused:
            return self.segments
        cache = self.__dict__.setdefault("_active_cache", {})
        if self.unused not in cache:
            holes = sorted((self.offsets[i], self.offsets[i] + self.params[i].numel()) for i in self.unused)
            out = []
            for s, e, lr, wd in self.segments:
                cur = s
                for hs, he in holes:
                    if he <= cur or hs >= e:
                        continue
                    if hs > cur:
                        out.append([cur, hs, lr, wd])
                    cur = max(cur, (he + 3) // 4 * 4)
                if cur < e:
                    out.append([cur, e, lr, wd])
            cache[self.unused] = out
        return cache[self.unused]

    def state_dict(self):
        """torch.optim.AdamW's layout over the reference's parameter order (one group per parameter, train_net.py:170-194):
        what detectron2's checkpointer stores as "optimizer"; plus the flat buffers under "flat"."""
        state, groups = {}, [None] * len(self.params)
        for j, (p, off) in enumerate(zip(self.params, self.offsets)):
            i, n = self.ref_order[j], p.numel()
            _, name, lr, wd = self.entries[j]
            groups[i] = {"lr": lr * self.lr_scale, "initial_lr": lr, "betas": self.betas, "eps": self.eps, "weight_decay": wd,
                         "amsgrad": False, "params": [i]}
            if self.step_count > 0 and j not in self.unused:
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.exp_avg[off:off + n].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(p).clone()}
        return {"state": state, "param_groups": groups,
                "flat": {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}}

    def load_state_dict(self, sd):
        """accepts torch.optim.AdamW.state_dict() of the reference's optimizer (same parameter order) or this class's own"""
        if "flat" in sd and sd["flat"]["exp_avg"].numel() == self.numel:
            self.exp_avg.copy_(sd["flat"]["exp_avg"])
            self.exp_avg_sq.copy_(sd["flat"]["exp_avg_sq"])
            self.step_count = int(sd["flat"]["step"])
            return
        if len(sd["param_groups"]) != len(self.params):
            raise ValueError(f"optimizer state for {len(sd['param_groups'])} parameter groups, this model has {len(self.params)}")
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for j, (p, off) in enumerate(zip(self.params, self.offsets)):
            st = sd["state"].get(self.ref_order[j])
            if st is None:
                continue
            n = p.numel()
            if st["exp_avg"].numel() != n:
                raise ValueError(f"optimizer state of parameter {self.entries[j][1]}: {tuple(st['exp_avg'].shape)} vs {tuple(p.shape)}")
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not representable in the flat optimizer")
        self.step_count = steps.pop() if steps else 0


def poly_lr_factor(it, max_iter, power=0.9, constant_ending=0.0):
    """detectron2 projects/deeplab WarmupPolyLR with WARMUP_ITERS = 0 (train_net.py:140-145)."""
    f = math.pow(1.0 - it / max_iter, power)
    return f if f >= constant_ending or constant_ending <= 0 else constant_ending


def train_step(model, optimizer, batched_inputs):
    """forward -> 39-term loss -> backward -> one all-reduce -> clip + AdamW.  Returns the loss dict (device tensors)."""
    from .ops.linear import grouped_presplit
    model.record_head_inputs = bool(0 < optimizer.n_early < len(optimizer.params) and optimizer._dist_on())
    with grouped_presplit():  # one grouped weight pre-split for all input-gradient GEMMs of the step
        loss_dict = model(batched_inputs)
        total = getattr(loss_dict, "total", None)  # family-wise sum from the meta-arch (modeling.criterion.LossDict)
        if total is None:
            total = torch.stack(list(loss_dict.values())).sum()
        cut = getattr(model, "_head_inputs", None)
        model._head_inputs = None
        if cut and 0 < optimizer.n_early < len(optimizer.params) and optimizer._dist_on():
            # data parallel: the head's gradients travel (side stream) while the backbones' backward runs
            cut_grads = optimizer.backward_early(total, cut)
            optimizer.all_reduce_grads("early", overlap=True)
            optimizer.backward_late(cut_grads)
            optimizer.all_reduce_grads("late")
            optimizer.wait_comm()
        else:
            optimizer.backward(total)
            optimizer.all_reduce_grads()
    optimizer.step()
    # detached: a caller holding the losses must not keep this step's autograd graph (and the parameters'
    # AccumulateGrad nodes, which remember the stream they were created on) alive into the next step
    return {k: v.detach() for k, v in loss_dict.items()}


def _input_leaves(batched_inputs):
    """Tensors of a batch in a fixed order + the hashable rest (heights, widths, flags given as python values)."""
    tensors, rest = [], []
    for b in batched_inputs:
        for k in sorted(b.keys()):
            v = b[k]
            if torch.is_tensor(v):
                tensors.append(v)
            elif k == "instances":
                for inst in v:
                    for name in ("gt_classes", "gt_masks"):
                        t = inst[name] if isinstance(inst, dict) else getattr(inst, name)
                        tensors.append(getattr(t, "tensor", t))
            else:
                rest.append((k, v))
    return tensors, tuple(rest)


def _clone_batch(batched_inputs):
    out = []
    for b in batched_inputs:
        nb = {}
        for k, v in b.items():
            if torch.is_tensor(v):
                nb[k] = v.clone()
            elif k == "instances":
                nb[k] = []
                for inst in v:
                    get = (lambda n, i=inst: i[n]) if isinstance(inst, dict) else (lambda n, i=inst: getattr(i, n))
                    nb[k].append({n: getattr(get(n), "tensor", get(n)).clone() for n in ("gt_classes", "gt_masks")})
            else:
                nb[k] = v
        out.append(nb)
    return out


def enable_library_gemm_tuning(max_ms_per_solution=20):
    """PyTorch TunableOp for the library GEMMs of the host-PyTorch backbones (the PVTv2 linears under bf16 autocast are small -
    7 840 x 320 x 1 280 in stage 3 - and hipBLASLt's default heuristic runs them at ~85 TFLOP/s).  Every new GEMM shape is timed
    over the available solutions the first time it is met, so the first (eager) step must see all shapes before a hipGraph is
    captured: GraphedTrainStep's eager warm-up does that.  Returns False when this PyTorch has no TunableOp."""
    try:
        from torch.cuda import tunable
    except ImportError:
        return False
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(int(max_ms_per_solution))
    import tempfile
    # the results file PyTorch writes at exit goes to the temp directory, not to the working directory
    tunable.set_filename(os.path.join(tempfile.gettempdir(), f"combo_tunableop_{os.getpid()}.csv"), True)
    return True


_memset_selftest = {}


def graph_memset_selftest(dev, replays=4):
    """Does a replayed hipGraph execute its memset nodes?  Captures a library reduction that zeroes its semaphores with
    hipMemsetAsync (ATen's sum of a [15680, 1280] bf16 matrix over the rows: one output split over several workgroups), replays
    it with an eager kernel between two synchronisations before each replay and compares with the eager result.  On the stack
    this was written for the check fails in 99 of 100 replays with the runtime's AQL packet capture on and never with it off
    (tools/graph_reduce_repro.py).  ~50 ms, once per process and device."""
    key = (dev.type, dev.index)
    if key in _memset_selftest:
        return _memset_selftest[key]
    with torch.no_grad():
        gen = torch.Generator(device=dev)
        gen.manual_seed(1234)
        x = torch.randn(15680, 1280, device=dev, generator=gen).to(torch.bfloat16)
        want = x.sum(0)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            x.sum(0)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            outs = [x.sum(0) for _ in range(4)]
        junk = torch.ones(64, device=dev)
        ok = True
        for _ in range(replays):
            torch.cuda.synchronize(dev)
            junk[:16].zero_()
            torch.cuda.synchronize(dev)
            g.replay()
            torch.cuda.synchronize(dev)
            ok = ok and all(torch.equal(o, want) for o in outs)
        del g, outs
    _memset_selftest[key] = ok
    return ok


class GraphedTrainStep:
    """`train_step` with forward + loss + backward replayed from ONE captured hipGraph.

    The eager step issues ~5 400 kernel launches (two ResNet-50s, 6+9 transformer layers, 39 losses) for ~70 ms of
    GPU work: the host is as busy as the GPU.  The device-side LSAP (csrc/lsap.hip) removed the step's last
    device->host dependency, so the whole forward/backward is a static launch sequence for a given input signature
    (tensor shapes + instances per frame) and is captured once per signature (`torch.cuda.CUDAGraph` = hipGraph on
    ROCm).  Per step the host then does: copy the batch into the static input buffers, one graph launch, the RCCL
    gradient all-reduce and the clip + AdamW launches (those stay eager: lr, bias corrections and the collective
    change per step / are not graph material).  Randomness stays fresh per replay: torch's Philox generator is
    graph-aware and the bilateral-fusion kernels mix a device-side step counter into their key
    (ops/bifuse.step_counter, incremented inside the graph).
    The returned loss dict holds STATIC tensors that the next call overwrites.
    Signatures beyond `max_graphs` fall back to the eager step."""

    def __init__(self, model, optimizer, warmup_iters=2, max_graphs=4, pad_targets_to=None):
        """pad_targets_to: G - every frame's instance list is padded to G entries (zero masks) before it enters the step and the
        real counts travel in a device tensor (modeling.criterion.SetCriterion.padded_counts): batches whose frames hold different
        numbers of instances (AVSS: 1 .. 4 classes per frame) then share ONE signature, i.e. one captured graph.  A frame with more
        than G instances runs the eager step."""
        self.model, self.opt = model, optimizer
        self.warmup_iters, self.max_graphs = warmup_iters, max_graphs
        self.pad_targets_to = pad_targets_to
        self.graphs = {}
        self._pool = None  # the memory pool all captured signatures share (they replay one after the other)
        # eager_only: the step is NOT captured (every call runs trainer.train_step) because this process replays hipGraph
        # memset nodes wrongly - callers that report a graphed number (bench.py) read this flag
        self.eager_only = False
        self._overflow_noticed = False
        dev = getattr(model, "device", None)
        if dev is not None and torch.device(dev).type == "cuda" and os.environ.get("COMBO_ALLOW_PACKET_CAPTURE") != "1":
            if not graph_memset_selftest(torch.device(dev)):
                msg = ("GraphedTrainStep: memset nodes of a replayed hipGraph are not executed reliably in this process (HIP runtime "
                       "graph packet capture; tools/graph_reduce_repro.py) and library calls inside the step use them.  Export "
                       "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, or import combo_avs_amd before the first HIP call of the process (it sets "
                       "the variable); COMBO_ALLOW_PACKET_CAPTURE=1 skips this check.")
                if os.environ.get("COMBO_GRAPH_STRICT") == "1":
                    raise RuntimeError(msg)
                import logging
                import sys
                logging.getLogger(__name__).warning("%s  Falling back to the eager (un-captured) training step.", msg)
                # an application without logging configured must still see that its step is NOT captured (slower): once, on stderr
                print("[combo_avs_amd] GraphedTrainStep: hipGraph memset self-test failed - running the EAGER (un-captured) training "
                      "step (COMBO_GRAPH_STRICT=1 raises instead); see GraphedTrainStep.eager_only", file=sys.stderr, flush=True)
                self.eager_only = True

    def _num_masks(self, batched_inputs, dev):
        n = 0
        for b in batched_inputs:
            for inst in b["instances"]:
                t = inst["gt_classes"] if isinstance(inst, dict) else inst.gt_classes
                n += int(t.shape[0])
        num = torch.tensor([float(n)], device=dev)
        world = 1
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(num)
            world = dist.get_world_size()
        return torch.clamp(num / world, min=1)

    def _avss_flags(self, batched_inputs):
        """AVSS batches: the VALUES of vid_temporal_mask_flag / gt_temporal_mask_flag select frames (maskformer_model.py:330-331,
        criterion_ss.py:246-257) - read here, on the host, once per step (free when the dataset mapper's CPU tensors arrive as
        they are; ONE device->host copy when the flags already sit on the GPU).  They become part of the graph key and, as
        constant index tensors, of the captured step.  -> ((vid flags), (gt flags)) as python ints, or None."""
        if not self.model.is_avss_data:
            return None
        vid = torch.cat([b["vid_temporal_mask_flag"].reshape(-1) for b in batched_inputs])
        gt = torch.cat([b["gt_temporal_mask_flag"].reshape(-1) for b in batched_inputs])
        both = torch.stack([vid.float(), gt.float()]).cpu()
        return tuple(int(v) for v in both[0].tolist()), tuple(int(v == 1) for v in both[1].tolist())

    def _pad_instances(self, batched_inputs):
        """-> (batch with every instance list padded to pad_targets_to entries, [real count per frame]) or None when a frame
        holds more.  New small tensors per step (two per frame), outside the captured graph."""
        G = self.pad_targets_to
        out, counts = [], []
        for b in batched_inputs:
            nb = dict(b)
            nb["instances"] = []
            for inst in b["instances"]:
                cls = inst["gt_classes"] if isinstance(inst, dict) else inst.gt_classes
                msk = inst["gt_masks"] if isinstance(inst, dict) else inst.gt_masks
                msk = getattr(msk, "tensor", msk)
                n = int(cls.shape[0])
                if n > G:
                    return None
                pc, pm = cls.new_zeros(G), msk.new_zeros((G,) + tuple(msk.shape[1:]))
                if n:
                    pc[:n], pm[:n] = cls, msk
                nb["instances"].append({"gt_classes": pc, "gt_masks": pm})
                counts.append(n)
            out.append(nb)
        return out, counts

    def _fwd_bwd(self, batch):
        from .ops.linear import grouped_presplit
        self.model.record_head_inputs = False
        with grouped_presplit():
            loss_dict = self.model(batch)
            total = getattr(loss_dict, "total", None)
            if total is None:
                total = torch.stack(list(loss_dict.values())).sum()
            self.opt.backward(total)
        return {k: v.detach() for k, v in loss_dict.items()}

    def _cut_backward(self):
        """Data parallel: the step is captured as TWO graphs - forward + loss + the head's backward, then the backbones'
        backward - so that the all-reduce of the head's gradients (the front region of the flat buffer) runs on a side stream
        while the second graph executes."""
        o = self.opt
        return o._dist_on() and 0 < o.n_early < len(o.params) and os.environ.get("COMBO_DP_OVERLAP", "1") == "1"

    def _fwd_early(self, batch):
        self.model.record_head_inputs = True
        loss_dict = self.model(batch)
        total = getattr(loss_dict, "total", None)
        if total is None:
            total = torch.stack(list(loss_dict.values())).sum()
        cut, self.model._head_inputs = self.model._head_inputs, None
        cut_grads = self.opt.backward_early(total, cut)
        return {k: v.detach() for k, v in loss_dict.items()}, cut_grads

    def _capture(self, batched_inputs, num_masks, flags=None, counts=None):
        from .ops import bifuse
        dev = num_masks.device
        static_batch = _clone_batch(batched_inputs)
        static_num = num_masks.clone()
        crit = self.model.criterion
        crit.num_masks_override = static_num
        static_counts = None
        if counts is not None:  # padded targets: the real counts per frame, refreshed before every replay
            static_counts = torch.tensor(counts, dtype=torch.int32, device=dev)
            crit.padded_counts = static_counts
        keep = None
        if flags is not None:  # AVSS: the selections the flag values stand for, as constants of this graph
            keep = (torch.tensor([i for i, v in enumerate(flags[0]) if v], dtype=torch.long, device=dev),
                    torch.tensor([i for i, v in enumerate(flags[1]) if v], dtype=torch.long, device=dev))
            self.model.avss_static_index = keep  # (allocated OUTSIDE the capture: the graph's kernels read them on every replay -
            # they must live as long as the graph does, hence `keep` in the returned record)
        counter = bifuse.step_counter(dev)
        from .ops.linear import grouped_presplit
        cut = self._cut_backward()
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            # (every signature gets its eager iterations: lazily built index tensors - host -> device copies - must exist before
            #  the capture.  At 512 x 512 a second signature does not fit next to the first graph's 200 GB pool: the caller falls back
            #  to the eager step; pad_targets_to makes the instance counts a non-issue)
            n_warm = self.warmup_iters
            with torch.cuda.stream(side):
                for _ in range(n_warm):  # autotuning / lazy initialisation outside the capture
                    if cut:
                        with grouped_presplit():
                            _, cg = self._fwd_early(static_batch)
                            self.opt.backward_late(cg)
                    else:
                        self._fwd_bwd(static_batch)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            if torch.cuda.memory_reserved(dev) > 0.3 * torch.cuda.get_device_properties(dev).total_memory:
                torch.cuda.empty_cache()  # the eager iterations' cached blocks (two streams' worth) back to the runtime: the pool needs them
            graph = torch.cuda.CUDAGraph()
            # graphs of different input signatures (AVSS: instance counts per frame vary from batch to batch) replay one after
            # the other, never concurrently: they share ONE memory pool - the second capture reuses the activations' blocks of
            # the first instead of reserving its own 100+ GB
            pool_kw = {} if self._pool is None else {"pool": self._pool}
            # thread_local: RCCL's watchdog thread may poll events while this thread captures; that must not
            # invalidate the capture (the default "global" mode would)
            if cut:
                graph_b = torch.cuda.CUDAGraph()
                with grouped_presplit():  # spans both captures: the second graph's input-gradient GEMMs use images split in the first
                    with torch.cuda.graph(graph, capture_error_mode="thread_local", **pool_kw):
                        counter.add_(1)
                        static_losses, cut_grads = self._fwd_early(static_batch)
                    with torch.cuda.graph(graph_b, pool=graph.pool(), capture_error_mode="thread_local"):
                        self.opt.backward_late(cut_grads)
                    del cut_grads
                graph = (graph, graph_b)
            else:
                with torch.cuda.graph(graph, capture_error_mode="thread_local", **pool_kw):
                    counter.add_(1)
                    static_losses = self._fwd_bwd(static_batch)
            if self._pool is None:
                self._pool = (graph[0] if isinstance(graph, tuple) else graph).pool()
        finally:
            crit.num_masks_override = None
            crit.padded_counts = None
            self.model.avss_static_index = None
        return graph, static_batch, static_num, static_losses, static_counts, keep

    def __call__(self, batched_inputs):
        tensors, rest = _input_leaves(batched_inputs)
        dev = self.model.device
        num_masks = self._num_masks(batched_inputs, dev)
        # (AVSS flag tensors may stay on the CPU, as the dataset mapper hands them over: their values travel in the key)
        flag_keys = ("vid_temporal_mask_flag", "gt_temporal_mask_flag")
        flag_ids = {id(b[k]) for b in batched_inputs for k in flag_keys if k in b} if self.model.is_avss_data else set()
        if self.eager_only or num_masks is None or not all(t.is_cuda for t in tensors if id(t) not in flag_ids):
            return train_step(self.model, self.opt, batched_inputs)
        flags = self._avss_flags(batched_inputs)
        counts = None
        original_batch = batched_inputs  # (the eager fallback below takes the caller's batch, not the padded one)
        if self.pad_targets_to:
            padded = self._pad_instances(batched_inputs)
            if padded is None:  # a frame with more instances than the padded length
                return train_step(self.model, self.opt, batched_inputs)
            batched_inputs, counts = padded
            tensors, rest = _input_leaves(batched_inputs)
        key = (tuple((tuple(t.shape), t.dtype) for t in tensors), rest, self.model.training, flags)
        if key not in self.graphs:
            if len(self.graphs) >= self.max_graphs:
                # AVSS: the flag VALUES are part of the key (three patterns per clip in the real data: v1s / v1m / v2,
                # register_avss_sem.py:36-43) - more signatures than graphs is the normal case there, not an error: such a
                # batch runs the eager step on the caller's own (unpadded) batch
                if not self._overflow_noticed:
                    self._overflow_noticed = True
                    import sys
                    print(f"[combo_avs_amd] GraphedTrainStep: more than max_graphs = {self.max_graphs} input signatures - further new "
                          "signatures run the eager (un-captured) step", file=sys.stderr, flush=True)
                return train_step(self.model, self.opt, original_batch)
            self.graphs[key] = self._capture(batched_inputs, num_masks, flags, counts)
        graph, static_batch, static_num, static_losses, static_counts, _keep = self.graphs[key]
        if static_counts is not None:
            static_counts.copy_(torch.tensor(counts, dtype=torch.int32), non_blocking=True)
        static_tensors, _ = _input_leaves(static_batch)
        src = [t for t, s in zip(tensors, static_tensors) if t.data_ptr() != s.data_ptr()]
        dst = [s for t, s in zip(tensors, static_tensors) if t.data_ptr() != s.data_ptr()]
        if src:
            torch._foreach_copy_(dst, src)
        static_num.copy_(num_masks)
        if isinstance(graph, tuple):  # cut backward: the head's gradients travel while the backbones' backward replays
            graph[0].replay()
            self.opt.all_reduce_grads("early", overlap=True)
            graph[1].replay()
            self.opt.all_reduce_grads("late")
            self.opt.wait_comm()
        else:
            graph.replay()
            self.opt.all_reduce_grads()
        self.opt.step()
        return static_losses
