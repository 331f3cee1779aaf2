"""Channel sums without ATen's multi-workgroup reduction (csrc/colsum.hip): bias gradients of the host-PyTorch backbones and
the level-embedding gradients.  ATen splits long reductions over workgroups behind a hipMemsetAsync of its semaphores; those
memset nodes do not replay reliably from a hipGraph on this stack (tools/graph_reduce_repro.py), so the training step uses
these ops wherever a reduction has >= ~2 000 inputs per output (tools/graph_reductions.py lists what is left)."""
import torch
from torch.autograd import Function

from .. import _lib

import ctypes

PENDING_CAP = 1024 << 20  # bytes of queued inputs (they stay alive until the flush) beyond which a queue flushes by itself
DEFER = True              # False: every bias gradient by its own pair of launches (A/B)


class _ColsumProblem(ctypes.Structure):  # combo_colsum_problem (include/combo_avs.h)
    _fields_ = [("x", ctypes.c_void_p), ("out", ctypes.c_void_p), ("partial", ctypes.c_void_p), ("rows", ctypes.c_longlong),
                ("C", ctypes.c_int), ("in_bf16", ctypes.c_int), ("out_bf16", ctypes.c_int)]


class DeferredColumnSums:
    """The bias gradients ONE backbone application queues during its backward pass (linear_bias(..., queue=q)) and the consumer
    of those gradients (backbone_pvt._CastAll.backward of the SAME application) computes with one grouped launch (+ one finish
    launch) per 40 problems.  One queue per application, not a module global: the two PVT backbones run their backward passes
    on different HIP streams, and a queue they shared would be flushed by whichever finishes first - on its own stream, with
    no event between it and the other backbone's producers.  Every entry is produced and flushed inside one autograd-stream
    context; `flush` checks that.  A queue that dies unflushed (an exception inside the backward pass) frees its inputs with
    the autograd graph that holds it - nothing is kept alive or summed into a later step."""

    def __init__(self):
        self.items, self.bytes, self.stream = [], 0, None

    def __len__(self):
        return len(self.items)

    def add(self, x2, out_dtype):
        """x2 [rows, C] contiguous -> a [C] tensor that is filled at the next flush()"""
        st = torch.cuda.current_stream(x2.device)
        if self.items and st != self.stream:
            raise RuntimeError("colsum: one deferred queue fed from two HIP streams (a queue belongs to ONE backbone application)")
        self.stream = st
        out = torch.empty(x2.shape[1], dtype=out_dtype, device=x2.device)
        self.items.append((x2, out))
        self.bytes += x2.numel() * x2.element_size()
        if self.bytes > PENDING_CAP:
            self.flush()
        return out

    def flush(self):
        if not self.items:
            return
        if torch.cuda.current_stream(self.items[0][0].device) != self.stream:
            raise RuntimeError("colsum: deferred bias gradients flushed on another HIP stream than the one that queued them")
        q, self.items, self.bytes = self.items, [], 0
        lib = _lib.lib()
        arr = (_ColsumProblem * len(q))()
        sl = [lib.combo_colsum_grouped_slices(x.shape[0], x.shape[1]) for x, _ in q]
        need = sum(s * x.shape[1] for s, (x, _) in zip(sl, q) if s > 1)
        scratch = torch.empty(max(need, 1), dtype=torch.float32, device=q[0][0].device)
        off = 0
        for i, (s_i, (x, out)) in enumerate(zip(sl, q)):
            part = 0
            if s_i > 1:
                part = scratch.data_ptr() + 4 * off
                off += s_i * x.shape[1]
            arr[i] = _ColsumProblem(x.data_ptr(), out.data_ptr(), part, x.shape[0], x.shape[1], _code(x.dtype), _code(out.dtype))
        _lib.check(lib.combo_colsum_grouped(ctypes.cast(arr, ctypes.c_void_p), len(q), _lib.current_stream()), "combo_colsum_grouped")


def _code(dt):
    if dt == torch.bfloat16:
        return 1
    if dt == torch.float32:
        return 0
    raise RuntimeError("colsum: float32 / bfloat16 tensors only")


def channel_sum(x, A, C, L, out_dtype=None):
    """out[c] = sum over a, l of x viewed as [A, C, L] (x contiguous in that view)."""
    if not (x.is_cuda and x.is_contiguous() and x.numel() == A * C * L):
        raise RuntimeError("colsum: a contiguous CUDA tensor of A*C*L elements expected")
    out = torch.empty(C, dtype=out_dtype or x.dtype, device=x.device)
    lib = _lib.lib()
    slices = lib.combo_colsum_slices(A, C, L)
    if slices < 1:
        raise RuntimeError(f"colsum: unsupported shape A={A} C={C} L={L} (L == 1 needs C % 4 == 0)")
    partial = torch.empty(slices, C, dtype=torch.float32, device=x.device) if slices > 1 else None
    _lib.check(lib.combo_colsum(x.data_ptr(), A, C, L, _code(x.dtype), out.data_ptr(), _code(out.dtype),
                                0 if partial is None else partial.data_ptr(), _lib.current_stream()), "combo_colsum")
    return out


def supported(x, channel_dim):
    """the kernel's envelope: CUDA fp32 / bf16; channels last needs C % 4 == 0"""
    if not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16) or x.numel() == 0:
        return False
    cd = channel_dim % x.dim()
    return cd != x.dim() - 1 or x.shape[-1] % 4 == 0


def sum_to_channels(x, channel_dim=-1, out_dtype=None):
    """sum of x over every dimension but `channel_dim` (last: x[..., C]; 1: x[B, C, ...])"""
    cd = channel_dim % x.dim()
    if cd == 1 and x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last) \
            and x.shape[1] % 4 == 0:
        x, cd = x.permute(0, 2, 3, 1), 3  # NHWC in memory: rows of C
    x = x.contiguous()
    if cd == x.dim() - 1:
        return channel_sum(x, x.numel() // x.shape[-1], x.shape[-1], 1, out_dtype)
    if cd == 1:
        return channel_sum(x, x.shape[0], x.shape[1], x.numel() // (x.shape[0] * x.shape[1]), out_dtype)
    raise RuntimeError("colsum: channel dimension must be 1 or the last one")


def sum_leading(x, dims):
    """x.sum(dims) for the two layouts above (used by tools/graph_reduce_repro.py)"""
    dims = tuple(sorted(d % x.dim() for d in dims))
    keep = [d for d in range(x.dim()) if d not in dims]
    if len(keep) != 1:
        raise RuntimeError("colsum: exactly one kept dimension")
    return sum_to_channels(x, keep[0])


class _AddChannelVector(Function):
    """y = x + v broadcast along `channel_dim`; dv by channel_sum (autograd would call ATen's sum)."""

    @staticmethod
    def forward(ctx, x, v, channel_dim):
        ctx.channel_dim = channel_dim % x.dim()
        ctx.v_dtype = v.dtype
        shape = [1] * x.dim()
        shape[ctx.channel_dim] = -1
        return x + v.to(x.dtype).view(shape)

    @staticmethod
    def backward(ctx, dy):
        dv = sum_to_channels(dy, ctx.channel_dim, out_dtype=ctx.v_dtype) if ctx.needs_input_grad[1] else None
        return dy, dv, None


def add_channel_vector(x, v, channel_dim=1):
    if supported(x, channel_dim) and v.dtype in (torch.float32, torch.bfloat16):
        return _AddChannelVector.apply(x, v, channel_dim)
    shape = [1] * x.dim()
    shape[channel_dim % x.dim()] = -1
    return x + v.view(shape)  # CPU / other dtypes (unit tests of the host logic)


class _LinearBias(Function):
    """F.linear with a bias whose gradient is channel_sum(dy) (backbone_pvt._linear).  Under autocast the forward computes in
    the autocast dtype while x / w / b may be fp32: the backward GEMMs then run in dy's dtype and the gradients are cast to the
    inputs' dtypes, as autocast's own cast nodes would do.  queue (a DeferredColumnSums): the bias gradient is queued there and
    computed by the queue's next flush() (the caller guarantees one before anything reads it)."""

    @staticmethod
    def forward(ctx, x, w, b, queue):
        ctx.save_for_backward(x, w)
        ctx.b_dtype, ctx.queue = b.dtype, queue
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = (dy2 @ w.to(dy2.dtype)).view_as(x).to(x.dtype)
        if ctx.needs_input_grad[1]:
            dw = (dy2.t() @ x.reshape(-1, x.shape[-1]).to(dy2.dtype)).to(w.dtype)
        if ctx.needs_input_grad[2]:
            if ctx.queue is not None and DEFER and dy2.is_contiguous():
                db = ctx.queue.add(dy2, ctx.b_dtype)
            else:
                db = sum_to_channels(dy2, -1, out_dtype=ctx.b_dtype)
        return dx, dw, db, None


def linear_bias(x, w, b, queue=None):
    ok = (torch.float32, torch.bfloat16)
    if x.is_cuda and x.numel() > 0 and w.shape[0] % 4 == 0 and x.dtype in ok and w.dtype in ok and b.dtype in ok:
        return _LinearBias.apply(x, w, b, queue)
    return torch.nn.functional.linear(x, w, b)


class _RowSum(Function):
    """x[R, n].sum(1) (the weighted class-loss sums of the criterion: n = frames x queries per output)"""

    @staticmethod
    def forward(ctx, x):
        ctx.n = x.shape[1]
        return channel_sum(x.contiguous(), 1, x.shape[0], x.shape[1])

    @staticmethod
    def backward(ctx, dy):
        return dy[:, None].expand(-1, ctx.n)


def row_sum(x):
    if x.is_cuda and x.dim() == 2 and x.dtype in (torch.float32, torch.bfloat16) and x.numel() > 0:
        return _RowSum.apply(x)
    return x.sum(1)
