"""CPU: host-side logic added with the round-1 GEMM work - routing predicate of the forward / dX GEMMs, the applicability
test of the implicit-GEMM 3x3 convolution, and the library fall-backs of `ffn` / `Conv2d` on tensors the HIP path does
not take (CPU tensors here: same code path as small or unaligned GPU tensors)."""
import torch
import torch.nn.functional as F


class _FakeCuda:
    """shape/stride/pointer view of a would-be CUDA tensor for the kernel-contract predicates (no GPU in this container)"""
    is_cuda = True
    dtype = torch.float32

    def __init__(self, rows, cols, stride0=None, ptr=4096):
        self.shape = (rows, cols)
        self._s = (stride0 or cols, 1)
        self._p = ptr

    def dim(self):
        return 2

    def stride(self, i):
        return self._s[i]

    def data_ptr(self):
        return self._p


def test_kernel_contract_predicates():
    """ops.linear routes every fp32 CUDA layer to the head's own kernels; only operands outside the kernels' contracts
    (K % 16, 16-byte alignment, row pitch % 4, 32-bit addressable output) go through torch."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import linear as L
    ok = lambda M, K, N, **kw: L.f32_ok(_FakeCuda(M, K, **kw), _FakeCuda(N, K))  # noqa: E731
    for M, K, N in ((41160, 256, 1024), (4000, 256, 256), (1960, 256, 512), (40, 128, 4096), (4000, 256, 3), (1, 16, 1)):
        assert ok(M, K, N)
    assert not ok(41160, 250, 1024)               # K % 16
    assert not ok(41160, 256, 1024, ptr=4100)     # 16-byte alignment
    assert not ok(41160, 256, 1024, stride0=258)  # row pitch % 4
    assert not ok(1 << 20, 256, 1024)             # output beyond 32-bit byte offsets
    assert L.x3_ok(_FakeCuda(4000, 256), 256) and not L.x3_ok(_FakeCuda(4000, 3), 256)
    # deferred weight gradients: destination / operand contracts of the grouped kernel
    assert L._dest_ok(torch.empty(256, 256)) and not L._dest_ok(torch.empty(3, 256)) and not L._dest_ok(torch.empty(256, 40))
    assert L._use_ok(_FakeCuda(4000, 256), _FakeCuda(4000, 256)) and not L._use_ok(_FakeCuda(40, 256), _FakeCuda(40, 256))


def test_conv3x3_applicability():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import conv3x3 as C
    x = torch.randn(2, 256, 8, 8).contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False)
    assert not C.usable(conv, x)  # CPU tensor: never (the op has no CPU path)
    for bad in (torch.nn.Conv2d(256, 256, 3, padding=1, stride=2), torch.nn.Conv2d(256, 256, 3, padding=0),
                torch.nn.Conv2d(256, 256, 3, padding=1, groups=2), torch.nn.Conv2d(256, 256, 3, padding=1, dilation=2),
                torch.nn.Conv2d(256, 256, 1)):
        assert not C.usable(bad, x)


def test_conv2d_wrapper_and_ffn_fall_back_to_the_library_off_gpu():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.modeling.layers import Conv2d
    from combo_avs_amd.ops import linear as L
    torch.manual_seed(0)
    conv = Conv2d(16, 16, kernel_size=3, stride=1, padding=1, bias=False)
    x = torch.randn(2, 16, 6, 6)
    torch.testing.assert_close(conv(x), F.conv2d(x, conv.weight, None, 1, 1))
    xf = torch.randn(40, 32, requires_grad=True)
    w1, b1, w2, b2 = (torch.randn(64, 32, requires_grad=True), torch.randn(64, requires_grad=True),
                      torch.randn(32, 64, requires_grad=True), torch.randn(32, requires_grad=True))
    y = L.ffn(xf, w1, b1, w2, b2)
    ref = F.linear(torch.relu(F.linear(xf, w1, b1)), w2, b2)
    torch.testing.assert_close(y, ref)
    g = torch.randn_like(y)
    for a, b in zip(torch.autograd.grad(y, (xf, w1, b1, w2, b2), g), torch.autograd.grad(ref, (xf, w1, b1, w2, b2), g)):
        torch.testing.assert_close(a, b)
