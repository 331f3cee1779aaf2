"""Token stage of the bilateral audio-visual fusion (LayerNorm -> scores -> softmax over HW -> rank-8 update
+ attention pooling), see modeling/fusion.py for the algebra.  HIP kernels: csrc/bifuse.hip."""
import torch
import torch.nn.functional as F

_IMPL = "torch"  # switched to "hip" once the kernels are built (set_impl)


def set_impl(name):
    global _IMPL
    assert name in ("torch", "hip")
    _IMPL = name


def _token_op_torch(x, ln_w, ln_b, eps, pos, u, c, z, b_ov, gamma_v, p_drop):
    xn = F.layer_norm(x, (x.shape[-1],), ln_w, ln_b, eps)
    s = torch.einsum("bic,bhc->bhi", xn + pos, u) + c[..., None]
    p = torch.softmax(s.clamp(min=-50000, max=50000), dim=-1)
    pv = pa = p
    if p_drop > 0.0:
        pv = p * ((torch.rand_like(p) >= p_drop).to(p.dtype) / (1.0 - p_drop))
        pa = p * ((torch.rand_like(p) >= p_drop).to(p.dtype) / (1.0 - p_drop))
    y = xn + gamma_v * (torch.einsum("bhi,bhc->bic", pv, z) + b_ov)
    pooled = torch.einsum("bhi,bic->bhc", pa, xn)
    return y, pooled, pa.sum(-1)


def token_op(x, ln_w, ln_b, eps, pos, u, c, z, b_ov, gamma_v, p_drop):
    return _token_op_torch(x, ln_w, ln_b, eps, pos, u, c, z, b_ov, gamma_v, p_drop)
