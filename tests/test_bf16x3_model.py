"""CPU: numpy model of the arithmetic of the head's GRADIENT GEMM kernels (csrc/gemm_nt3.hip, gemm_tn.hip) - every fp32
operand is split into hi = round_to_nearest_bf16(x) and lo = round_to_nearest_bf16(x - hi), and a product x*w is the sum of
the three bf16 products hi*hi + hi*lo + lo*hi accumulated in fp32 (the MFMA accumulator).  Pins the error model DESIGN.md
quotes (operands exact to 2^-17, a single product to 2^-15.5 worst case, ~4e-6 on a K = 256 dot product: ~3x better than
round 1's truncated `hi`, still ~15x fp32 - which is why the FORWARD GEMMs do not use this split: exact fp32 MFMA until round 5, three
products on fp16 pieces since round 6, tests/test_f16x3_model.py) and the
pre-split image layout of combo_presplit_bf16x2_f32 (per 8 k a 16-byte hi group followed by a 16-byte lo group)."""
import numpy as np


def trunc_bf16(x):
    return (x.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def rne_bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split(x, rounded=True):
    hi = rne_bf16(x) if rounded else trunc_bf16(x)
    lo = rne_bf16((x - hi).astype(np.float32))
    return hi, lo


def gemm_bf16x3(a, w, rounded=True):
    ah, al = split(a, rounded)
    wh, wl = split(w, rounded)
    f = np.float64  # the fp32 accumulator's own rounding is not what is modelled here
    return (al.astype(f) @ wh.astype(f).T) + (ah.astype(f) @ wl.astype(f).T) + (ah.astype(f) @ wh.astype(f).T)


def test_split_is_exact_to_17_bits_and_products_to_2e_minus_15():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(1 << 16).astype(np.float32) * np.float32(3.7)
    hi, lo = split(x)
    rel = np.abs((x.astype(np.float64) - hi - lo.astype(np.float64)) / x)
    assert rel.max() <= 2.0 ** -17  # |x - hi| <= 2^-9 |x| after rounding, lo carries the next 8 bits of it
    w = rng.standard_normal(1 << 16).astype(np.float32)
    wh, wl = split(w)
    p3 = hi.astype(np.float64) * wh + hi.astype(np.float64) * wl + lo.astype(np.float64) * wh
    relp = np.abs((p3 - x.astype(np.float64) * w) / (x.astype(np.float64) * w))
    # the dropped lo*lo term: |lo| <= 2^-8 |x| after rounding, so a single product is off by at most ~2^-16 + 2 * 2^-17
    assert relp.max() <= 2.0 ** -15 and np.median(relp) < 2.0 ** -18


def test_rounded_hi_beats_truncated_hi():
    """round 1 truncated `hi` (lo up to 2^-7 |x| and of x's sign: a biased lo*lo term); the rounded split is ~3x closer"""
    rng = np.random.default_rng(3)
    a = rng.standard_normal((64, 256)).astype(np.float32)
    w = (rng.standard_normal((96, 256)) / 16).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    e_r = np.linalg.norm(gemm_bf16x3(a, w, True) - ref) / np.linalg.norm(ref)
    e_t = np.linalg.norm(gemm_bf16x3(a, w, False) - ref) / np.linalg.norm(ref)
    assert e_r < 6e-6 and e_r < e_t / 2


def test_dot_products_land_between_fp32_and_plain_bf16():
    rng = np.random.default_rng(1)
    a = rng.standard_normal((64, 256)).astype(np.float32)
    w = (rng.standard_normal((96, 256)) / 16).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    e3 = np.linalg.norm(gemm_bf16x3(a, w) - ref) / np.linalg.norm(ref)
    e32 = np.linalg.norm((a @ w.T).astype(np.float64) - ref) / np.linalg.norm(ref)
    e16 = np.linalg.norm(rne_bf16(a).astype(np.float64) @ rne_bf16(w).astype(np.float64).T - ref) / np.linalg.norm(ref)
    assert e3 < 1e-5      # what tests/test_gemm_gpu.py demands of the kernels
    assert e3 < e16 / 100  # two orders of magnitude better than one bf16 product
    assert e32 < e3 / 5    # and not fp32: the reason no forward GEMM runs on bf16 pieces


def test_presplit_image_layout():
    """image row n: for every group of 8 k: 8 bf16 `hi` values (16 bytes) then 8 bf16 `lo` values (16 bytes)"""
    rng = np.random.default_rng(2)
    w = rng.standard_normal((4, 32)).astype(np.float32)
    hi, lo = split(w)
    hi16 = (hi.view(np.uint32) >> 16).astype(np.uint16)
    lo16 = (lo.view(np.uint32) >> 16).astype(np.uint16)
    img = np.empty((4, 32 // 8, 2, 8), dtype=np.uint16)
    img[:, :, 0, :] = hi16.reshape(4, 4, 8)
    img[:, :, 1, :] = lo16.reshape(4, 4, 8)
    as_f32 = img.reshape(4, -1).view(np.float32)
    assert as_f32.shape == w.shape  # the same 4 bytes per element, row pitch K floats
    # a stage of 16 k = 64 bytes per row = chunks [hi(k0..7), lo(k0..7), hi(k8..15), lo(k8..15)]: lane half g reads chunks 2g, 2g+1
    row = img[1].reshape(-1)
    np.testing.assert_array_equal(row[0:8], hi16[1, 0:8])
    np.testing.assert_array_equal(row[8:16], lo16[1, 0:8])
    np.testing.assert_array_equal(row[16:24], hi16[1, 8:16])
    np.testing.assert_array_equal(row[24:32], lo16[1, 8:16])
