"""CPU: numpy model of the arithmetic of the head's FORWARD GEMMs since round 6 ("f16x3", csrc/gemm_nt3.hip template parameter F16,
csrc/gemm_x3.hip split_pair): every fp32 operand is split into hi = rne_f16(x) and lo = rne_f16(x - hi) (numpy's float16 rounds to
nearest even and keeps subnormals, like v_cvt_pk_f16_f32 on gfx950: tools/ubench/mfma_f16_denorm.hip), a product x * w is hi*hi + lo*hi
+ hi*lo accumulated in fp32, the weight image is split from 2^8 * w and the accumulator multiplied by 2^-8.  Pins the error model
DESIGN.md section 2 quotes: operands exact to 2^-22, dot products as close to float64 as fp32's own (why the mode may replace the exact
fp32 instruction: tests/test_bf16x3_model.py shows the bf16 split may not), what the 2^8 buys for typical weights, the range (65 504 /
255) and the stated weakness (activations of magnitude << 0.1)."""
import numpy as np

from test_bf16x3_model import gemm_bf16x3

WSCALE = np.float32(256.0)


def split16(x):
    x = x.astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def gemm_f16x3(a, w, wscale=WSCALE):
    ah, al = split16(a)
    wh, wl = split16(w * wscale)
    f = np.float64  # (the products of two 11-bit pieces are exact in fp32; the accumulator's own rounding is not modelled)
    acc = (al.astype(f) @ wh.astype(f).T) + (ah.astype(f) @ wl.astype(f).T) + (ah.astype(f) @ wh.astype(f).T)
    return acc / float(wscale)


def test_split_is_exact_to_22_bits_in_the_normal_range():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(1 << 16) * 3.7).astype(np.float32)
    x = x[np.abs(x) > 0.25]  # |lo| <= 2^-11 |x| stays an fp16 NORMAL (>= 2^-14) for |x| >= 2^-3
    hi, lo = split16(x)
    rel = np.abs((x.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64)) / x)
    assert rel.max() <= 2.0 ** -22
    # ... and small operands hit the absolute floor of fp16 subnormals, 2^-25 (half a spacing of 2^-24)
    y = (rng.standard_normal(1 << 12) * 1e-3).astype(np.float32)
    hi, lo = split16(y)
    assert np.abs(y.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64)).max() <= 2.0 ** -25


def test_dot_products_are_as_close_to_float64_as_fp32s_own():
    rng = np.random.default_rng(1)
    a = rng.standard_normal((64, 256)).astype(np.float32)            # LayerNorm-like activations
    w = (rng.standard_normal((96, 256)) * 0.03).astype(np.float32)   # |w| ~ 0.03
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    e16 = np.linalg.norm(gemm_f16x3(a, w) - ref) / np.linalg.norm(ref)
    e32 = np.linalg.norm((a @ w.T).astype(np.float64) - ref) / np.linalg.norm(ref)
    eb3 = np.linalg.norm(gemm_bf16x3(a, w) - ref) / np.linalg.norm(ref)
    assert e16 < 2e-7 and e16 < 2 * e32  # (measured on MI355X: 2.0e-7 against the exact kernel's 2.9e-7)
    assert e16 < eb3 / 20                # the bf16 split: 4.4e-6


def test_the_weight_scale_keeps_the_lo_pieces_of_typical_weights_normal():
    rng = np.random.default_rng(2)
    a = (rng.standard_normal((64, 512)) * 30).astype(np.float32)
    w = (rng.standard_normal((96, 512)) * 0.02).astype(np.float32)  # lo pieces ~ 1e-5: fp16 subnormals when split unscaled
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    e_scaled = np.linalg.norm(gemm_f16x3(a, w) - ref) / np.linalg.norm(ref)
    e_plain = np.linalg.norm(gemm_f16x3(a, w, np.float32(1.0)) - ref) / np.linalg.norm(ref)
    assert e_scaled < 2e-7 and e_plain > 3 * e_scaled  # (measured: 8.7e-7 -> 1.2e-7)


def test_range_and_the_stated_weakness():
    hi, lo = split16(np.array([7.0e4, 6.0e4, 250.0 * 256, 260.0 * 256], dtype=np.float32))
    assert np.isinf(hi[0]) and np.isfinite(hi[1]) and np.isfinite(hi[2]) and np.isinf(hi[3])  # activations < 65 504, weights < 255
    rng = np.random.default_rng(3)
    a = (rng.standard_normal((64, 256)) * 1e-3).astype(np.float32)  # an activation tensor of magnitude 1e-3: NOT what the head holds
    w = (rng.standard_normal((96, 256)) * 0.03).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    e = np.linalg.norm(gemm_f16x3(a, w) - ref) / np.linalg.norm(ref)
    assert 3e-6 < e < 4e-5  # the absolute floor 2^-25 over |a| ~ 1e-3 (tests/test_f16x3_gpu.py holds the kernel to the same)
