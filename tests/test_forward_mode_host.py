"""CPU: host-side logic of the head's forward-arithmetic modes (ops/linear.py, round 6) - no kernel is launched: the mode table and its
environment default, the nesting rule of forward_precision_scope, range_safe, which tensors may enter the grouped forward pre-split plan
(parameters and views of parameters only: a tensor written during the step could be read before it is written), and the `products` codes
the C ABI documents (include/combo_avs.h combo_gemm_nt2_products / combo_presplit_pieces)."""
import os
import re

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mode_table_default_and_products_codes():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import linear as L
    assert L.FORWARD_MODES == ("fp32", "f16x3", "x3", "bf16")
    assert L.DEFAULT_FORWARD_PRECISION == os.environ.get("COMBO_HEAD_FORWARD", "f16x3") and L.FORWARD_PRECISION in L.FORWARD_MODES
    prev = L.FORWARD_PRECISION
    try:
        for mode, products, f16 in (("fp32", 3, False), ("f16x3", 19, True), ("x3", 3, False), ("bf16", 1, False)):
            L.set_forward_precision(mode)
            assert (L.forward_products(), L.forward_f16()) == (products, f16)
        try:
            L.set_forward_precision("fp16")
            raise AssertionError("an unknown mode was accepted")
        except ValueError:
            pass
    finally:
        L.set_forward_precision(prev)
    hdr = open(os.path.join(ROOT, "include", "combo_avs.h")).read()
    assert "int combo_presplit_pieces(int f16);" in hdr and re.search(r"19 = the 3-product\s+\*?\s*split on fp16 hi / lo pieces", hdr)
    nt3 = open(os.path.join(ROOT, "combo-avs_amd", "csrc", "gemm_nt3.h")).read()
    assert "COMBO_PRODUCTS_F16X3 = 19" in nt3 and "COMBO_PRODUCTS_F16X3_UNSCALED = 35" in nt3 and "#define COMBO_F16_BSCALE_LOG2 8" in nt3


def test_scopes_nest_against_the_outer_mode_and_range_safe_only_touches_fp16_pieces():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import linear as L
    prev = L.FORWARD_PRECISION
    try:
        L.set_forward_precision("fp32")
        with L.forward_precision_scope("x3"):
            assert L.FORWARD_PRECISION == "x3"
            with L.forward_precision_scope("f16x3"):  # judged against the OUTER mode (fp32), not against the enclosing scope
                assert L.FORWARD_PRECISION == "f16x3"
            assert L.FORWARD_PRECISION == "x3"
        assert L.FORWARD_PRECISION == "fp32" and L.forward_precision_scope._base is None
        for mode, inside in (("f16x3", "fp32"), ("x3", "x3"), ("bf16", "bf16"), ("fp32", "fp32")):
            L.set_forward_precision(mode)
            with L.range_safe():
                assert L.FORWARD_PRECISION == inside
            assert L.FORWARD_PRECISION == mode
    finally:
        L.set_forward_precision(prev)


def test_only_parameters_and_their_views_enter_the_forward_plan():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import linear as L
    lin = torch.nn.Linear(64, 48)
    attn = torch.nn.Parameter(torch.randn(192, 64))
    assert L._plannable(lin.weight) and L._plannable(attn[:64])
    assert not L._plannable(attn[64:128].detach())  # (a detached view no longer names its base: conservatively left out)
    with torch.no_grad():
        assert L._plannable(attn[128:])
    assert not L._plannable(torch.randn(48, 64))                      # a plain tensor
    assert not L._plannable(torch.cat([attn[:64], attn[128:]], 0))    # computed during the step (ops.linear.memory_kv's Wcat)
    assert not L._plannable(lin.weight.t().contiguous())              # a re-laid-out copy (ops.conv3x3's [cout, 9 cin] matrix)
    assert not L._plannable(lin.bias) and not L._plannable(torch.nn.Parameter(torch.randn(8, 12)))  # not 2-D / K % 8 != 0
    # the plan is adopted by the OUTERMOST completed grouped_presplit context only
    saved = dict(L._fwd_plan)
    try:
        L.reset_forward_plan()
        with L.grouped_presplit():
            L._fwd_plan_next[("k", False)] = lin.weight
            with L.grouped_presplit():
                L._fwd_plan_next[("inner", False)] = lin.weight
            assert ("k", False) in L._fwd_plan_next and ("inner", False) not in L._fwd_plan_next and not L._fwd_plan
        assert list(L._fwd_plan) == [("k", False)]
        try:
            with L.grouped_presplit():
                L._fwd_plan_next[("broken", False)] = lin.weight
                raise RuntimeError("step failed")
        except RuntimeError:
            pass
        assert list(L._fwd_plan) == [("k", False)]  # a failed step leaves the plan alone
    finally:
        L.reset_forward_plan()
        L._fwd_plan.update(saved)
