"""Host-side planning logic of the 3-product kernels (no GPU: the plan functions fall back to 256 CUs without a device) and the
layer classification of ops/convwrw.py for the ResNet-50 backbones."""
import torch


def test_conv_tap_split_plans_for_the_resnet_maps():
    """combo_conv3x3_x3_splitk_plan: many-tile maps are not split; res4 (7 840 tokens) / res5 (1 960) split over kernel rows"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import _lib
    lib = _lib.lib()
    assert lib.combo_conv3x3_x3_splitk_plan(125440, 64, 64) == 1      # res2 at 40 frames: 980 tiles
    assert lib.combo_conv3x3_x3_splitk_plan(31360, 128, 128) == 1      # res3: 245 tiles
    assert lib.combo_conv3x3_x3_splitk_plan(7840, 256, 256) in (3, 9)  # res4: 124 tiles on 256 CUs
    assert lib.combo_conv3x3_x3_splitk_plan(1960, 512, 512) in (3, 9)  # res5: 64 tiles
    assert lib.combo_conv3x3_x3_splitk_plan(1960, 512, 48) == 1        # channel count the split kernel does not take
    for m, n, k in ((4000, 256, 2048), (7840, 256, 1024), (1960, 512, 2048)):
        s = lib.combo_gemm_nt_x3_splitk_plan(m, n, k)
        assert s >= 2 and k % (s * 32) == 0 and k // s >= 128, (m, n, k, s)
    assert lib.combo_gemm_nt_x3_splitk_plan(41160, 256, 1024) == 1     # enough tiles already
    assert lib.combo_gemm_nt_x3_splitk_plan(4000, 256, 256) == 1       # short K


def test_weight_kinds_of_a_resnet50():
    """which convolutions of the fp32 ResNet-50 run on the head's kernels: 33 stride-1 1x1, 13 stride-1 3x3 (forward + input
    gradient), 3 + 3 stride-2 layers (forward only); the 7x7 stem stays with the library"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.backbone import ResNet
    from combo_avs_amd.ops import convwrw
    net = ResNet(50)
    kinds = [convwrw.weight_kind(c.weight, c.stride, c.padding) for c in net._conv_list()]
    assert len(kinds) == 53
    assert kinds.count(1) == 33 and kinds.count(3) == 13 and kinds.count(21) == 3 and kinds.count(23) == 3 and kinds.count(0) == 1
    assert kinds[0] == 0
    prev = convwrw.FWD_X3
    convwrw.FWD_X3 = False
    try:
        kinds = [convwrw.weight_kind(c.weight, c.stride, c.padding) for c in net._conv_list()]
    finally:
        convwrw.FWD_X3 = prev
    assert kinds.count(21) == 0 and kinds.count(23) == 0 and kinds.count(0) == 7  # stride-2 layers: all the library's then
    assert convwrw.weight_kind(torch.empty(48, 64, 1, 1), 1, 0) == 0  # < 64 channels
