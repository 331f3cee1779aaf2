"""Evaluator metric (reference: models/evaluation/sem_seg_evaluation.py:66-137, 219-245) against golden vectors produced by the
reference's own functions (tests/golden/gen_golden_eval.py -> eval_metric.npz): the oracle restatement on CPU, the product's
device functions on the GPU - including the empty-ground-truth rule and the evaluator's second softmax."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden", "eval_metric.npz")


def _cases():
    z = np.load(G)
    return z


def test_oracle_metric_matches_reference():
    from oracle import combo_oracle as O
    z = _cases()
    for c in ("a", "b"):
        pred, gt = torch.from_numpy(z[f"{c}/pred"]), torch.from_numpy(z[f"{c}/gt"])
        assert abs(float(O.mask_iou(pred, gt)) - float(z[f"{c}/miou"])) < 1e-6
        assert abs(O.eval_fmeasure(pred, gt) - float(z[f"{c}/fscore"])) < 1e-6
    miou, f = O.s4_clip_metrics(torch.from_numpy(z["c/sem_seg"]), torch.from_numpy(z["c/gt"]))
    assert abs(miou - float(z["c/miou"])) < 1e-6 and abs(f - float(z["c/fscore"])) < 1e-6
    assert 0.05 < float(z["a/miou"]) < 0.95  # the fixture is not degenerate


@pytest.mark.gpu
def test_device_metric_matches_reference_and_scores_the_fused_inference_tail():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import evaluation as E
    from combo_avs_amd.ops import infer
    z = _cases()
    for c in ("a", "b"):
        pred, gt = torch.from_numpy(z[f"{c}/pred"]).cuda(), torch.from_numpy(z[f"{c}/gt"]).cuda()
        assert abs(float(E.mask_iou(pred, gt)) - float(z[f"{c}/miou"])) < 1e-6
        assert abs(E.eval_fmeasure(pred, gt) - float(z[f"{c}/fscore"])) < 1e-6
    sem, gt = torch.from_numpy(z["c/sem_seg"]).cuda(), torch.from_numpy(z["c/gt"]).cuda()
    miou, f = E.s4_clip_metrics([{"sem_seg": s} for s in sem], gt)
    assert abs(miou - float(z["c/miou"])) < 1e-6 and abs(f - float(z["c/fscore"])) < 1e-6
    # end to end on the product's eval output: the fused upsample + sigmoid + class-mix kernel (csrc/infer.hip) feeds the metric;
    # the same numbers must come out of the oracle's inference tail + metric on the CPU
    from oracle import combo_oracle as O
    torch.manual_seed(3)
    cls = torch.randn(5, 100, 3)
    masks = torch.randn(5, 100, 56, 56) * 3
    yy, xx = torch.meshgrid(torch.arange(224), torch.arange(224), indexing="ij")
    gts = torch.stack([(((xx - 90 - 5 * i) ** 2 + (yy - 100) ** 2) < (30 + 2 * i) ** 2).float() for i in range(5)])
    sem_gpu = infer.semantic_inference(cls.cuda(), masks.cuda(), (224, 224))
    m_gpu, f_gpu = E.s4_clip_metrics(sem_gpu, gts.cuda())
    sem_cpu = O.semantic_inference(cls, masks, (224, 224))
    m_cpu, f_cpu = O.s4_clip_metrics(sem_cpu, gts)
    assert abs(m_gpu - m_cpu) < 1e-4 and abs(f_gpu - f_cpu) < 1e-4


# ---- AVSS metric (sem_seg_evaluation_ss.py:66-118, 254-266), vectors from the reference's own functions ------------------------
G_SS = os.path.join(os.path.dirname(__file__), "golden", "eval_metric_ss.npz")


def _check_ss(calc, evaluate, dev):
    z = np.load(G_SS)
    batches = []
    for clip in range(2):
        logits = torch.from_numpy(z[f"ss{clip}/logits"]).to(dev)
        tgt = torch.from_numpy(z[f"ss{clip}/target"]).to(dev)
        miou, f, cls, vid = calc(logits, tgt)
        np.testing.assert_allclose(miou.cpu().numpy(), z[f"ss{clip}/miou"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(f.cpu().numpy(), z[f"ss{clip}/fscore"], rtol=1e-6, atol=1e-7)
        assert np.array_equal(cls.cpu().numpy(), z[f"ss{clip}/cls_count"])  # integer counts: exact
        np.testing.assert_allclose(np.array([float(v) for v in vid]), z[f"ss{clip}/vid_miou"], rtol=1e-6)
        batches.append((miou, f, cls))
        assert (z[f"ss{clip}/cls_count"] == 0).any() and (z[f"ss{clip}/cls_count"] == 10).any()  # absent and always-present classes
    res = evaluate(batches)
    assert res["mIoU"] == float(z["ss/mIoU"]) and res["f_score"] == float(z["ss/f_score"])
    assert abs(res["mIoU_noBg"] - float(z["ss/mIoU_noBg"])) < 1e-6


def test_oracle_avss_metric_matches_reference():
    from oracle import combo_oracle as O
    _check_ss(O.avss_calc_color_miou_fscore, O.avss_evaluate, "cpu")


def test_product_avss_metric_host_logic_matches_reference():
    """the product's device-side formulation (one scatter-add per histogram) is plain torch: checked on the CPU here, on the
    GPU below"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import evaluation as E

    def evaluate(batches):
        m = E.AVSSMeter()
        m.batches = batches
        return m.evaluate()
    _check_ss(E.calc_color_miou_fscore, evaluate, "cpu")


@pytest.mark.gpu
def test_device_avss_metric_matches_reference():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import evaluation as E
    z = np.load(G_SS)
    m = E.AVSSMeter()
    for clip in range(2):
        sem = torch.from_numpy(z[f"ss{clip}/logits"]).cuda()
        m.process([{"sem_seg": s} for s in sem], torch.from_numpy(z[f"ss{clip}/target"]).cuda())
    res = m.evaluate()
    assert res["mIoU"] == float(z["ss/mIoU"]) and res["f_score"] == float(z["ss/f_score"])

    def evaluate(batches):
        mm = E.AVSSMeter()
        mm.batches = batches
        return mm.evaluate()
    _check_ss(E.calc_color_miou_fscore, evaluate, "cuda")
