#!/usr/bin/env python3
"""Test tooling (build container only): golden vectors for the S4 / MS3 evaluator metric.
Imports the reference's models/evaluation/sem_seg_evaluation.py (third-party names it imports at module level - pycocotools,
detectron2.data / .utils / .evaluation - are stubbed: none of them is touched by the metric functions), calls its
`mask_iou`, `Eval_Fmeasure` and the softmax step of `SemSegEvaluator.process` (:243-245) on seeded inputs and writes
tests/golden/eval_metric.npz.  Cases: random probabilities, a frame with empty ground truth (the "no object" rule of
mask_iou :83-89 and the skip rule of Eval_Fmeasure :127), all-background predictions, and a full clip in the evaluator's input
form (K = 2 `sem_seg` maps per frame -> softmax over K -> channel 1)."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/models/evaluation/sem_seg_evaluation.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    if not os.path.exists(REF):
        sys.exit("reference checkout needed")
    stub("pycocotools")
    stub("pycocotools.mask")
    stub("detectron2")
    stub("detectron2.data", DatasetCatalog=object(), MetadataCatalog=object())
    stub("detectron2.utils")
    stub("detectron2.utils.comm", all_gather=None, is_main_process=None, synchronize=None)
    stub("detectron2.utils.file_io", PathManager=object())
    stub("detectron2.evaluation")
    stub("detectron2.evaluation.evaluator", DatasetEvaluator=object)
    torch.Tensor.cuda = lambda self, *a, **k: self  # the metric code calls .cuda() unconditionally (_eval_pr)
    spec = importlib.util.spec_from_file_location("ref_sem_seg_evaluation", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    R = load_reference()
    g = torch.Generator().manual_seed(7)
    out = {}
    # case A: 5 frames, random probabilities, blob targets, frame 3 with EMPTY ground truth
    H = W = 56
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    gt = torch.stack([(((xx - 20 - 3 * i) ** 2 + (yy - 25) ** 2) < (8 + i) ** 2).float() for i in range(5)])
    gt[3] = 0
    pred = (0.3 * gt + 0.7 * torch.rand(5, H, W, generator=g)).clamp(0, 1)
    pred[3] = 0.6 * torch.rand(H, W, generator=g)
    out["a/pred"], out["a/gt"] = pred.numpy(), gt.numpy()
    out["a/miou"] = np.float64(float(R.mask_iou(pred, gt)))
    out["a/fscore"] = np.float64(R.Eval_Fmeasure(pred, gt))
    # case B: all-background prediction against a non-empty target
    pred_b = torch.zeros(2, H, W)
    gt_b = gt[:2].clone()
    out["b/pred"], out["b/gt"] = pred_b.numpy(), gt_b.numpy()
    out["b/miou"] = np.float64(float(R.mask_iou(pred_b, gt_b)))
    out["b/fscore"] = np.float64(R.Eval_Fmeasure(pred_b, gt_b))
    # case C: the evaluator's input form: per frame `sem_seg` [K = 2, H, W] (already class-probability mixes of the
    # inference tail) -> softmax over K AGAIN (sem_seg_evaluation.py:243) -> channel 1 (:244-245)
    sem = torch.rand(5, 2, H, W, generator=g) * 3.0
    sem[:, 1] += 2.0 * gt
    probs = torch.nn.functional.softmax(sem, dim=1)
    out["c/sem_seg"], out["c/gt"] = sem.numpy(), gt.numpy()
    out["c/miou"] = np.float64(float(R.mask_iou(probs[:, 1], gt)))
    out["c/fscore"] = np.float64(R.Eval_Fmeasure(probs[:, 1], gt))
    np.savez_compressed(os.path.join(HERE, "eval_metric.npz"), **out)
    print({k: float(v) for k, v in out.items() if v.ndim == 0})


if __name__ == "__main__":
    main()
