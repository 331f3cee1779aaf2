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
REF_SS = "/root/reference/models/evaluation/sem_seg_evaluation_ss.py"
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


def load_reference_ss():
    """the AVSS evaluator module (same third-party stubs; none is touched by the metric functions)"""
    spec = importlib.util.spec_from_file_location("ref_sem_seg_evaluation_ss", REF_SS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_avss(out_path):
    """Round 3: the AVSS metric - `calc_color_miou_fscore` / `_batch_miou_fscore` (sem_seg_evaluation_ss.py:66-118) on two
    seeded 10-frame clips with K = 8 classes (some never present, some predicted but absent, one frame all background), and
    the aggregation of SemSegEvaluator_SS.evaluate (:254-266) over the two `process` calls."""
    R = load_reference_ss()
    g = torch.Generator().manual_seed(11)
    K, T, H, W = 8, 10, 24, 32
    out = {}
    sums = [torch.zeros(K), torch.zeros(K), torch.zeros(K)]
    for clip in range(2):
        tgt = torch.randint(0, 5, (T, H, W), generator=g)  # classes 5..7 never in the ground truth
        tgt[:, :, W // 2:] = torch.randint(0, 2, (T, H, W // 2), generator=g) * (3 + clip)
        tgt[4] = 0  # one frame: background only
        logits = torch.randn(T, K, H, W, generator=g)
        logits.scatter_add_(1, tgt[:, None], torch.full((T, 1, H, W), 1.5))  # mostly right, often wrong
        logits[:, 6] += 0.8 * (torch.rand(T, H, W, generator=g) > 0.8)     # a class that is predicted but never present
        logits[:, 7] -= 100.0        # never predicted, never present: empty union in every frame (0 / 0 in `evaluate`)
        logits[1::2, 5] -= 100.0     # predicted (wrongly) in the even frames only: cls_count 5
        if clip == 1:                # round 4: the ignore label the reference registers for AVSS (255 -> 256 after the + 1:
            tgt[:, :3, :] = 255      # above histc's range, i.e. in NO class's ground-truth area; the prediction there counts)
        miou, fscore, cls, vid = R.calc_color_miou_fscore(logits, tgt, T=T)
        out[f"ss{clip}/logits"] = logits.numpy().astype(np.float32)
        out[f"ss{clip}/target"] = tgt.numpy().astype(np.int64)
        out[f"ss{clip}/miou"] = miou.numpy().astype(np.float64)
        out[f"ss{clip}/fscore"] = fscore.numpy().astype(np.float64)
        out[f"ss{clip}/cls_count"] = cls.numpy().astype(np.float64)
        out[f"ss{clip}/vid_miou"] = np.array([float(v) for v in vid], dtype=np.float64)
        for acc, v in zip(sums, (miou, fscore, cls)):
            acc += v
    # SemSegEvaluator_SS.evaluate, single process (:254-266): the AverageMeter means of the per-batch vectors
    miou_pc, f_pc, cls_pc = (v / 2 for v in sums)
    miou_pc = miou_pc / cls_pc
    miou_pc[torch.isnan(miou_pc)] = 0
    f_pc = f_pc / cls_pc
    f_pc[torch.isnan(f_pc)] = 0
    out["ss/mIoU"] = np.float64(round(torch.mean(miou_pc).item(), 4))
    out["ss/f_score"] = np.float64(round(torch.mean(f_pc).item(), 4))
    out["ss/mIoU_noBg"] = np.float64(torch.mean(miou_pc[:-1]).item())
    np.savez_compressed(out_path, **out)
    print({k: (float(v) if v.ndim == 0 else v.shape) for k, v in out.items() if "logits" not in k and "target" not in k})


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
    gen_avss(os.path.join(HERE, "eval_metric_ss.npz"))


if __name__ == "__main__":
    main()
