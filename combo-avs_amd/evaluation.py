"""S4 / MS3 evaluator metric of the reference (models/evaluation/sem_seg_evaluation.py:66-137, 219-281) as pure device-side
functions - what `SemSegEvaluator.process` computes per batch from the model's eval output, so that mIoU / F-score can be
checked on the GPU box without detectron2's evaluator plumbing.  All reductions stay on the device; one
host read per metric."""
import torch
import torch.nn.functional as F


def mask_iou(pred, target, eps=1e-7):
    """pred [N,H,W] probabilities (thresholded at 0.5), target [N,H,W] 0/1 -> scalar tensor.  Frames with empty ground truth
    score the background agreement over all pixels (sem_seg_evaluation.py:83-89)."""
    assert pred.dim() == 3 and pred.shape == target.shape
    n = pred.shape[0]
    target = target.to(torch.float32)
    p = (pred > 0.5).to(torch.float32)
    pixels = float(pred.shape[-1] * pred.shape[-2])
    empty = target.sum((1, 2)) == 0
    inter = torch.where(empty, ((1 - target) * (1 - p)).sum((1, 2)), (p * target).sum((1, 2)))
    union = torch.where(empty, torch.full((n,), pixels, device=pred.device), torch.max(p, target).sum((1, 2)))
    return (inter / (union + eps)).sum() / n


def eval_fmeasure(pred, gt, pr_num=255):
    """Maximum over `pr_num` thresholds of the F-beta curve (beta^2 = 0.3) averaged over frames with non-empty ground truth
    (:95-137); all thresholds of a frame are evaluated in one batched comparison instead of a 255-iteration Python loop."""
    gt = gt.to(torch.float32)
    keep = gt.flatten(1).mean(1) != 0
    if not bool(keep.any()):
        return 0.0
    pred, gt = pred[keep], gt[keep]
    th = torch.linspace(0, 1 - 1e-10, pr_num, device=pred.device)
    curves = []
    for i in range(pred.shape[0]):  # [pr_num, H, W] per frame keeps the working set small
        y = (pred[i][None] >= th[:, None, None]).to(torch.float32)
        tp = (y * gt[i][None]).sum((1, 2))
        prec, rec = tp / (y.sum((1, 2)) + 1e-20), tp / (gt[i].sum() + 1e-20)
        curves.append(torch.nan_to_num(1.3 * prec * rec / (0.3 * prec + rec), nan=0.0))
    return float(torch.stack(curves).mean(0).max())


def s4_clip_metrics(outputs, gts):
    """outputs: the meta-architecture's eval result (list of {"sem_seg": [K,H,W]}, or a stacked [N,K,H,W] tensor) for the
    frames of a batch; gts [N,H,W] 0/1.  Applies the evaluator's own softmax over K (:243) and scores channel 1."""
    sem = outputs if torch.is_tensor(outputs) else torch.stack([o["sem_seg"] for o in outputs])
    probs = F.softmax(sem.float(), dim=1)[:, 1]
    return float(mask_iou(probs, gts)), eval_fmeasure(probs, gts)


class AverageMeter:
    """the evaluator's running mean over batches (sem_seg_evaluation.py:37-63)"""

    def __init__(self):
        self.sum, self.n = {}, {}

    def add(self, values):
        for k, v in values.items():
            self.sum[k] = self.sum.get(k, 0.0) + float(v)
            self.n[k] = self.n.get(k, 0) + 1

    def mean(self, key):
        return self.sum[key] / max(self.n.get(key, 0), 1)
