"""Evaluator metrics of the reference - S4 / MS3 (models/evaluation/sem_seg_evaluation.py:66-137, 219-281) and AVSS
(models/evaluation/sem_seg_evaluation_ss.py:66-118, 212-266) - as pure device-side
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


# ---- AVSS (semantic) metric: per-class IoU / F-score histograms, sem_seg_evaluation_ss.py:66-118 ------------------------------
def batch_miou_fscore(output, target, nclass, T=10, beta2=0.3):
    """`_batch_miou_fscore` (:66-104) for all frames at once on the device: output [BF,C,H,W] scores, target [BF,H,W] class ids
    -> (ious [C], fscores [C], cls_count [C], vid_miou [BF]): the per-frame IoU / F-score of every class summed over the
    frames, the number of frames in which a class has a non-empty union, and the per-frame mean IoU over the classes with
    IoU != 0.  The reference copies every frame to the host and calls torch.histc three times per frame (bins = classes,
    range [1, C]: integer values 1..C fall in bin value - 1, zeros are out of range); here the three histograms of all
    frames are ONE scatter-add each (exact integer counts) and nothing leaves the device."""
    BF = target.shape[0]
    predict = torch.argmax(output, 1) + 1
    tgt = target.long() + 1
    predict = predict * (tgt > 0)
    inter = predict * (predict == tgt)

    def hist(x):  # [BF, C] counts of the values 1..C; anything else is not counted, as histc(min = 1, max = C) drops it:
        # 0 and below (unlabelled) AND ids above C - 1 such as the AVSS ignore label 255 (256 after the + 1)
        h = torch.zeros(BF, nclass + 1, device=x.device, dtype=torch.float32)
        x = x.reshape(BF, -1)
        x = torch.where((x < 0) | (x > nclass), torch.zeros_like(x), x)
        h.scatter_add_(1, x, torch.ones(BF, x.shape[1], device=x.device))
        return h[:, 1:]
    a_i, a_p, a_l = hist(inter), hist(predict), hist(tgt)
    a_u = a_p + a_l - a_i
    iou = a_i / (2.220446049250313e-16 + a_u)
    prec, rec = a_i / a_p, a_i / a_l
    f = torch.nan_to_num((1 + beta2) * prec * rec / (beta2 * prec + rec), nan=0.0, posinf=float("inf"), neginf=float("-inf"))
    vid = iou.sum(1) / (iou != 0).float().sum(1)
    return iou.sum(0), f.sum(0), (a_u != 0).float().sum(0), vid


def calc_color_miou_fscore(pred, target, T=10):
    """:107-118.  pred [BF,C,H,W] (the eval output's `sem_seg` maps stacked), target [BF,H,W]; the reference's softmax over C
    does not move the arg-max and is skipped."""
    return batch_miou_fscore(pred, target, pred.shape[1], T)


class AVSSMeter:
    """What SemSegEvaluator_SS accumulates over `process` calls (:212-230) and reports in `evaluate` (:254-266)."""

    def __init__(self):
        self.batches = []

    def process(self, outputs, gts):
        """outputs: list of {"sem_seg": [K,H,W]} (or a stacked tensor) of the frames of the batch; gts [BF,H,W] class ids"""
        sem = outputs if torch.is_tensor(outputs) else torch.stack([o["sem_seg"] for o in outputs])
        miou, f, cls, _ = calc_color_miou_fscore(sem.float(), gts)
        self.batches.append((miou, f, cls))

    def evaluate(self):
        n = len(self.batches)
        miou_pc = sum(b[0] for b in self.batches) / n
        f_pc = sum(b[1] for b in self.batches) / n
        cls_pc = sum(b[2] for b in self.batches) / n
        miou_pc = torch.nan_to_num(miou_pc / cls_pc, nan=0.0)
        f_pc = torch.nan_to_num(f_pc / cls_pc, nan=0.0)
        return {"mIoU": round(miou_pc.mean().item(), 4), "f_score": round(f_pc.mean().item(), 4),
                "mIoU_noBg": miou_pc[:-1].mean().item(), "f_score_noBg": f_pc[:-1].mean().item()}
