"""SetCriterion / SetCriterion_SS, mirroring models/modeling/criterion.py and
criterion_ss.py: same constructor, same 39 loss keys, same frame-selection rules.

MI355X notes: no `.cuda()` literals (criterion.py:218,225,244 cannot run on CPU); the `num_masks` all-reduce
stays a device tensor (no `.item()` sync, :262-265); all 10 decoder outputs are matched with one D2H copy
(matcher.match_layers); `point_source` injects the random points (tests replay the reference's stream).
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from ..ops.points import point_sample
from .matcher import default_point_source


class LossDict(dict):
    """The reference's loss dict (39 scalar entries) that also carries the losses in vector form: `families` =
    [(keys, tensor[len(keys)])].  Weighting and summing family-wise costs 4 multiplications and 4 sums instead of 39 + 39
    tiny kernels forward and ~100 select/mul backward kernels (meta_arch.MaskFormer.forward, trainer.train_step)."""
    families = None
    total = None


def dice_loss(inputs, targets, num_masks):
    inputs = inputs.sigmoid().flatten(1)
    numerator = 2 * (inputs * targets).sum(-1)
    denominator = inputs.sum(-1) + targets.sum(-1)
    return (1 - (numerator + 1) / (denominator + 1)).sum() / num_masks


def sigmoid_ce_loss(inputs, targets, num_masks):
    return F.binary_cross_entropy_with_logits(inputs, targets, reduction="none").mean(1).sum() / num_masks


def calculate_uncertainty(logits):
    assert logits.shape[1] == 1
    return -(torch.abs(logits))


def get_uncertain_point_coords_with_randomness(coarse_logits, num_points, oversample_ratio, importance_sample_ratio, ps):
    """detectron2 PointRend importance sampling with uncertainty = -|logit| (criterion.py:159-165)."""
    num_boxes = coarse_logits.shape[0]
    num_sampled = int(num_points * oversample_ratio)
    point_coords = ps(num_boxes, num_sampled)
    point_logits = point_sample(coarse_logits, point_coords)
    point_uncertainties = calculate_uncertainty(point_logits)
    num_uncertain_points = int(importance_sample_ratio * num_points)
    num_random_points = num_points - num_uncertain_points
    idx = torch.topk(point_uncertainties[:, 0, :], k=num_uncertain_points, dim=1)[1]
    shift = num_sampled * torch.arange(num_boxes, dtype=torch.long, device=coarse_logits.device)
    idx = idx + shift[:, None]
    point_coords = point_coords.view(-1, 2)[idx.view(-1), :].view(num_boxes, num_uncertain_points, 2)
    if num_random_points > 0:
        point_coords = torch.cat([point_coords, ps(num_boxes, num_random_points)], dim=1)
    return point_coords


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses, num_points, oversample_ratio,
                 importance_sample_ratio, n_frame=5):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict = num_classes, matcher, weight_dict
        self.eos_coef, self.losses = eos_coef, losses
        empty_weight = torch.ones(self.num_classes + 1)
        empty_weight[-1] = self.eos_coef
        self.register_buffer("empty_weight", empty_weight)
        self.num_points, self.oversample_ratio = num_points, oversample_ratio
        self.importance_sample_ratio = importance_sample_ratio
        self.n_frame = n_frame  # hard-coded 5 in the reference (criterion.py:243,284)
        self.point_source = None  # test hook
        # Padded targets (trainer.GraphedTrainStep(pad_targets_to=G)): every frame's target list arrives padded to the same length
        # (zero masks, label 0) and the REAL counts live in this device tensor [F] int32 - the step's shapes no longer depend on how
        # many instances a frame holds, so ONE captured graph serves every batch (AVSS: 1 .. 4 classes per frame).  Matching is
        # unchanged (the device LSAP takes the counts and ignores the padded columns); padded pairs carry weight 0 in the mask
        # losses and write their class target into a dummy query.
        self.padded_counts = None
        # test hook: {"match_src", "match_tgt": int64 [L, Nm], "topk": bool [L, Nm, oversampled points]} - the reference's own
        # Hungarian pairs and importance-sampling top-k sets (tests/golden `*/match_all_*`, `*/topk_bits`) used INSTEAD of the
        # device LSAP's / the selection kernel's, so that a gradient comparison does not hinge on a near-tie
        self.frozen_choices = None
        # test hook: keep this forward's point coordinates (`last_coords`, next to `last_indices`) so that a second run can be
        # given the same discrete choices as frozen_choices = {"match_src", "match_tgt", "coords"}
        self.record_choices = False
        # True: SciPy on the host like the reference (always used when a frame has > 6 instances)
        self.host_lsap = False

    # ---- individual losses -----------------------------------------------------------------------------
    def loss_labels(self, outputs, targets, indices, num_masks):
        src_logits = outputs["pred_logits"].float()
        idx = self._get_src_permutation_idx(indices)
        target_classes_o = torch.cat([t["labels"][J.to(t["labels"].device)] for t, (_, J) in zip(targets, indices)])
        target_classes = torch.full(src_logits.shape[:2], self.num_classes, dtype=torch.int64, device=src_logits.device)
        target_classes[idx] = target_classes_o
        return {"loss_ce": F.cross_entropy(src_logits.transpose(1, 2), target_classes, self.empty_weight)}

    def loss_masks(self, outputs, targets, indices, num_masks):
        ps = self.point_source or default_point_source(outputs["pred_masks"].device)
        src_idx = self._get_src_permutation_idx(indices)
        src_masks = outputs["pred_masks"][src_idx]
        target_masks = torch.cat([t["masks"][J.to(t["masks"].device)] for t, (_, J) in zip(targets, indices)]).to(src_masks)
        src_masks, target_masks = src_masks[:, None], target_masks[:, None]
        with torch.no_grad():
            point_coords = get_uncertain_point_coords_with_randomness(
                src_masks.float(), self.num_points, self.oversample_ratio, self.importance_sample_ratio, ps)
            point_labels = point_sample(target_masks, point_coords).squeeze(1)
        point_logits = point_sample(src_masks, point_coords).squeeze(1)
        return {"loss_mask": sigmoid_ce_loss(point_logits, point_labels, num_masks),
                "loss_dice": dice_loss(point_logits, point_labels, num_masks)}

    @staticmethod
    def _get_src_permutation_idx(indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        src_idx = torch.cat([src for (src, _) in indices])
        return batch_idx, src_idx

    def get_loss(self, loss, outputs, targets, indices, num_masks):
        loss_map = {"labels": self.loss_labels, "masks": self.loss_masks}
        assert loss in loss_map, f"do you really want to compute {loss} loss?"
        return loss_map[loss](outputs, targets, indices, num_masks)

    def get_similarity_loss(self, middle_attn_mask, cosine_weight=None, n_frame=5):
        """criterion.py:208-231: c_f = 1 - cos(m_f, m_{f+1}); sum_f c_f exp(-c_f); / clips / (n_frame-1)."""
        bs, n_query, HW = middle_attn_mask.shape
        bs = bs // n_frame
        m = middle_attn_mask.reshape(bs, n_frame, n_query * HW).float()
        x1, x2 = m[:, :-1], m[:, 1:]
        cos = (x1 * x2).sum(-1) / torch.sqrt(((x1 * x1).sum(-1) + 1e-12) * ((x2 * x2).sum(-1) + 1e-12))
        c = 1 - cos  # [bs, n_frame-1]
        if cosine_weight is None:
            c = c * torch.exp(-c)
        else:
            c = c * cosine_weight[None, : n_frame - 1]
        return {"loss_cosine": c.sum() / bs / (n_frame - 1)}

    # ---- frame selection ---------------------------------------------------------------------------------
    def _select(self, outputs, index):
        sel = {"pred_logits": outputs["pred_logits"].index_select(0, index),
               "pred_masks": outputs["pred_masks"].index_select(0, index)}
        sel["aux_outputs"] = [{k: v.index_select(0, index) for k, v in a.items()} for a in outputs.get("aux_outputs", [])]
        if "middles_attn_mask" in outputs:
            sel["middles_attn_mask"] = outputs["middles_attn_mask"]  # NOT sub-selected (criterion.py:241-254)
        return sel

    def _static_match_index(self, G, L, Gmax, dev):
        """Small index tensors that depend only on the per-frame instance counts: cached on the device."""
        key = (G, L, Gmax, str(dev))
        cache = self.__dict__.setdefault("_match_index_cache", {})
        if key not in cache:
            gcount = torch.tensor(list(G) * L, dtype=torch.int32, device=dev)
            valid = torch.tensor([f * Gmax + g for f in range(len(G)) for g in range(G[f])], dtype=torch.int64, device=dev)
            frame = torch.cat([torch.full((g,), f, dtype=torch.int64) for f, g in enumerate(G)]).to(dev)
            cache[key] = (gcount, valid, frame)
        return cache[key]

    def _fused_index(self, frame_ids, heads, BT, Q, dev):
        """Cached index tensors of the fused path: GT-frame selector, head of criterion layer l (l = 0 is the FINAL
        prediction = last head, l >= 1 is aux output l-1), global frame of GT frame f, and the map index of query 0 of
        every (layer, GT frame) problem inside the [heads, BT, Q] stack."""
        key = ("fused", frame_ids, heads, BT, Q, str(dev))
        cache = self.__dict__.setdefault("_match_index_cache", {})
        if key not in cache:
            hol = [heads - 1] + list(range(heads - 1))
            sel = torch.tensor(list(frame_ids), dtype=torch.int64, device=dev)
            head_of_layer = torch.tensor(hol, dtype=torch.int64, device=dev)
            base = torch.tensor([(h * BT + f) * Q for h in hol for f in frame_ids], dtype=torch.int64, device=dev)
            cache[key] = (sel, head_of_layer, sel, base)
        return cache[key]

    def _num_masks(self, targets, device):
        if getattr(self, "num_masks_override", None) is not None:
            # trainer.GraphedTrainStep: the (all-reduced, clamped) value lives in a static device tensor that is refreshed
            # outside the captured hipGraph (a host->device copy / collective cannot be part of the graph)
            return self.num_masks_override
        if self.padded_counts is not None:  # padded targets: the real counts (a device tensor)
            num_masks = self.padded_counts.sum().to(torch.float).reshape(1)
        else:
            num_masks = torch.as_tensor([sum(len(t["labels"]) for t in targets)], dtype=torch.float, device=device)
        world = 1
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(num_masks)
            world = dist.get_world_size()
        return torch.clamp(num_masks / world, min=1)  # stays on the device: no .item() sync

    def _losses_no_targets(self, outputs, logits, L, F_, Q, dev):
        """No ground-truth instance in the whole batch: every query is "no object", the mask losses are empty sums (0, still
        attached to the graph like the reference's), the cosine term is unchanged."""
        target_classes = torch.full((L * F_ * Q,), self.num_classes, dtype=torch.int64, device=dev)
        nll = F.cross_entropy(logits.view(L * F_ * Q, -1), target_classes, reduction="none").view(L, -1)
        loss_ce = nll.mean(1)  # all weights equal eos_coef: the weighted mean is the plain mean
        zero = (outputs["pred_masks"].sum() * 0.0).expand(L)
        return self._assemble(loss_ce, zero, zero, outputs)

    def _assemble(self, loss_ce, loss_mask, loss_dice, outputs, lc="compute"):
        L = loss_ce.shape[0]
        sfx = [""] + [f"_{l - 1}" for l in range(1, L)]
        losses = LossDict()
        losses.families = []
        for name, vec in (("loss_ce", loss_ce), ("loss_mask", loss_mask), ("loss_dice", loss_dice)):
            keys = [name + s_ for s_ in sfx]
            losses.families.append((keys, vec))
            losses.update(zip(keys, vec.unbind(0)))
        if isinstance(lc, str):
            lc = self._cosine_vector(outputs)
        if lc is not None:
            keys = [f"loss_cosine_{i}" for i in range(lc.shape[0])]
            losses.families.append((keys, lc))
            losses.update(zip(keys, lc.unbind(0)))
        return losses

    def _cosine_vector(self, outputs):
        """frame-to-frame cosine loss on the intermediate mask logits (criterion.py:208-231, 282-286) -> [9] or None"""
        from ..ops import maskloss
        if "middles_attn_mask" in outputs and len(outputs["middles_attn_mask"]):
            mid = torch.stack(outputs["middles_attn_mask"])  # [9,BT,Q,HW]
            n9, bt = mid.shape[0], mid.shape[1]
            nf = self.n_frame
            dot, nrm = maskloss.cosine_stats(mid.reshape(n9 * bt, -1), nf)
            dot, nrm = dot.view(n9, bt // nf, nf), nrm.view(n9, bt // nf, nf)
            cos = dot[..., :-1] / torch.sqrt((nrm[..., :-1] + 1e-12) * (nrm[..., 1:] + 1e-12))
            c = 1 - cos
            return (c * torch.exp(-c)).sum((1, 2)) / (bt // nf) / (nf - 1)
        return None

    def _losses(self, outputs, targets, frame_ids=None):
        """frame_ids (host list of the frames that carry ground truth) + outputs["_logits_all"]: fused path - the mask
        logits of all heads are read from the decoder's single buffer and differentiated by one node.
        All decoder outputs (final + aux) are processed TOGETHER: one batched cost computation + one D2H copy for
        the 10 x F assignment problems, one batched importance-sampling / point-sampling / loss evaluation for the
        10 x (matched masks), one batched cosine loss.  Random points are drawn in the reference's order
        (per output: matcher points frame by frame, then the 3x-oversampled and the extra uniform loss points,
        criterion.py:259-277) so that an injected `point_source` replays the reference exactly."""
        layers = [{"pred_logits": outputs["pred_logits"], "pred_masks": outputs["pred_masks"]}] + list(outputs.get("aux_outputs", []))
        dev = outputs["pred_logits"].device
        ps = self.point_source or default_point_source(dev)
        L, F_ = len(layers), len(targets)
        logits = torch.stack([lo["pred_logits"] for lo in layers]).float()  # [L,F,Q,K+1]
        xall = outputs.get("_logits_all") if frame_ids is not None else None
        fused = (xall is not None and xall.is_cuda and xall.dtype == torch.float32 and xall.is_contiguous()
                 and xall.shape[0] == L and len(frame_ids) == F_)
        if fused:
            heads, BT = xall.shape[0], xall.shape[1]
            sel, head_of_layer, gframe, mask_base = self._fused_index(tuple(frame_ids), heads, BT, xall.shape[2], dev)
            logits = logits.index_select(1, sel)  # class logits of the GT frames (small); the maps are addressed in place
            masks = None
        else:
            masks = torch.stack([lo["pred_masks"] for lo in layers])  # [L,F,Q,h,w]
            if frame_ids is not None and logits.shape[1] != F_:
                # the caller passed all BT frames for the fused path, which does not apply (layout / dtype / head count):
                # pick the ground-truth frames here - never fold 5 frames into the class dimension
                if len(frame_ids) != F_:
                    raise ValueError(f"{len(frame_ids)} ground-truth frame ids for {F_} targets")
                sel = torch.tensor(list(frame_ids), dtype=torch.int64, device=dev)
                logits, masks = logits.index_select(1, sel), masks.index_select(1, sel)
        Q = logits.shape[2]
        G = [int(t["labels"].shape[0]) for t in targets]
        Gmax, Nm = max(G), sum(G)
        if Nm == 0:
            return self._losses_no_targets(outputs, logits, L, F_, Q, dev)
        P = self.num_points
        n_over = int(P * self.oversample_ratio)
        n_unc = int(self.importance_sample_ratio * P)
        n_rand = P - n_unc
        # ---- random points, reference order -----------------------------------------------------------------
        if self.point_source is None:
            # default generator: three launches for all outputs (an injected stream is consumed in the reference's call order
            # below; with torch's own generator there is no stream to stay compatible with)
            mpts = torch.rand(L * F_, P, 2, device=dev)
            over = torch.rand(L * Nm, n_over, 2, device=dev)
            extra = torch.rand(L * Nm, n_rand, 2, device=dev) if n_rand > 0 else None
        else:
            mpts, over, extra = [], [], []
            for _ in range(L):
                mpts.append(torch.cat([ps(1, P) for _ in range(F_)], 0))  # [F,P,2]
                over.append(ps(Nm, n_over))
                if n_rand > 0:
                    extra.append(ps(Nm, n_rand))
            mpts = torch.stack(mpts).view(L * F_, P, 2)
            over = torch.stack(over).view(L * Nm, n_over, 2)
            extra = torch.stack(extra).view(L * Nm, n_rand, 2) if n_rand > 0 else None
        # ---- Hungarian matching: batched costs, one host sync -------------------------------------------------
        H, W = targets[0]["masks"].shape[-2:]
        if all(g == Gmax for g in G) and Gmax > 0:  # equal target counts: two stacks instead of two fills + two copies per frame
            gt = torch.stack([t["masks"] for t in targets]).to(device=dev, dtype=torch.float32)
            lab = torch.stack([t["labels"] for t in targets]).to(device=dev, dtype=torch.int64)
        else:
            gt = torch.zeros(F_, Gmax, H, W, device=dev)
            lab = torch.zeros(F_, Gmax, dtype=torch.int64, device=dev)
            for f, t in enumerate(targets):
                gt[f, : G[f]] = t["masks"].to(gt)
                lab[f, : G[f]] = t["labels"]
        with torch.no_grad():
            if fused:
                C = self.matcher.batched_cost(logits.reshape(L * F_, Q, -1), xall, lab.repeat(L, 1), gt.repeat(L, 1, 1, 1), mpts,
                                              mask_base=mask_base)
            else:
                C = self.matcher.batched_cost(logits.view(L * F_, Q, -1), masks.view(L * F_, Q, *masks.shape[-2:]).float(),
                                              lab.repeat(L, 1), gt.repeat(L, 1, 1, 1), mpts)  # [L*F,Q,Gmax]
        frame = torch.cat([torch.full((g,), f, dtype=torch.int64) for f, g in enumerate(G)])
        pair_w = None  # padded mode: [F * Gmax] weights of the (frame, slot) pairs (1 = a real target, 0 = padding)
        if self.padded_counts is not None:
            if self.frozen_choices is not None or Gmax > self.matcher.LSAP_DEVICE_MAX_G or len(set(G)) != 1:
                raise ValueError("padded targets: every frame must carry the same (padded) number of targets <= the device LSAP's limit")
            pc = self.padded_counts.to(torch.int32)
            _, _, frame = self._static_match_index(tuple(G), L, Gmax, dev)
            slot_ok = torch.arange(Gmax, device=dev)[None, :] < pc[:, None]  # [F, Gmax]
            rfc = self.matcher.solve_device(C, pc.repeat(L)).view(L, F_, Gmax)
            big = torch.where(rfc < 0, torch.full_like(rfc, 1 << 40), rfc)
            rows, order = torch.sort(big, dim=2)  # the real pairs of a frame first, sorted by query (as scipy returns them)
            ok = slot_ok[None].expand(L, F_, Gmax)
            src_q = torch.where(ok, rows, torch.zeros_like(rows)).view(L, F_ * Gmax)
            tgt_g = order.view(L, F_ * Gmax)
            pair_w = slot_ok.reshape(-1).to(torch.float32)
            self.last_indices = (src_q, tgt_g, frame)
        elif self.frozen_choices is not None:
            src_q = self.frozen_choices["match_src"].to(dev)
            tgt_g = self.frozen_choices["match_tgt"].to(dev)
            frame = self._static_match_index(tuple(G), L, Gmax, dev)[2]  # cached on the device: no copy inside a captured step
            assert src_q.shape == (L, Nm) and tgt_g.shape == (L, Nm)
            self.last_indices = (src_q, tgt_g, frame)
        elif Gmax <= self.matcher.LSAP_DEVICE_MAX_G and self.host_lsap is False:
            # exact assignment on the device: the step has no device->host synchronisation
            gcount, valid, frame = self._static_match_index(tuple(G), L, Gmax, dev)
            rfc = self.matcher.solve_device(C, gcount).view(L, F_, Gmax)  # query matched to (frame, target)
            # scipy returns the pairs of a frame sorted by query index; keep that order so that an injected random-point
            # stream lands on the same masks as in the reference
            big = torch.where(rfc < 0, torch.full_like(rfc, 1 << 40), rfc)
            rows, order = torch.sort(big, dim=2)
            src_q = rows.view(L, F_ * Gmax).index_select(1, valid)  # [L,Nm]
            tgt_g = order.view(L, F_ * Gmax).index_select(1, valid)
            self.last_indices = (src_q, tgt_g, frame)
        else:
            C_host = C.cpu().numpy()  # the only device->host sync of the criterion (SciPy LSAP, as in the reference)
            src_q = torch.empty(L, Nm, dtype=torch.int64)
            tgt_g = torch.empty(L, Nm, dtype=torch.int64)
            for l in range(L):
                off = 0
                for f in range(F_):
                    i, j = self.matcher.solve([C_host[l * F_ + f, :, : G[f]]])[0]
                    src_q[l, off:off + G[f]] = i
                    tgt_g[l, off:off + G[f]] = j
                    off += G[f]
            self.last_indices = (src_q, tgt_g, frame)
            src_q, tgt_g, frame = src_q.to(dev), tgt_g.to(dev), frame.to(dev)
        num_masks = self._num_masks(targets, dev)
        # ---- classification loss (criterion.py:121-135), per output ---------------------------------------------
        lidx = torch.arange(L, device=dev)[:, None].expand(L, Nm)
        if pair_w is None:
            target_classes = torch.full((L, F_, Q), self.num_classes, dtype=torch.int64, device=dev)
            target_classes[lidx, frame[None].expand(L, Nm), src_q] = lab[frame[None].expand(L, Nm), tgt_g]
        else:  # padded pairs write into a dummy query Q that is cut away
            target_classes = torch.full((L, F_, Q + 1), self.num_classes, dtype=torch.int64, device=dev)
            q_ce = torch.where(pair_w[None].expand(L, Nm) > 0, src_q, torch.full_like(src_q, Q))
            target_classes[lidx, frame[None].expand(L, Nm), q_ce] = lab[frame[None].expand(L, Nm), tgt_g]
            target_classes = target_classes[..., :Q].contiguous()
        nll = F.cross_entropy(logits.reshape(L * F_ * Q, -1), target_classes.view(-1), reduction="none").view(L, -1)
        wgt = self.empty_weight[target_classes].view(L, -1)
        from ..ops.colsum import row_sum  # (no ATen reduction over >= 2 000 inputs inside the captured step: ops/colsum.py)
        loss_ce = row_sum(nll * wgt) / row_sum(wgt)
        # ---- mask losses (criterion.py:137-186), all outputs at once: three fused launches (csrc/maskloss.hip) ----------
        from ..ops import maskloss
        frame_b = frame[None].expand(L, Nm)
        gt_index = (frame_b * Gmax + tgt_g).reshape(-1).contiguous()
        if fused:
            mask_index = ((head_of_layer[lidx] * BT + gframe[frame_b]) * Q + src_q).reshape(-1).contiguous()
            coords = self._frozen_coords(over, extra, n_unc)
            if coords is None:
                coords = maskloss.uncertain_points(xall.view(-1, *xall.shape[-2:]), mask_index, over, extra, n_unc)
            if self.record_choices:
                self.last_coords = coords
            n_mid = len(outputs.get("middles_attn_mask", []))
            bce, dice, dot, nrm = maskloss.mask_and_cosine(xall, n_mid, self.n_frame, mask_index, gt, gt_index, coords)
            bce, dice = bce.view(L, Nm), dice.view(L, Nm)
            if pair_w is not None:
                bce, dice = bce * pair_w, dice * pair_w
            lc = None
            if n_mid:
                nf = self.n_frame
                dot, nrm = dot.view(n_mid, BT // nf, nf), nrm.view(n_mid, BT // nf, nf)
                c = 1 - dot[..., :-1] / torch.sqrt((nrm[..., :-1] + 1e-12) * (nrm[..., 1:] + 1e-12))
                lc = (c * torch.exp(-c)).sum((1, 2)) / (BT // nf) / (nf - 1)
            return self._assemble(loss_ce, bce.sum(1) / num_masks, dice.sum(1) / num_masks, outputs, lc)
        masks32 = masks.float().contiguous()  # [L,F,Q,h,w]; pairs address their maps by flat index, no gather copies
        mask_index = ((lidx * F_ + frame_b) * Q + src_q).reshape(-1).contiguous()
        coords = self._frozen_coords(over, extra, n_unc)
        if coords is None:
            coords = maskloss.uncertain_points(masks32, mask_index, over, extra, n_unc)  # importance sampling, no sort
        if self.record_choices:
            self.last_coords = coords
        bce, dice = maskloss.mask_losses(masks32, mask_index, gt, gt_index, coords)
        bce, dice = bce.view(L, Nm), dice.view(L, Nm)
        if pair_w is not None:
            bce, dice = bce * pair_w, dice * pair_w
        loss_mask = bce.sum(1) / num_masks
        loss_dice = dice.sum(1) / num_masks
        return self._assemble(loss_ce, loss_mask, loss_dice, outputs)

    def _frozen_coords(self, over, extra, n_unc):
        """test hook (frozen_choices): the injected top-k SETS applied to this step's oversampled points -> [L*Nm, P, 2]"""
        if self.frozen_choices is None:
            return None
        if "coords" in self.frozen_choices:  # the sampled coordinates themselves [L*Nm, P, 2] (a recorded run's `last_coords`)
            return self.frozen_choices["coords"]
        keep = self.frozen_choices["topk"].to(over.device).reshape(over.shape[0], over.shape[1])
        assert bool((keep.sum(1) == n_unc).all())
        coords = over[keep].view(over.shape[0], n_unc, 2)
        return (torch.cat([coords, extra], 1) if extra is not None else coords).contiguous()

    def forward(self, outputs, targets):
        bt = len(outputs["pred_logits"])
        s4 = bt != len(targets)  # S4 training: GT on the first frame of each clip only
        if "_logits_all" in outputs and outputs["_logits_all"].is_cuda:
            return self._losses(outputs, targets, frame_ids=list(range(0, bt, 5)) if s4 else list(range(bt)))
        if s4:
            index = torch.arange(0, bt, 5, device=outputs["pred_logits"].device)
            outputs = self._select(outputs, index)
        return self._losses(outputs, targets)

    def __repr__(self):
        body = [f"matcher: {self.matcher.__repr__(_repr_indent=8)}", f"losses: {self.losses}",
                f"weight_dict: {self.weight_dict}", f"num_classes: {self.num_classes}", f"eos_coef: {self.eos_coef}",
                f"num_points: {self.num_points}", f"oversample_ratio: {self.oversample_ratio}",
                f"importance_sample_ratio: {self.importance_sample_ratio}"]
        return "\n".join(["Criterion " + self.__class__.__name__] + [" " * 4 + line for line in body])


class SetCriterion_SS(SetCriterion):
    """AVSS variant: frames are chosen by gt_temporal_mask_flag (criterion_ss.py:246-257)."""

    def forward(self, outputs, targets, vid_temporal_mask_flag, gt_temporal_mask_flag, gt_index=None):
        """gt_index: the rows `torch.where(gt_temporal_mask_flag == 1)[0]` as a ready device tensor (a captured step: the flag
        values were read on the host before the graph launch, trainer.GraphedTrainStep)"""
        index = gt_index if gt_index is not None else torch.where(gt_temporal_mask_flag == 1)[0].to(outputs["pred_logits"].device)
        return self._losses(self._select(outputs, index), targets)
