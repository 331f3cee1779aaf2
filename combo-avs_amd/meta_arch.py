"""`MaskFormer` meta-architecture (mirrors models/maskformer_model.py:28-471): same registry name, same
`forward(batched_inputs) -> loss dict | list[{"sem_seg"}]` contract and the same state-dict layout
(backbone.*, pre_sam_backbone.*, scale_factor_module.N.*, audio_backbone.*, sem_seg_head.*, criterion.empty_weight).

`instances` may be detectron2 `Instances` (anything with .gt_classes/.gt_masks) or plain dicts with those keys.
"""
import os
from typing import Tuple

import contextlib

import torch
from torch import nn
from torch.nn import functional as F

from . import backbone_pvt  # noqa: F401  (registers build_pvtv2_b5_backbone)
from .backbone import VGGish
from .modeling.criterion import SetCriterion, SetCriterion_SS
from .modeling.head import MaskFormerHead
from .modeling.matcher import HungarianMatcher
from .modeling.semmix import channel_weighted_block, sem_mix
from .registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY


def build_backbone(cfg):
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, None)


def build_sem_seg_head(cfg, input_shape):
    cls = SEM_SEG_HEADS_REGISTRY.get(cfg.MODEL.SEM_SEG_HEAD.NAME)
    return cls(**cls.from_config(cfg, input_shape))


def _field(inst, name):
    return inst[name] if isinstance(inst, dict) else getattr(inst, name)


@META_ARCH_REGISTRY.register()
class MaskFormer(nn.Module):
    def __init__(self, *, backbone, use_pre_sam: bool, pre_sam_backbone, scale_factor_module, audio_backbone,
                 audio_transformation, sem_seg_head, fusion_module, criterion, is_avss_data: bool, num_queries: int,
                 object_mask_threshold: float, overlap_threshold: float, metadata, size_divisibility: int,
                 sem_seg_postprocess_before_inference: bool, pixel_mean: Tuple[float], pixel_std: Tuple[float],
                 semantic_on: bool, panoptic_on: bool, instance_on: bool, test_topk_per_image: int):
        super().__init__()
        self.backbone = backbone
        self.use_pre_sam = use_pre_sam
        if self.use_pre_sam:
            self.pre_sam_backbone = pre_sam_backbone
            self.scale_factor_module = scale_factor_module
        else:
            self.pre_sam_backbone = None
        self.audio_backbone = audio_backbone
        self.sem_seg_head = sem_seg_head
        if fusion_module is not None:
            raise NotImplementedError("FUSION_STEP 'early' is not used by any shipped config; only late fusion is built")
        self.early_fusion = False
        self.criterion = criterion
        self.is_avss_data = is_avss_data
        self.num_queries = num_queries
        self.overlap_threshold, self.object_mask_threshold = overlap_threshold, object_mask_threshold
        self.metadata = metadata
        if size_divisibility < 0:
            size_divisibility = self.backbone.size_divisibility
        self.size_divisibility = size_divisibility
        self.sem_seg_postprocess_before_inference = sem_seg_postprocess_before_inference
        self.register_buffer("pixel_mean", torch.Tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.Tensor(pixel_std).view(-1, 1, 1), False)
        self.semantic_on, self.instance_on, self.panoptic_on = semantic_on, instance_on, panoptic_on
        self.test_topk_per_image = test_topk_per_image
        self.backbone_dtype = torch.float32  # bench/trainer may switch the host-PyTorch backbones to bf16
        if not self.semantic_on:
            assert self.sem_seg_postprocess_before_inference

    @classmethod
    def from_config(cls, cfg):
        backbone = build_backbone(cfg)
        use_pre_sam = cfg.MODEL.PRE_SAM.USE_PRE_SAM
        if use_pre_sam:
            pre_sam_backbone = build_backbone(cfg)
            scale_factor_module = nn.ModuleList([channel_weighted_block(d) for d in cfg.MODEL.PRE_SAM.PRE_SAM_DIM])
        else:
            pre_sam_backbone = scale_factor_module = None
        audio_backbone = VGGish(cfg)
        if cfg.MODEL.AUDIO.FREEZE_AUDIO_EXTRACTOR:
            for p in audio_backbone.parameters():
                p.requires_grad = False
        sem_seg_head = build_sem_seg_head(cfg, backbone.output_shape())
        if cfg.MODEL.FUSE_CONFIG.FUSION_STEP == "early":
            raise NotImplementedError("FUSION_STEP 'early' is not used by any shipped config")
        mf = cfg.MODEL.MASK_FORMER
        matcher = HungarianMatcher(cost_class=mf.CLASS_WEIGHT, cost_mask=mf.MASK_WEIGHT, cost_dice=mf.DICE_WEIGHT,
                                   num_points=mf.TRAIN_NUM_POINTS)
        weight_dict = {"loss_ce": mf.CLASS_WEIGHT, "loss_mask": mf.MASK_WEIGHT, "loss_dice": mf.DICE_WEIGHT,
                       "loss_cosine": mf.COSINE_WEIGHT}
        if mf.DEEP_SUPERVISION:
            aux = {}
            for i in range(mf.DEC_LAYERS - 1):
                aux.update({k + f"_{i}": v for k, v in weight_dict.items()})
            weight_dict.update(aux)
        is_avss_data = cfg.INPUT.DATASET_MAPPER_NAME == "avss_semantic"
        crit_cls = SetCriterion_SS if is_avss_data else SetCriterion
        criterion = crit_cls(sem_seg_head.num_classes, matcher=matcher, weight_dict=weight_dict, eos_coef=mf.NO_OBJECT_WEIGHT,
                             losses=["labels", "masks"], num_points=mf.TRAIN_NUM_POINTS,
                             oversample_ratio=mf.OVERSAMPLE_RATIO, importance_sample_ratio=mf.IMPORTANCE_SAMPLE_RATIO)
        return dict(
            backbone=backbone, use_pre_sam=use_pre_sam, pre_sam_backbone=pre_sam_backbone, audio_backbone=audio_backbone,
            scale_factor_module=scale_factor_module, audio_transformation=None, sem_seg_head=sem_seg_head,
            fusion_module=None, criterion=criterion, is_avss_data=is_avss_data, num_queries=mf.NUM_OBJECT_QUERIES,
            object_mask_threshold=mf.TEST.OBJECT_MASK_THRESHOLD, overlap_threshold=mf.TEST.OVERLAP_THRESHOLD,
            metadata=None, size_divisibility=mf.SIZE_DIVISIBILITY,
            sem_seg_postprocess_before_inference=(mf.TEST.SEM_SEG_POSTPROCESSING_BEFORE_INFERENCE or mf.TEST.PANOPTIC_ON
                                                  or mf.TEST.INSTANCE_ON),
            pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD, semantic_on=mf.TEST.SEMANTIC_ON,
            instance_on=mf.TEST.INSTANCE_ON, panoptic_on=mf.TEST.PANOPTIC_ON,
            test_topk_per_image=cfg.TEST.DETECTIONS_PER_IMAGE)

    @property
    def device(self):
        return self.pixel_mean.device

    # The Siam pair of encoders is independent until the SEM mix: the second one runs on its own HIP stream (fork / join with events,
    # also inside a captured hipGraph; autograd replays each backward pass on its forward stream, so the two backward chains run
    # side by side as well).  Round 1 (library convolutions): 67.4 vs 66.7 ms - no gain, opt-in.  Round 6 (own persistent kernels,
    # ~200 launches of 10 - 60 us per encoder and direction): 46.0 -> 43.9 ms per step, same box, A B A B
    # (profiles/r06_ab_parallel_backbones.txt) - a launch's fixed costs (dispatch, ring priming, drain of the last stores: ~8 us)
    # overlap with the other chain's streaming.  Giving each chain HALF of the CUs (ops.convwrw.backbone_cus, csrc/abi.hip
    # combo_set_cu_limit) measured the same as full-size launches (43.90 vs 43.87 ms): off.
    # ONLY for backbones marked `concurrent_safe` (backbone.ResNet): the PVTv2 encoders run library GEMMs, and the solutions
    # TunableOp / hipBLASLt pick for them include stream-K kernels (`..._SK3_...`), whose workgroups spin-wait for each other - two
    # of those on two streams each hold part of the chip waiting for workgroups that cannot be scheduled: the replayed graph of
    # `pvt_ms3_t10` never finished (tools/run_with_dump.py: torch.cuda.synchronize after the first replay).  PVT stays on one stream.
    parallel_backbones = True
    parallel_backbones_split = False  # (with parallel_backbones) half of the CUs per encoder
    # VGGish (no gradient) on a third stream: 44.35 -> 44.05 ms (profiles/r06_ab_parallel_backbones.txt).  Measured and NOT kept: the
    # head's deferred weight gradients launched on a side stream at the head / backbone boundary of the backward pass instead of as a
    # serial tail after it (44.35 vs 44.40 ms: two throughput-bound streams gain nothing from running side by side)
    parallel_audio = True

    def _loss_weights(self, keys, device):
        cache = self.__dict__.setdefault("_loss_weight_cache", {})
        ck = (tuple(keys), str(device))
        if ck not in cache:
            cache[ck] = torch.tensor([float(self.criterion.weight_dict[k]) for k in keys], dtype=torch.float32, device=device)
        return cache[ck]

    def _side_stream(self, device, index=0):
        streams = self.__dict__.setdefault("_side_streams", {})
        key = (str(device), index)
        if key not in streams:
            streams[key] = torch.cuda.Stream(device=device)
        return streams[key]

    def _pad(self, x):
        """ImageList.from_tensors: zero-pad H, W up to a multiple of size_divisibility (all frames share a size)."""
        d = self.size_divisibility
        if d > 1:
            h, w = x.shape[-2:]
            ph, pw = (d - h % d) % d, (d - w % d) % d
            if ph or pw:
                x = F.pad(x, (0, pw, 0, ph))
        return x

    def forward(self, batched_inputs):
        dev = self.device
        # (a captured AVSS step: the flag values were read before the capture and arrive as constant index tensors - the flag
        #  tensors themselves, CPU tensors when they come straight from the dataset mapper, are not touched inside the capture)
        avss_index = getattr(self, "avss_static_index", None) if (self.is_avss_data and self.training) else None
        vid_flag = gt_flag = None
        if self.is_avss_data and avss_index is None:
            vid_flag = torch.cat([b["vid_temporal_mask_flag"] for b in batched_inputs], dim=0).to(dev)
            gt_flag = torch.cat([b["gt_temporal_mask_flag"] for b in batched_inputs], dim=0).to(dev)
        images = torch.cat([b["images"].to(dev, non_blocking=True) for b in batched_inputs], dim=0)
        image_size = tuple(images.shape[-2:])
        audio_log_mels = torch.cat([b["audio_log_mel"].to(dev, non_blocking=True) for b in batched_inputs], dim=0)
        # (x - mean) / std as in maskformer_model.py:324-325; uint8 - fp32 promotes inside ONE kernel, the division is in place
        images = self._pad(images.sub(self.pixel_mean).div_(self.pixel_std) if images.dtype == torch.uint8
                           else (images.float() - self.pixel_mean) / self.pixel_std)
        amp = torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.backbone_dtype == torch.bfloat16)
        audio_stream = None
        side_ok = images.is_cuda and getattr(self.backbone, "concurrent_safe", False)  # (see parallel_backbones)
        if side_ok and self.parallel_audio:
            audio_stream = self._side_stream(images.device, 1)
            audio_stream.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), amp, (torch.cuda.stream(audio_stream) if audio_stream is not None else contextlib.nullcontext()):
            if audio_stream is not None:
                audio_log_mels.record_stream(audio_stream)
            audio_feature = self.audio_backbone(audio_log_mels).float()  # :327-328
        audio_feature = audio_feature.unsqueeze(1)
        if self.is_avss_data:
            if audio_stream is not None:
                torch.cuda.current_stream().wait_stream(audio_stream)
                audio_feature.record_stream(torch.cuda.current_stream())
                audio_stream = None
            # maskformer_model.py:330-331: the audio rows of the frames that exist.  Boolean indexing reads the flag VALUES on the
            # host (a synchronisation: not capturable); trainer.GraphedTrainStep reads them once per step before the graph
            # launch, keys its graphs by them and hands over the same selection as constant index tensors
            audio_feature = audio_feature.index_select(0, avss_index[0]) if avss_index is not None else audio_feature[vid_flag.bool()]
        if self.use_pre_sam:
            pre = torch.cat([b["pre_masks"].to(dev, non_blocking=True) for b in batched_inputs], dim=0)
            pre = self._pad(pre.sub(self.pixel_mean).div_(self.pixel_std) if pre.dtype == torch.uint8
                            else (pre.float() - self.pixel_mean) / self.pixel_std)
            if side_ok and self.parallel_backbones:
                # the Siam pair is independent until the SEM mix: run the second encoder on its own HIP stream (fork/join
                # with events, also inside a captured hipGraph).  Autograd replays each backward on its forward stream, so
                # the two backward passes overlap as well; the late stages (14x14, 7x7 maps) do not fill 256 CUs alone.
                cur, side = torch.cuda.current_stream(), self._side_stream(images.device)
                side.wait_stream(cur)
                # round 6: each encoder's persistent GEMM launches take HALF of the CUs (ops.convwrw.backbone_cus): full-size
                # launches of the two streams only queue behind each other (measured in round 4: 50.87 vs 50.89 ms)
                from .ops.convwrw import backbone_cus
                half = torch.cuda.get_device_properties(images.device).multi_processor_count // 2 if self.parallel_backbones_split else 0
                with torch.cuda.stream(side), backbone_cus(half):
                    pre.record_stream(side)
                    with amp:
                        pre_sam_features = self.pre_sam_backbone(pre)
                with amp, backbone_cus(half):
                    features = self.backbone(images)
                cur.wait_stream(side)
                for v in pre_sam_features.values():
                    v.record_stream(cur)
            else:
                with amp:
                    features = self.backbone(images)
                    pre_sam_features = self.pre_sam_backbone(pre)
            features = sem_mix(features, pre_sam_features, self.scale_factor_module)  # :345-352
        else:
            with amp:
                features = self.backbone(images)
        # the head's inputs: where a data-parallel trainer cuts the backward pass in two (the head's gradients are complete - and
        # on their way through the all-reduce - while the backbones' backward still runs; trainer.FlatAdamW.backward_early)
        # (recorded only on request: holding them keeps the step's autograd graph alive beyond the backward pass)
        self._head_inputs = ([v for v in features.values() if torch.is_tensor(v) and v.requires_grad]
                             if self.training and getattr(self, "record_head_inputs", False) else None)
        if audio_stream is not None:
            torch.cuda.current_stream().wait_stream(audio_stream)
            audio_feature.record_stream(torch.cuda.current_stream())
        outputs = self.sem_seg_head(features, audio_feature)
        if self.training:
            if "instances" not in batched_inputs[0]:
                raise ValueError("MaskFormer requires `instances` in training!")
            gt_instances = [inst for b in batched_inputs for inst in b["instances"]]
            targets = self.prepare_targets(gt_instances, images)
            if self.is_avss_data:
                losses = self.criterion(outputs, targets, vid_flag, gt_flag, gt_index=None if avss_index is None else avss_index[1])
            else:
                losses = self.criterion(outputs, targets)
            if getattr(losses, "families", None):
                # family-wise weighting (4 multiplications + 4 sums); same 39 weighted entries, plus `.total`
                from .modeling.criterion import LossDict
                weighted, total = LossDict(), None
                weighted.families = []
                for keys, vec in losses.families:
                    for k in keys:
                        if k not in self.criterion.weight_dict:
                            raise ValueError(f"Found useless Loss! {k}")
                    wv = vec * self._loss_weights(keys, vec.device)
                    weighted.families.append((keys, wv))
                    weighted.update(zip(keys, wv.unbind(0)))
                    total = wv.sum() if total is None else total + wv.sum()
                weighted.total = total
                return weighted
            for k in list(losses.keys()):
                if k in self.criterion.weight_dict:
                    losses[k] = losses[k] * self.criterion.weight_dict[k]
                else:
                    losses.pop(k)
                    raise ValueError(f"Found useless Loss! {k}")
            return losses
        if self.semantic_on and not self.sem_seg_postprocess_before_inference and not self.is_avss_data:
            # fused tail (csrc/infer.hip): upsample + sigmoid + class-weighted sum without the [BT,Q,H,W] intermediate
            from .ops.infer import semantic_inference
            sem = semantic_inference(outputs["pred_logits"], outputs["pred_masks"], tuple(images.shape[-2:]))
            res = []
            for num_img in range(sem.shape[0]):
                inp = batched_inputs[num_img // 5]
                height, width = inp.get("height", image_size[0]), inp.get("width", image_size[1])
                res.append({"sem_seg": sem_seg_postprocess(sem[num_img], image_size, height, width)})
            return res
        mask_cls_results = outputs["pred_logits"]
        mask_pred_results = F.interpolate(outputs["pred_masks"].float(), size=tuple(images.shape[-2:]), mode="bilinear",
                                          align_corners=False)
        del outputs
        num_frames = int(vid_flag.sum()) if self.is_avss_data else 5
        processed_results, num_video = [], -1
        for num_img, (mask_cls_result, mask_pred_result) in enumerate(zip(mask_cls_results, mask_pred_results)):
            if num_img % num_frames == 0:
                num_video += 1
                input_per_image = batched_inputs[num_video]
            height = input_per_image.get("height", image_size[0])
            width = input_per_image.get("width", image_size[1])
            processed_results.append({})
            if self.sem_seg_postprocess_before_inference:
                mask_pred_result = sem_seg_postprocess(mask_pred_result, image_size, height, width)
            if self.semantic_on:
                if self.is_avss_data:
                    r = self.semantic_inference_ss(mask_cls_result, mask_pred_result, vid_flag[num_img])
                else:
                    r = self.semantic_inference(mask_cls_result, mask_pred_result)
                if not self.sem_seg_postprocess_before_inference:
                    r = sem_seg_postprocess(r, image_size, height, width)
                processed_results[-1]["sem_seg"] = r
        return processed_results

    def prepare_targets(self, targets, images):
        h_pad, w_pad = images.shape[-2:]
        new_targets = []
        for t in targets:
            gt_masks = _field(t, "gt_masks").to(self.device)
            if hasattr(gt_masks, "tensor"):
                gt_masks = gt_masks.tensor
            if tuple(gt_masks.shape[1:]) == (h_pad, w_pad):
                padded = gt_masks  # nothing to pad (every shipped recipe: fixed-size inputs): no zero-fill + copy per clip
            else:
                padded = torch.zeros((gt_masks.shape[0], h_pad, w_pad), dtype=gt_masks.dtype, device=gt_masks.device)
                padded[:, : gt_masks.shape[1], : gt_masks.shape[2]] = gt_masks
            new_targets.append({"labels": _field(t, "gt_classes").to(self.device), "masks": padded})
        return new_targets

    def semantic_inference(self, mask_cls, mask_pred):
        mask_cls = F.softmax(mask_cls.float(), dim=-1)[..., :-1]
        return torch.einsum("qc,qhw->chw", mask_cls, mask_pred.sigmoid())

    def semantic_inference_ss(self, mask_cls, mask_pred, vid_temporal_mask_flag):
        return self.semantic_inference(mask_cls, mask_pred) * vid_temporal_mask_flag


def sem_seg_postprocess(result, img_size, output_height, output_width):
    """detectron2 sem_seg_postprocess: crop the padding away, resize to the requested output resolution."""
    result = result[:, : img_size[0], : img_size[1]].expand(1, -1, -1, -1)
    if (output_height, output_width) == tuple(img_size):
        return result[0]
    return F.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]


def build_model(cfg):
    cls = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)
    model = cls(**cls.from_config(cfg))
    return model
