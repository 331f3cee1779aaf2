"""Fused inference tail (csrc/infer.hip): semseg[k] = sum_q softmax(cls[q])[k] * sigmoid(bilinear_up(mask[q]))."""
import torch

from .. import _lib


@torch.no_grad()
def semantic_inference(pred_logits, pred_masks, out_size):
    """pred_logits [F,Q,K+1], pred_masks [F,Q,h,w] -> [F,K,H,W] (maskformer_model.py:397-402, 460-464)"""
    prob = torch.softmax(pred_logits.float(), dim=-1)[..., :-1].contiguous()
    masks = pred_masks.float().contiguous()
    _lib.require_cuda(prob, masks)
    F_, Q, K = prob.shape
    h, w = masks.shape[-2:]
    H, W = out_size
    outs = []
    for k0 in range(0, K, 8):
        pk = prob[..., k0:k0 + 8].contiguous()
        o = torch.empty(F_, pk.shape[-1], H, W, device=masks.device, dtype=torch.float32)
        _lib.check(_lib.lib().combo_semantic_inference_f32(pk.data_ptr(), masks.data_ptr(), F_, Q, pk.shape[-1], h, w, H, W,
                                                           o.data_ptr(), _lib.current_stream()), "combo_semantic_inference_f32")
        outs.append(o)
    return outs[0] if len(outs) == 1 else torch.cat(outs, 1)
