"""Host-PyTorch backbones (outside the HIP hot path, BASELINE.json north_star): a detectron2-compatible
ResNet-50 (BasicStem + Bottleneck, FrozenBN, STRIDE_IN_1X1=False: configs/avs_s4/R50-AVSS4-SemanticSegmentation.yaml:2-23
of the reference) and the VGGish audio encoder (audio_backbone/torchvggish/vggish.py:9-27,89-100).
Parameter / buffer names follow detectron2 (`stem.conv1.weight`, `res2.0.conv1.norm.running_mean`, ...) and
torchvggish (`features.N`, `embeddings.N`) so reference checkpoints load 1:1.

MI355X notes: FrozenBN is an affine map with constant statistics, so it is folded into the preceding conv's
weights each step (w' = w * scale, b' = shift) and the conv runs with bias; activations are channels_last.
The fold of ALL convolutions of a ResNet (+ the cast to the compute dtype, + the way back for the gradients) is ONE
autograd node built on multi-tensor (`torch._foreach_*`) kernels: ~4 launches per backbone and direction instead of
~10 tiny launches per convolution (53 convolutions x 2 backbones); scale/shift are cached (the statistics are frozen)."""
import torch
from torch import nn
from torch.nn import functional as F

from .registry import BACKBONE_REGISTRY, ShapeSpec


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def scale_shift(self):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        return scale, self.bias - self.running_mean * scale

    def forward(self, x):
        scale, shift = self.scale_shift()
        return x * scale.to(x.dtype)[None, :, None, None] + shift.to(x.dtype)[None, :, None, None]


CHAIN_RELU = True  # block-to-block ReLU gradient + fan-out sum inside the next block's first dX GEMM (A/B: tools/ab_const.py)
GRAD_IN_PLACE = True  # gradients of the folded convolution weights are written into the optimiser's flat buffer (A/B: tools/ab_const.py)


class _FoldAll(torch.autograd.Function):
    """folded_i = (weight_i * scale_i).to(dtype) for every convolution of a backbone, as one node."""

    @staticmethod
    def forward(ctx, dtype, scales, *weights):
        ctx.scales = scales
        ctx.wptrs = [w.data_ptr() for w in weights]
        if weights[0].is_cuda and all(w.is_contiguous() for w in weights):
            from .ops.foldcast import fold_cast  # ONE launch per 56 tensors (csrc/foldcast.hip)
            out = [torch.empty_like(w, dtype=dtype) for w in weights]
            fold_cast([w.detach() for w in weights], out, scales)
            if GRAD_IN_PLACE and dtype == torch.float32 and any(ctx.needs_input_grad[2:]):
                # the gradient of a folded weight may be written where its parameter's gradient lives (the optimiser's flat
                # buffer): this node scales it in place
                from .ops.linear import register_grad_aliases
                register_grad_aliases(out, weights)
            return tuple(out)
        folded = torch._foreach_mul(weights, scales)
        if dtype != torch.float32:
            out = [torch.empty_like(w, dtype=dtype) for w in weights]
            torch._foreach_copy_(out, folded)
        else:
            out = folded
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        out = [None] * len(grads)
        idx = [i for i, g in enumerate(grads) if g is not None]
        if idx and grads[idx[0]].is_cuda and grads[idx[0]].dtype == torch.float32:
            # weight gradients that are problems of the deferred grouped launch (ops/convwrw.py) do not exist yet: they pass
            # through as they are and take their FrozenBN scale in place once the launch has written them
            from .ops import linear as L
            from .ops.foldcast import fold_cast
            late = [i for i in idx if L.is_deferred_dest(grads[i])]
            if late:
                tensors, scales = [grads[i] for i in late], [ctx.scales[i] for i in late]
                L.after_flush(lambda: fold_cast(tensors, tensors, scales))
                for i in late:
                    out[i] = grads[i]
                idx = [i for i in idx if i not in set(late)]
        g32 = [None] * len(idx)
        if idx and grads[idx[0]].is_cuda:
            from .ops import linear as L
            from .ops.foldcast import fold_cast
            for j, i in enumerate(idx):  # straight into the optimiser's flat gradient buffer where the parameter is registered
                g32[j] = L.grad_target(ctx.wptrs[i], grads[i].shape, torch.float32) if GRAD_IN_PLACE else None
        g32 = [g if g is not None else torch.empty(grads[i].shape, dtype=torch.float32, device=grads[i].device) for g, i in zip(g32, idx)]
        if idx and grads[idx[0]].is_cuda:
            fold_cast([grads[i].contiguous() for i in idx], g32, [ctx.scales[i] for i in idx])
        elif idx:
            torch._foreach_copy_(g32, [grads[i] for i in idx])
            torch._foreach_mul_(g32, [ctx.scales[i] for i in idx])
        for i, g in zip(idx, g32):
            out[i] = g
        if GRAD_IN_PLACE:
            # every consumer of the folded copies has run its weight-gradient kernel by now (their gradients arrived here): the
            # aliases "folded copy -> parameter" end with this backward pass, so a recycled address of a dead folded copy can never
            # resolve to another parameter's flat-buffer view in a later step (eval forward, frozen backbone, switch flipped)
            from .ops.linear import drop_grad_aliases
            drop_grad_aliases(ctx.wptrs)
        return (None, None) + tuple(out)


class ConvBN(nn.Conv2d):
    """conv (no bias) + FrozenBN, folded: conv(x, w*scale) + shift."""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__(cin, cout, k, stride=stride, padding=padding, bias=False)
        self.norm = FrozenBatchNorm2d(cout)
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x, folded=None, bias=True, mask_dx=False):
        if folded is not None:  # (w * scale, shift) prepared for the whole backbone by ResNet.forward
            if not bias:
                from .ops.convwrw import conv2d  # fp32 recipe: weight gradient on the head's kernels where they apply
                return conv2d(x, folded[0], self.stride, self.padding, mask_dx)
            return F.conv2d(x, folded[0], folded[1] if bias else None, self.stride, self.padding)
        scale, shift = self.norm.scale_shift()
        w = self.weight * scale[:, None, None, None]
        return F.conv2d(x, w, shift, self.stride, self.padding)  # autocast (if enabled) picks the compute dtype


class Bottleneck(nn.Module):
    def __init__(self, cin, cout, mid, stride):
        super().__init__()
        self.shortcut = ConvBN(cin, cout, 1, stride=stride) if cin != cout else None
        self.conv1 = ConvBN(cin, mid, 1)
        self.conv2 = ConvBN(mid, mid, 3, stride=stride, padding=1)  # STRIDE_IN_1X1: False
        self.conv3 = ConvBN(mid, cout, 1)

    def convs(self):
        """the order `forward` consumes the folded weights in"""
        return [self.conv1, self.conv2, self.conv3] + ([self.shortcut] if self.shortcut is not None else [])

    def forward(self, x, folded=None, x_res=None, fanout=False, chain_in=False):
        """x_res: the same values as x as a second autograd output of the producing block (its identity / shortcut consumer);
        fanout: hand the output over in that form (-> a pair); fanout == "chain": hand over ONE output whose gradient comes
        back complete and ReLU-masked (the next block is called with chain_in=True: its first convolution's input-gradient GEMM
        sums the identity branch's gradient and applies this block's ReLU mask, ops.convwrw.conv_bias_act(passthrough=True))."""
        if folded is None:
            out = F.relu_(self.conv1(x))
            out = F.relu_(self.conv2(out))
            out = self.conv3(out)
            sc = self.shortcut(x) if self.shortcut is not None else x
            return F.relu_(out + sc)
        # folded path, fp32: bias (+ residual) + ReLU ride in the convolution's GEMM epilogue where the layer runs on the head's
        # kernels (ops/convwrw.py conv_bias_act), else one fused pass after the library's convolution (csrc/biasact.hip)
        xr = x if x_res is None else x_res
        f1, f2, f3 = next(folded), next(folded), next(folded)
        if x.dtype == torch.float32:
            from .ops import convwrw
            cba = convwrw.conv_bias_act
            c1, c2, c3 = self.conv1, self.conv2, self.conv3
            # a ReLU whose output feeds ONE convolution: its gradient mask rides in the epilogue of that convolution's
            # input-gradient GEMM instead of a pass of its own (conv1 -> conv2 when conv2's dX is an own kernel, conv2 -> conv3)
            # (the kind conv_bias_act will pick for conv2 / conv3, from the activation shapes they will see: conv1 keeps the map,
            # conv2 strides it - a res5 3x3 on a 1 x 1 map, say, is NOT taken by the own kernels)
            ok = x.is_cuda and torch.is_grad_enabled() and not torch.is_autocast_enabled() and x.is_contiguous(memory_format=torch.channels_last)
            B_, _, H_, W_ = x.shape
            s2 = c2.stride[0] if isinstance(c2.stride, (tuple, list)) else c2.stride
            k2 = convwrw.weight_kind(f2[0], c2.stride, c2.padding, (B_, f2[0].shape[1], H_, W_)) if (ok and f2[0].requires_grad) else 0
            k3 = convwrw.weight_kind(f3[0], c3.stride, c3.padding, (B_, f3[0].shape[1], (H_ - 1) // s2 + 1, (W_ - 1) // s2 + 1)) if (ok and f3[0].requires_grad) else 0
            fold1 = convwrw.ENABLED and convwrw.MASK_3X3 and k2 == 3 and bool(convwrw.DX_OWN & 1) and f2[3] is not None
            fold2 = convwrw.ENABLED and convwrw.MASK_1X1 and k3 == 1
            chain_out = fanout == "chain"
            fo = False if chain_out else fanout
            if chain_in:
                assert self.shortcut is None and x_res is None
                out, xr = cba(x, f1[0], f1[2], c1.stride, c1.padding, f1[3], grad_masked=fold1, passthrough=True)
            else:
                out = cba(x, f1[0], f1[2], c1.stride, c1.padding, f1[3], grad_masked=fold1)
            out = cba(out, f2[0], f2[2], c2.stride, c2.padding, f2[3], grad_masked=fold2, mask_dx=fold1)
            if self.shortcut is not None:
                fs = next(folded)
                sc = cba(xr, fs[0], None, self.shortcut.stride, self.shortcut.padding, fs[3])
                return cba(out, f3[0], self._merged_shift(f3[2], fs[2]), c3.stride, c3.padding, f3[3], residual=sc, fanout=fo,
                           grad_masked=chain_out, mask_dx=fold2)
            return cba(out, f3[0], f3[2], c3.stride, c3.padding, f3[3], residual=xr, fanout=fo, grad_masked=chain_out, mask_dx=fold2)
        from .ops.biasact import bias_act
        out = bias_act(self.conv1(x, f1, bias=False), f1[2])
        out = bias_act(self.conv2(out, f2, bias=False), f2[2])
        out = self.conv3(out, f3, bias=False)
        if self.shortcut is not None:
            fs = next(folded)
            return bias_act(out, f3[2] + fs[2], self.shortcut(xr, fs, bias=False), fanout=fanout)
        return bias_act(out, f3[2], xr, fanout=fanout)

    def _merged_shift(self, a, b):
        """FrozenBN shifts of conv3 and of the shortcut as one bias vector (constants: computed once)"""
        key = (a.data_ptr(), b.data_ptr())
        if getattr(self, "_shift_key", None) != key:
            self._shift_key, self._shift_sum = key, (a + b).detach()
        return self._shift_sum


class BasicStem(nn.Module):
    def __init__(self, cin=3, cout=64):
        super().__init__()
        self.conv1 = ConvBN(cin, cout, 7, stride=2, padding=3)

    def forward(self, x, folded=None):
        if folded is None:
            return F.max_pool2d(F.relu_(self.conv1(x)), kernel_size=3, stride=2, padding=1)
        from .ops.biasact import bias_act
        f = next(folded)
        return F.max_pool2d(bias_act(self.conv1(x, f, bias=False), f[2]), kernel_size=3, stride=2, padding=1)


class ResNet(nn.Module):
    # the convolutions of this backbone run on the package's own kernels, none of which waits for another workgroup: two instances
    # may run side by side on two HIP streams (meta_arch.MaskFormer.parallel_backbones)
    concurrent_safe = True

    def __init__(self, depth=50, out_features=("res2", "res3", "res4", "res5")):
        super().__init__()
        blocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[depth]
        self.stem = BasicStem()
        cin, self._out_features = 64, list(out_features)
        self._strides, self._channels = {}, {}
        for i, (n, mid, cout, stride) in enumerate(zip(blocks, (64, 128, 256, 512), (256, 512, 1024, 2048), (1, 2, 2, 2))):
            name = f"res{i + 2}"
            layers = []
            for b in range(n):
                layers.append(Bottleneck(cin, cout, mid, stride if b == 0 else 1))
                cin = cout
            setattr(self, name, nn.Sequential(*layers))
            self._strides[name], self._channels[name] = 4 * 2 ** i, cout
        self.size_divisibility = 0
        self._bn_cache = {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._drop_constants())

    def _drop_constants(self):
        """forget everything derived from the frozen statistics (they were loaded or moved)"""
        self._bn_cache.clear()
        for m in self.modules():
            if isinstance(m, Bottleneck):
                m._shift_key = None

    def _apply(self, fn, *args, **kwargs):
        self._drop_constants()  # .to() / .cuda() moved the frozen statistics
        return super()._apply(fn, *args, **kwargs)

    def _conv_list(self):
        convs = [self.stem.conv1]
        for name in ("res2", "res3", "res4", "res5"):
            for blk in getattr(self, name):
                convs += blk.convs()
        return convs

    def _frozen_affine(self, convs, dtype):
        """(scale [C,1,1,1] fp32, shift [C] compute-dtype) of every FrozenBN: constants, computed once."""
        key = (dtype, str(convs[0].weight.device))
        if key not in self._bn_cache:
            with torch.no_grad():
                ss = [c.norm.scale_shift() for c in convs]
                self._bn_cache[key] = ([sc[:, None, None, None].contiguous() for sc, _ in ss], [sh.to(dtype) for _, sh in ss],
                                       [sh.float().contiguous() for _, sh in ss])
        return self._bn_cache[key]

    def forward(self, x):
        # compute dtype: the caller's autocast dtype, made explicit (weights are folded + cast by one node for the whole
        # backbone, activations are cast once at the input)
        dtype = torch.get_autocast_dtype("cuda") if (x.is_cuda and torch.is_autocast_enabled("cuda")) else x.dtype
        convs = self._conv_list()
        scales, shifts, shifts32 = self._frozen_affine(convs, dtype)
        folded_w = _FoldAll.apply(dtype, scales, *[c.weight for c in convs])
        images = [None] * len(convs)
        if dtype == torch.float32 and x.is_cuda:
            from .ops import convwrw
            training = torch.is_grad_enabled() and folded_w[0].requires_grad
            if convwrw.ENABLED and (convwrw.FWD_X3 or (convwrw.DX_OWN and training)):
                # bf16 hi/lo images of every weight the head's 3-product kernels will read (forward and input gradient)
                with torch.no_grad():
                    images = convwrw.weight_images([w.detach() for w in folded_w], [(c.stride, c.padding) for c in convs])
        folded = iter(zip(folded_w, shifts, shifts32, images))
        with torch.autocast("cuda", enabled=False):
            x = self.stem(x.to(dtype).contiguous(memory_format=torch.channels_last), folded)
            out = {}
            # fp32: a block output goes to the next block twice (first convolution, identity / shortcut branch); handed over as
            # two autograd outputs, the two gradients are added inside the block's ReLU-gradient pass (ops/biasact.py)
            blocks = [(name, blk) for name in ("res2", "res3", "res4", "res5") for blk in getattr(self, name)]
            x_res, chain_in = None, False
            for i, (name, blk) in enumerate(blocks):
                fan = dtype == torch.float32 and i + 1 < len(blocks) and torch.is_grad_enabled()
                is_out = name in self._out_features and (i + 1 == len(blocks) or blocks[i + 1][0] != name)
                # a block inside a stage (next block: no shortcut convolution, same map size): ONE output, "chain" - the sum of its
                # two consumers' gradients and its ReLU mask ride in the next block's first input-gradient GEMM
                chain = (fan and CHAIN_RELU and not is_out and blocks[i + 1][0] == name and blocks[i + 1][1].shortcut is None
                         and x.is_cuda and x.requires_grad)
                # (a stage's last block has a third consumer - the head: a third alias, summed in the same ReLU-gradient pass)
                y = blk(x, folded, x_res, "chain" if chain else ((3 if is_out else 2) if fan else False), chain_in=chain_in)
                if chain:
                    x, x_res = y, None
                else:
                    x, x_res = (y[0], y[1]) if fan else (y, None)
                chain_in = chain
                if is_out:
                    out[name] = y[2] if fan else x
        return out

    def output_shape(self):
        return {n: ShapeSpec(channels=self._channels[n], stride=self._strides[n]) for n in self._out_features}


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, input_shape=None):
    return ResNet(cfg.MODEL.RESNETS.DEPTH, cfg.MODEL.RESNETS.OUT_FEATURES)


class VGGish(nn.Module):
    """[N,1,96,64] log-mel -> [N,128]; PREPROCESS/POSTPROCESS are disabled by every shipped config."""

    def __init__(self, cfg=None, device=None):
        super().__init__()
        layers, cin = [], 1
        for v in (64, "M", 128, "M", 256, 256, "M", 512, 512, "M"):
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.embeddings = nn.Sequential(nn.Linear(512 * 4 * 6, 4096), nn.ReLU(True), nn.Linear(4096, 4096), nn.ReLU(True),
                                        nn.Linear(4096, 128), nn.ReLU(True))
        if cfg is not None and cfg.MODEL.AUDIO.FREEZE_AUDIO_EXTRACTOR:
            import os
            path = cfg.MODEL.AUDIO.PRETRAINED_VGGISH_MODEL_PATH
            if os.path.exists(path):
                self.load_state_dict(torch.load(path, map_location="cpu"))
            if cfg.MODEL.AUDIO.PREPROCESS_AUDIO_TO_LOG_MEL or cfg.MODEL.AUDIO.POSTPROCESS_LOG_MEL_WITH_PCA:
                raise NotImplementedError("wav->log-mel preprocessing / PCA post-processing are offline steps (disabled in all shipped configs)")

    def forward(self, x):
        own = x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and not torch.is_autocast_enabled()
        x = self._features_own(x) if own else self.features(x)
        x = x.permute(0, 2, 3, 1).reshape(x.size(0), -1)  # vggish.py:21-25
        if own:
            # (round 6) the three dense layers on the head's weight-streaming kernel (csrc/gemm_smallm.hip, <= 64 rows) instead of a
            # library GEMM: the extractor may run on a side stream next to the encoders (meta_arch.parallel_audio), and a library
            # solution may be a stream-K kernel whose workgroups wait for each other
            from .ops.linear import linear
            for m in self.embeddings:
                if isinstance(m, nn.Linear):
                    x = linear(x, m.weight, m.bias, relu=True)  # every Linear of vggish.py:13-19 is followed by a ReLU
            return x
        return self.embeddings(x)

    def _features_own(self, x):
        """the frozen extractor's forward (no gradient): the 3x3 convolutions with >= 64 input channels + bias + ReLU as ONE
        launch each of the head's 3-product implicit-GEMM kernel (ops/convwrw.py) on channels_last maps.  Besides the ReLU pass it
        saves, the result is reproducible from run to run - the library's kernel for the last convolution (512 -> 512 on 12 x 8,
        3 840 tokens) accumulates with atomics, and a 1e-7 wobble of the audio token is enough to flip a decoder mask cell."""
        from .ops import convwrw
        mods = list(self.features)
        convs = [m for m in mods if isinstance(m, nn.Conv2d)]
        own = [c.in_channels % 64 == 0 and c.out_channels % 64 == 0 and c.kernel_size == (3, 3) and c.stride == (1, 1)
               and c.padding == (1, 1) and c.bias is not None for c in convs]
        images = convwrw.weight_images([c.weight.detach() for c, o in zip(convs, own) if o], [((1, 1), (1, 1))] * sum(own)) if any(own) else []
        images = iter(images)
        x = x.contiguous(memory_format=torch.channels_last)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv2d) and own[convs.index(m)] and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                x = convwrw._x3_forward(x, 3, next(images)[0], m.bias.detach(), None, True)
                i += 2
                continue
            if isinstance(m, nn.Conv2d) and own[convs.index(m)]:
                next(images)
            x = m(x)
            i += 1
        return x
