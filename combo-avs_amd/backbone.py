"""Host-PyTorch backbones (outside the HIP hot path, BASELINE.json north_star): a detectron2-compatible
ResNet-50 (BasicStem + Bottleneck, FrozenBN, STRIDE_IN_1X1=False: configs/avs_s4/R50-AVSS4-SemanticSegmentation.yaml:2-23
of the reference) and the VGGish audio encoder (audio_backbone/torchvggish/vggish.py:9-27,89-100).
Parameter / buffer names follow detectron2 (`stem.conv1.weight`, `res2.0.conv1.norm.running_mean`, ...) and
torchvggish (`features.N`, `embeddings.N`) so reference checkpoints load 1:1.

MI355X notes: FrozenBN is an affine map with constant statistics, so it is folded into the preceding conv's
weights each step (w' = w * scale, b' = shift) and the conv runs with bias; activations are channels_last."""
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


class ConvBN(nn.Conv2d):
    """conv (no bias) + FrozenBN, folded: conv(x, w*scale) + shift."""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__(cin, cout, k, stride=stride, padding=padding, bias=False)
        self.norm = FrozenBatchNorm2d(cout)
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
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

    def forward(self, x):
        out = F.relu_(self.conv1(x))
        out = F.relu_(self.conv2(out))
        out = self.conv3(out)
        sc = self.shortcut(x) if self.shortcut is not None else x
        return F.relu_(out + sc)


class BasicStem(nn.Module):
    def __init__(self, cin=3, cout=64):
        super().__init__()
        self.conv1 = ConvBN(cin, cout, 7, stride=2, padding=3)

    def forward(self, x):
        return F.max_pool2d(F.relu_(self.conv1(x)), kernel_size=3, stride=2, padding=1)


class ResNet(nn.Module):
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

    def forward(self, x):
        x = self.stem(x.contiguous(memory_format=torch.channels_last))
        out = {}
        for name in ("res2", "res3", "res4", "res5"):
            x = getattr(self, name)(x)
            if name in self._out_features:
                out[name] = x
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
        x = self.features(x)
        x = x.permute(0, 2, 3, 1).reshape(x.size(0), -1)  # vggish.py:21-25
        return self.embeddings(x)
