"""Drop-in registration for the reference's drivers (train_net.py:52-62, 250-278; pred.py:50-62): puts this package's
components under the names the reference registers - META_ARCH_REGISTRY["MaskFormer"] (maskformer_model.py:28),
SEM_SEG_HEADS_REGISTRY["MaskFormerHead" | "MSDeformAttnPixelDecoder"] (mask_former_head.py:18, msdeformattn.py:168),
BACKBONE_REGISTRY["build_pvtv2_b5_backbone"] (pvtv2.py:391) - so that detectron2's `build_model(cfg)` resolves to the
MI355X implementation.  detectron2 calls a registry entry as `entry(cfg[, input_shape])`; the entries installed here are
factories doing `cls(**cls.from_config(cfg, ...))` (what d2's @configurable does for the reference's classes).

Registration is EXPLICIT: importing combo_avs_amd never writes into detectron2's global registries (a side-by-side run
with the reference's own `models` package would otherwise depend on import order - fvcore's Registry asserts on a second
registration of a name).  `install()` leaves names already taken alone; `install(override=True)` replaces entries the
reference's own `models` package registered - that is what the alias package compat/models does, so `from models import ...`
in the unchanged drivers picks this implementation for the hot path.
The registries are passed in by tests as detectron2-shaped stubs (detectron2 is not installed in the build image)."""


def _factory(cls, name):
    def build(cfg, *args, **kwargs):
        return cls(**cls.from_config(cfg, *args, **kwargs))
    build.__name__ = build.__qualname__ = name
    build.combo_class = cls
    build.__doc__ = f"combo_avs_amd factory for {cls.__module__}.{cls.__name__} (from_config)"
    return build


def entries():
    """{registry name: {entry name: object}} of everything this package offers under the reference's names."""
    from .backbone_pvt import build_pvtv2_b5_backbone
    from .meta_arch import MaskFormer
    from .modeling.head import MaskFormerHead
    from .modeling.pixel_decoder import MSDeformAttnPixelDecoder
    return {
        "META_ARCH_REGISTRY": {"MaskFormer": _factory(MaskFormer, "MaskFormer")},
        "SEM_SEG_HEADS_REGISTRY": {"MaskFormerHead": _factory(MaskFormerHead, "MaskFormerHead"),
                                   "MSDeformAttnPixelDecoder": _factory(MSDeformAttnPixelDecoder, "MSDeformAttnPixelDecoder")},
        "BACKBONE_REGISTRY": {"build_pvtv2_b5_backbone": build_pvtv2_b5_backbone},
    }


def _d2_registries():
    from detectron2.modeling import BACKBONE_REGISTRY, META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY  # noqa: PLC0415
    return {"META_ARCH_REGISTRY": META_ARCH_REGISTRY, "SEM_SEG_HEADS_REGISTRY": SEM_SEG_HEADS_REGISTRY,
            "BACKBONE_REGISTRY": BACKBONE_REGISTRY}


def install(registries=None, override=False):
    """-> list of "REGISTRY[name]" strings that now point at this package.  registries: {name: registry}; a registry needs
    `__contains__`, `register(obj)` (name = obj.__name__) and, for override, the `_obj_map` dict of fvcore's Registry."""
    if registries is None:
        registries = _d2_registries()
    done = []
    for reg_name, objs in entries().items():
        reg = registries.get(reg_name)
        if reg is None:
            continue
        for name, obj in objs.items():
            if name in reg:
                if not override:
                    continue
                reg._obj_map[name] = obj
            else:
                reg.register(obj)
            done.append(f"{reg_name}[{name}]")
    return done
