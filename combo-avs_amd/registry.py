"""Name registries mirroring the ones the reference resolves its components through
(META_ARCH_REGISTRY / SEM_SEG_HEADS_REGISTRY / BACKBONE_REGISTRY of detectron2 and the reference's own
TRANSFORMER_DECODER_REGISTRY, transformer_decoder.py:15)."""


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def register(self, obj=None):
        def deco(o):
            name = o.__name__
            if name in self._obj_map:
                raise KeyError(f"'{name}' already registered in {self._name}")
            self._obj_map[name] = o
            return o
        return deco if obj is None else deco(obj)

    def get(self, name):
        if name not in self._obj_map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map


META_ARCH_REGISTRY = Registry("META_ARCH")
SEM_SEG_HEADS_REGISTRY = Registry("SEM_SEG_HEADS")
BACKBONE_REGISTRY = Registry("BACKBONE")
TRANSFORMER_DECODER_REGISTRY = Registry("TRANSFORMER_MODULE")


class ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride
