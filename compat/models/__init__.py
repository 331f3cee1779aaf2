"""`models` alias package: put `<repo>/compat` in front of the reference checkout on PYTHONPATH and the reference's unchanged
drivers (`from models import (BestCheckpointer, SemSegEvaluator, SemSegEvaluator_SS, AVSS4_/AVSMS3_/AVSS_SemanticDatasetMapper,
add_maskformer2_config, add_audio_config, add_fuse_config)`, train_net.py:52-62; pred.py:50-62 adds inference_on_dataset(_ss))
run with the MI355X implementation of the hot path:

  * the config functions and `MaskFormer` come from combo_avs_amd, whose components are installed into detectron2's
    registries under the reference's names (combo_avs_amd.d2_register.install(override=True));
  * everything OUTSIDE the hot path (dataset mappers, evaluators, the BestCheckpointer hook: data / evaluation / engine
    sub-packages) is the reference's own code, found through COMBO_REFERENCE_ROOT=<reference checkout> whose `models/`
    directory is appended to this package's search path - those names are resolved lazily, on first use."""
import importlib
import os
import sys

_repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _repo not in sys.path:
    sys.path.insert(0, _repo)

import combo_avs_amd  # noqa: E402
from combo_avs_amd import add_audio_config, add_fuse_config, add_maskformer2_config  # noqa: E402,F401
from combo_avs_amd.meta_arch import MaskFormer  # noqa: E402,F401

try:
    INSTALLED = combo_avs_amd.d2_register.install(override=True)
except ImportError:  # no detectron2: nothing to register into (the package's own registries are always populated)
    INSTALLED = []

_ref = os.environ.get("COMBO_REFERENCE_ROOT")
if _ref and os.path.isdir(os.path.join(_ref, "models")):
    __path__.append(os.path.join(_ref, "models"))  # data / evaluation / engine sub-packages resolve to the reference's files

_LAZY = {
    "BestCheckpointer": ("engine.hooks", "BestCheckpointer"),
    "SemSegEvaluator": ("evaluation.sem_seg_evaluation", "SemSegEvaluator"),
    "SemSegEvaluator_SS": ("evaluation.sem_seg_evaluation_ss", "SemSegEvaluator_SS"),
    "inference_on_dataset": ("evaluation.evaluator", "inference_on_dataset"),
    "inference_on_dataset_ss": ("evaluation.evaluator", "inference_on_dataset_ss"),
    "AVSS4_SemanticDatasetMapper": ("data.dataset_mappers.avss4_semantic_dataset_mapper", "AVSS4_SemanticDatasetMapper"),
    "AVSMS3_SemanticDatasetMapper": ("data.dataset_mappers.avsms3_semantic_dataset_mapper", "AVSMS3_SemanticDatasetMapper"),
    "AVSS_SemanticDatasetMapper": ("data.dataset_mappers.avss_semantic_dataset_mapper", "AVSS_SemanticDatasetMapper"),
}


def __getattr__(name):
    if name in _LAZY:
        mod, attr = _LAZY[name]
        try:
            return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
        except ImportError as e:
            raise ImportError(f"models.{name} is outside the hot path and lives in the reference checkout: set "
                              f"COMBO_REFERENCE_ROOT (and install its dependencies) - {e}") from e
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
