"""AMG state caching (reference ``micro_sam/precompute_state.py:27-87``).

``cache_amg_state`` computes the automatic-mask-generator state for an image (or the slice ``i`` of a volume) or loads
it from ``save_path/amg_state.pickle`` (``save_path/amg_state/state-{i}.pkl``).  The file is the reference's:
``{"crop_list": [segment_anything.utils.amg.MaskData with host tensors and RLE dicts], "crop_boxes", "original_size"}``
(``micro_sam/precompute_state.py:84-85`` pickles ``amg.get_state()``).  ``save_amg_state`` writes the ``crop_list``
entries under the class path ``segment_anything.utils.amg.MaskData`` with the reference's columns (``iou_preds``,
``points``, ``stability_score``, ``boxes``, ``rles`` = uncompressed column-major RLE dicts with list counts), so a stock
micro_sam install - without this package and without a GPU - unpickles them; ``load_amg_state`` reads such a file whether
or not ``segment_anything`` is importable here.  The in-memory state of the returned generator stays in HBM.
"""
import contextlib
import io
import os
import pickle
import sys
import types
from typing import Any, Dict, Optional, Union

import numpy as np
import torch

from . import amg_utils, instance_segmentation, util
from .predictor import SamPredictor

_REF_MODULE, _REF_NAME = "segment_anything.utils.amg", "MaskData"
_REF_COLUMNS = ("iou_preds", "points", "stability_score", "boxes", "rles", "crop_boxes")


def _reference_maskdata_class():
    """``segment_anything.utils.amg.MaskData`` when the package is installed, else a stand-in class object that pickles
    under the same path (it only ever carries ``_stats``, exactly like the upstream class)."""
    try:
        import importlib
        return getattr(importlib.import_module(_REF_MODULE), _REF_NAME), False
    except ImportError:
        cls = type(_REF_NAME, (), {"__module__": _REF_MODULE, "__qualname__": _REF_NAME})
        return cls, True


@contextlib.contextmanager
def _stub_reference_modules(cls):
    """pickle resolves a class by importing its module: make ``segment_anything.utils.amg`` resolvable while dumping when the
    real package is absent (the stubs are removed again: nothing else must mistake them for segment_anything)."""
    names = ["segment_anything", "segment_anything.utils", _REF_MODULE]
    saved = {n: sys.modules.get(n) for n in names}
    try:
        for n in names:
            m = types.ModuleType(n)
            m.__path__ = []
            sys.modules[n] = m
        sys.modules["segment_anything"].utils = sys.modules["segment_anything.utils"]
        sys.modules["segment_anything.utils"].amg = sys.modules[_REF_MODULE]
        setattr(sys.modules[_REF_MODULE], _REF_NAME, cls)
        yield
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m


def _to_reference_stats(data) -> Dict[str, Any]:
    stats = {}
    for k in _REF_COLUMNS:
        if k not in data and not (k == "rles" and "bits" in data):      # "rles" of a device state is produced on first access
            continue
        v = data[k]
        if k == "rles":
            v = [{"size": [int(r["size"][0]), int(r["size"][1])], "counts": np.asarray(r["counts"]).astype(np.int64).tolist()}
                 for r in v]
        elif torch.is_tensor(v):
            v = v.detach().cpu()
        stats[k] = v
    return stats


def save_amg_state(state: Dict[str, Any], path: Union[str, os.PathLike]) -> None:
    """Write an AMG state (``AMGBase.get_state()``) as the reference's ``amg_state.pickle``."""
    cls, stub = _reference_maskdata_class()
    crop_list = []
    for data in state["crop_list"]:
        obj = cls.__new__(cls)
        obj.__dict__["_stats"] = _to_reference_stats(data)
        crop_list.append(obj)
    out = {"crop_list": crop_list, "crop_boxes": state["crop_boxes"], "original_size": state["original_size"]}
    with (_stub_reference_modules(cls) if stub else contextlib.nullcontext()):
        payload = pickle.dumps(out)
    with open(path, "wb") as f:
        f.write(payload)


class _StateUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == _REF_MODULE and name == _REF_NAME:
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                return amg_utils.MaskData             # same dict-of-columns container
        return super().find_class(module, name)


def load_amg_state(path: Union[str, os.PathLike]) -> Dict[str, Any]:
    """Read an ``amg_state.pickle`` written by the reference or by ``save_amg_state``; the crop_list entries come back as
    ``amg_utils.MaskData`` (host tensors + RLE dicts) whatever class they were pickled under."""
    with open(path, "rb") as f:
        state = _StateUnpickler(io.BytesIO(f.read())).load()
    crop_list = []
    for d in state["crop_list"]:
        md = amg_utils.MaskData()
        md._stats = dict(d._stats)
        crop_list.append(md)
    state["crop_list"] = crop_list
    return state


def cache_amg_state(predictor: SamPredictor, raw: np.ndarray, image_embeddings: util.ImageEmbeddings,
                    save_path: Union[str, os.PathLike], verbose: bool = True, i: Optional[int] = None,
                    **kwargs) -> instance_segmentation.AMGBase:
    is_tiled = image_embeddings["input_size"] is None
    amg = instance_segmentation.get_instance_segmentation_generator(predictor, is_tiled=is_tiled, **kwargs)
    if i is None:
        save_path_amg = os.path.join(save_path, "amg_state.pickle")
    else:
        os.makedirs(os.path.join(save_path, "amg_state"), exist_ok=True)
        save_path_amg = os.path.join(save_path, "amg_state", f"state-{i}.pkl")
    if os.path.exists(save_path_amg):
        if verbose:
            print("Load the AMG state from", save_path_amg)
        amg.set_state(load_amg_state(save_path_amg))
        return amg
    if verbose:
        print("Precomputing the state for instance segmentation.")
    amg.initialize(raw if i is None else raw[i], image_embeddings=image_embeddings, verbose=verbose, i=i)
    os.makedirs(save_path, exist_ok=True)
    save_amg_state(amg.get_state(), save_path_amg)   # device columns -> host tensors / RLE dicts, reference class path
    return amg
