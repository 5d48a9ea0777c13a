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


# ------------------------------------------------------------------------------------------------ is_state (decoder-based segmenters)

class _NpzGroup(dict):
    """One group of the h5py-less state container: dataset name -> array."""

    def create_dataset(self, name, data=None, compression=None):
        self[name] = np.asarray(data)
        return self[name]


class _NpzStateFile:
    """The subset of ``h5py.File(path, "a")`` that ``cache_is_state`` uses (``key in f``, ``f[key][name][:]``, ``create_group`` +
    ``create_dataset(name, data=, compression="gzip")``) over one compressed ``.npz`` file with keys ``<group>/<dataset>`` - the
    container of installations WITHOUT h5py.  It is this package's own format (``is_state.npz`` next to where ``is_state.h5`` would
    be): the reference's tools read ``is_state.h5`` only, which is written whenever h5py is importable."""

    def __init__(self, path):
        self._path = path
        self._groups: Dict[str, _NpzGroup] = {}
        if os.path.exists(path):
            with np.load(path) as z:
                for key in z.files:
                    g, name = key.split("/", 1)
                    self._groups.setdefault(g, _NpzGroup())[name] = z[key]
        self._dirty = False

    def __contains__(self, key):
        return key in self._groups

    def __getitem__(self, key):
        return self._groups[key]

    def create_group(self, key):
        if key in self._groups:
            raise ValueError(f"Unable to create group (name already exists): {key}")
        self._dirty = True
        self._groups[key] = _NpzGroup()
        return self._groups[key]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self._dirty and exc[0] is None:
            tmp = self._path + ".tmp.npz"
            np.savez_compressed(tmp, **{f"{g}/{name}": arr for g, grp in self._groups.items() for name, arr in grp.items()})
            os.replace(tmp, self._path)
        return False


def _open_is_state(save_path):
    """(context manager, file path): ``is_state.h5`` through h5py when it is importable (the reference's container, gzip datasets,
    precompute_state.py:124-153), else ``is_state.npz`` (see ``_NpzStateFile``)."""
    try:
        import h5py
    except ImportError:
        path = os.path.join(save_path, "is_state.npz")
        return _NpzStateFile(path), path
    path = os.path.join(save_path, "is_state.h5")
    return h5py.File(path, "a"), path


def cache_is_state(predictor: SamPredictor, decoder, raw: np.ndarray, image_embeddings: util.ImageEmbeddings,
                   save_path: Union[str, os.PathLike], verbose: bool = True, i: Optional[int] = None, skip_load: bool = False,
                   **kwargs):
    """Reference ``cache_is_state`` (precompute_state.py:90-155): compute - or load - the state of the decoder-based segmenter
    (foreground, centre distances, boundary distances) under ``save_path/is_state.h5``, group ``state`` or ``state-{i}``."""
    is_tiled = image_embeddings["input_size"] is None
    amg = instance_segmentation.get_instance_segmentation_generator(predictor, is_tiled=is_tiled, decoder=decoder, **kwargs)
    os.makedirs(save_path, exist_ok=True)
    save_key = "state" if i is None else f"state-{i}"
    store, path = _open_is_state(str(save_path))
    with store as f:
        if save_key in f:
            if skip_load:
                return None
            if verbose:
                print("Load instance segmentation state from", path, ":", save_key)
            g = f[save_key]
            amg.set_state({"foreground": g["foreground"][:], "boundary_distances": g["boundary_distances"][:],
                           "center_distances": g["center_distances"][:]})
            return amg
    if verbose:
        print("Precomputing the state for instance segmentation.")
    amg.initialize(raw, image_embeddings=image_embeddings, verbose=verbose, i=i)
    state = amg.get_state()
    store, _ = _open_is_state(str(save_path))
    with store as f:
        g = f.create_group(save_key)
        g.create_dataset("foreground", data=state["foreground"], compression="gzip")
        g.create_dataset("boundary_distances", data=state["boundary_distances"], compression="gzip")
        g.create_dataset("center_distances", data=state["center_distances"], compression="gzip")
    return amg


def _precompute_state_for_file(predictor, input_path, output_path, key, ndim, tile_shape, halo, precompute_amg_state, decoder, verbose):
    """Reference ``_precompute_state_for_file`` (precompute_state.py:158-192): embeddings of one image / volume into ``<output>.zarr``,
    then - optionally - the AMG (or, with a decoder, the AIS) state next to them, per slice for volumes."""
    from functools import partial
    from pathlib import Path
    image_data = input_path if isinstance(input_path, np.ndarray) else util.load_image_data(input_path, key)
    output_path = str(Path(output_path).with_suffix(".zarr"))
    embeddings = util.precompute_image_embeddings(predictor, image_data, output_path, ndim=ndim, tile_shape=tile_shape, halo=halo,
                                                  verbose=verbose)
    if not precompute_amg_state:
        return
    if decoder is None:
        cache_function = partial(cache_amg_state, predictor=predictor, image_embeddings=embeddings, save_path=output_path)
    else:
        cache_function = partial(cache_is_state, predictor=predictor, decoder=decoder, image_embeddings=embeddings, save_path=output_path)
    if ndim is None:
        ndim = image_data.ndim
    if ndim == 2:
        cache_function(raw=image_data, verbose=verbose)
    else:
        _, pbar_init, pbar_update, pbar_close = util.handle_pbar(verbose, None, None)
        pbar_init(image_data.shape[0], "Precompute instance segmentation state")
        for i in range(image_data.shape[0]):
            cache_function(raw=image_data, i=i, verbose=False)
            pbar_update(1)
        pbar_close()


def _precompute_state_for_files(predictor, input_files, output_path, key=None, ndim=None, tile_shape=None, halo=None,
                                precompute_amg_state: bool = False, decoder=None):
    """Reference ``_precompute_state_for_files`` (:195-224): one ``.zarr`` per input, named after the file (arrays: ``embedding_%05d``)."""
    os.makedirs(output_path, exist_ok=True)
    for idx, file_path in enumerate(input_files):
        name = f"embedding_{idx:05}.tif" if isinstance(file_path, np.ndarray) else os.path.basename(file_path)
        _precompute_state_for_file(predictor, file_path, os.path.join(output_path, name), key=key, ndim=ndim, tile_shape=tile_shape,
                                   halo=halo, precompute_amg_state=precompute_amg_state, decoder=decoder, verbose=False)


def precompute_state(input_path, output_path, pattern: Optional[str] = None, model_type: str = util._DEFAULT_MODEL,
                     checkpoint_path=None, key: Optional[str] = None, ndim: Optional[int] = None, tile_shape=None, halo=None,
                     precompute_amg_state: bool = False, state_dict=None, device=None) -> None:
    """Reference ``precompute_state`` (precompute_state.py:227-278): embeddings - and optionally the automatic-segmentation state - for
    one image file / container dataset / in-memory array, or (``pattern``) for every matching file of a folder.  ``state_dict`` /
    ``device`` (extensions): a model without a checkpoint file (there is no download here) and its device.  A checkpoint that carries a
    ``decoder_state`` (the reference's ``*_lm`` / ``*_em`` models) gives the AIS state through ``models.unetr.get_decoder``."""
    from glob import glob
    predictor, state = util.get_sam_model(model_type=model_type, device=device, checkpoint_path=checkpoint_path, return_state=True,
                                          state_dict=state_dict)
    decoder = None
    if isinstance(state, dict) and "decoder_state" in state:
        from .models.unetr import get_decoder
        decoder = get_decoder(predictor.model.image_encoder, state["decoder_state"], device=predictor.device)
    if pattern is None:
        _precompute_state_for_file(predictor, input_path, output_path, key, ndim=ndim, tile_shape=tile_shape, halo=halo,
                                   precompute_amg_state=precompute_amg_state, decoder=decoder, verbose=True)
    else:
        input_files = sorted(glob(os.path.join(input_path, pattern)))
        _precompute_state_for_files(predictor, input_files, output_path, key=key, ndim=ndim, tile_shape=tile_shape, halo=halo,
                                    precompute_amg_state=precompute_amg_state, decoder=decoder)
