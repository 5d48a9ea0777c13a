"""AMG state caching (reference ``micro_sam/precompute_state.py:27-87``).

``cache_amg_state`` computes the automatic-mask-generator state for an image (or the slice ``i`` of a volume) or loads
it from ``save_path/amg_state.pickle`` (``save_path/amg_state/state-{i}.pkl``).  The pickle has the reference's
format: ``{"crop_list": [MaskData with host tensors and RLE dicts], "crop_boxes", "original_size"}`` - the device bit
masks are serialised as the reference's ``rles`` column (``DeviceMaskData.__getstate__``), so a state written here can
be read without a GPU and vice versa.  The in-memory state of the returned generator stays in HBM.
"""
import os
import pickle
from typing import Optional, Union

import numpy as np

from . import instance_segmentation, util
from .predictor import SamPredictor


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
        with open(save_path_amg, "rb") as f:
            amg_state = pickle.load(f)
        amg.set_state(amg_state)
        return amg
    if verbose:
        print("Precomputing the state for instance segmentation.")
    amg.initialize(raw if i is None else raw[i], image_embeddings=image_embeddings, verbose=verbose, i=i)
    os.makedirs(save_path, exist_ok=True)
    with open(save_path_amg, "wb") as f:
        pickle.dump(amg.get_state(), f)          # device columns are converted to host tensors / RLEs while pickling
    return amg
