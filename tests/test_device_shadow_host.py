"""util._device_shadow: the device copy of a freshly computed 3-d embedding stack is kept OUTSIDE the returned dict, dies with the
host array and is used by set_precomputed only while the requested slice of the host array is still what was downloaded."""
import gc

import numpy as np
import torch

from micro_sam_amd import util


def test_shadow_is_used_only_for_the_unchanged_host_array():
    dev = torch.randn(3, 1, 256, 64, 64)                     # (a CPU tensor stands in for the device copy)
    host = dev.numpy().copy()
    util._remember_device_shadow(host, dev)
    assert util._device_shadow(host, 1) is not None and torch.equal(util._device_shadow(host, 1), dev[1])
    assert util._device_shadow(host.copy(), 1) is None       # another array with the same values: unknown
    host[2] *= 0.5                                           # the caller masks one slice: that slice falls back to the host data
    assert util._device_shadow(host, 2) is None and util._device_shadow(host, 0) is not None
    # an edit that a strided sample cannot see - half of every channel of slice 1 zeroed, a single value of slice 0 changed
    host[1, 0, :, 32:, :] = 0
    assert util._device_shadow(host, 1) is None
    host[0, 0, 200, 63, 63] += 1.0
    assert util._device_shadow(host, 0) is None
    n = len(util._DEVICE_SHADOWS)
    del host
    gc.collect()
    assert len(util._DEVICE_SHADOWS) == n - 1                # the entry died with the array


def test_embeddings_dict_has_the_reference_keys_only():
    class P:                                                 # set_precomputed on a host-only predictor stand-in
        device = torch.device("cpu")
    host = np.random.default_rng(0).standard_normal((2, 1, 256, 64, 64)).astype(np.float32)
    emb = {"features": host, "input_size": (1024, 1024), "original_size": (1024, 1024)}
    p = util.set_precomputed(P(), emb, i=1)
    assert torch.equal(p.features, torch.from_numpy(host[1])) and p.is_image_set
    assert set(emb) == {"features", "input_size", "original_size"}
