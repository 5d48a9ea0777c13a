"""Host-side behaviour of micro_sam_amd.inference that needs no GPU: argument validation with the reference's exceptions
(micro_sam/inference.py:22-72) and the local-Otsu threshold restatement (inference.py:75-134)."""
import numpy as np
import pytest
import torch

from micro_sam_amd import inference


def test_validate_inputs_errors():
    b = np.zeros((3, 4), np.float32); pts = np.zeros((3, 1, 2), np.float32); lab = np.ones((3, 1))
    assert inference._validate_inputs(b, None, None, False, True, None, None) == (3, True, False, False)
    assert inference._validate_inputs(None, pts, lab, True, True, None, None) == (3, False, True, False)
    with pytest.raises(ValueError):
        inference._validate_inputs(None, None, None, False, True, None, None)        # neither boxes nor points
    with pytest.raises(ValueError):
        inference._validate_inputs(None, pts, None, False, True, None, None)         # points without labels
    with pytest.raises(ValueError):
        inference._validate_inputs(None, pts, lab[:2], False, True, None, None)      # count mismatch
    with pytest.raises(ValueError):
        inference._validate_inputs(b[:2], pts, lab, False, True, None, None)         # boxes vs points
    with pytest.raises(ValueError):
        inference._validate_inputs(b, None, None, False, True, [1, 2], None)         # segmentation ids
    with pytest.raises(ValueError):
        inference._validate_inputs(b, None, None, False, True, None, torch.zeros(2, 1, 256, 256))
    with pytest.raises(NotImplementedError):
        inference._validate_inputs(b, None, None, True, False, [1, 2, 3], None)


def test_local_otsu_threshold_two_level_image():
    """A two-level image: every window that sees both levels puts the Otsu threshold at the lower level's bin, the
    spatial maximum is taken and clamped at >= 0."""
    x = torch.full((2, 1, 64, 64), -4.0)
    x[0, 0, 20:44, 20:44] = 6.0
    x[1, 0, :, :] = -3.0                                    # constant image: range clamps to eps, threshold clamps to 0
    t = inference._local_otsu_threshold(x)
    assert tuple(t.shape) == (2, 1, 1)
    assert -4.0 <= t[0].item() <= 6.0 and t[0].item() >= 0.0
    assert t[1].item() == 0.0
    m = x[0, 0] > t[0, 0, 0]
    assert m[30, 30].item() and not m[2, 2].item()
