import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vit_b_sd():
    from micro_sam_amd.synthetic import synthetic_state_dict
    return synthetic_state_dict("vit_b", 0)


@pytest.fixture(scope="session", autouse=True)
def _oracle_decoder_dtype():
    """The oracle's "bf16" (= HIP-like) mode rounds the decoder sites to the decoder type of the library that is loaded."""
    import torch
    if torch.cuda.is_available():
        from micro_sam_amd import _lib
        from oracle import sam_ref as S
        S.DECODER_DTYPE = _lib.decoder_dtype()
    yield
