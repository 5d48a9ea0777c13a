"""The differentiable primitives of the fine-tuning path (training/functional.py: linear on the MFMA GEMM with split-K weight
gradients, LayerNorm, the decoder's attention incl. its row-block kernels) forward AND backward on the CPU: the test bodies are the
device tests of tests/test_gpu_training.py themselves, collected here with a ``dev`` fixture that is the CPU while the product runs
against the host-compiled library (tests/host_product.py) - same shapes, same references (torch autograd), same tolerances."""
import pytest
import torch

import test_gpu_training as G
from host_product import product_on_host


@pytest.fixture(scope="module")
def dev(tmp_path_factory):
    with product_on_host(str(tmp_path_factory.mktemp("host_train"))):
        yield torch.device("cpu")


test_linear_forward_backward = G.test_linear_forward_backward
test_linear_with_a_frozen_weight_and_a_3d_input = G.test_linear_with_a_frozen_weight_and_a_3d_input
test_layer_norm_forward_backward = G.test_layer_norm_forward_backward
test_attention_forward_backward = G.test_attention_forward_backward
