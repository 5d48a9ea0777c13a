"""tests/test_gpu_training_encoders.py on the CPU (host-compiled library behind the product's training primitives): the rel-pos attention
kernels forward + backward against the dense fp64 attention, the library-GEMM route against them, LayerNorm backward at the encoder
widths - the device tests' own bodies with a CPU ``dev`` fixture (tests/host_product.py)."""
import pytest
import torch

import test_gpu_training_encoders as G
from host_product import product_on_host


@pytest.fixture(scope="module")
def dev(tmp_path_factory):
    with product_on_host(str(tmp_path_factory.mktemp("host_train_enc"))):
        yield torch.device("cpu")


@pytest.mark.parametrize("BH,Gh,Gw,D", [(6, 14, 14, 64), (3, 14, 14, 80), (2, 5, 9, 64)])
def test_relpos_attention_forward_backward(dev, BH, Gh, Gw, D):
    G.test_relpos_attention_forward_backward(dev, BH, Gh, Gw, D)


@pytest.mark.parametrize("BH,Gh,Gw,D", [(6, 14, 14, 64), (3, 14, 14, 80)])
def test_relpos_attention_gemm_route_against_the_fp32_kernels(dev, BH, Gh, Gw, D):
    G.test_relpos_attention_gemm_route_against_the_fp32_kernels(dev, BH, Gh, Gw, D)


test_layer_norm_backward_encoder_widths = G.test_layer_norm_backward_encoder_widths
