"""tests/test_gpu_segment.py and tests/test_gpu_postprocess.py on the CPU: the integer / byte stages (box NMS, mask NMS behind apply_nms,
paint + connected components + relabel, the 3-d overlap table and merge, post-processing + RLE bit for bit) - the device tests' own
bodies, run while the product's ops drive the host-compiled library (tests/host_product.py); ``Tensor.cuda()`` is the identity for the
duration of these tests.  Sizes are the device tests' (a few 1024 x 1024 cases are left to the device)."""
import pytest
import torch

import test_gpu_postprocess as PP
import test_gpu_segment as SG
from host_product import product_on_host


@pytest.fixture(scope="module", autouse=True)
def host(tmp_path_factory):
    saved = (torch.Tensor.cuda, SG._gpu, PP._gpu, torch.cuda.is_available)
    with product_on_host(str(tmp_path_factory.mktemp("host_segment"))):
        torch.Tensor.cuda = lambda self, *a, **k: self
        SG._gpu = PP._gpu = lambda: None
        torch.cuda.is_available = lambda: True                 # (the test bodies skip themselves otherwise; nothing here touches a GPU)
        try:
            yield
        finally:
            torch.Tensor.cuda, SG._gpu, PP._gpu, torch.cuda.is_available = saved


@pytest.mark.parametrize("k", [63, 700])
def test_box_nms_matches_oracle(k):
    SG.test_box_nms_matches_oracle(k)


@pytest.mark.parametrize("mode", ["mask", "box", "iomin"])
def test_apply_nms_matches_oracle(mode):
    SG.test_apply_nms_matches_oracle(mode)


def test_label_components_worst_cases():
    SG.test_label_components_worst_cases()


def test_slice_overlaps_and_merge_3d():
    SG.test_slice_overlaps_and_merge_3d()


@pytest.mark.parametrize("in_hw,out_hw", [((1024, 1024), (512, 512))])
def test_postprocess_and_rle_bit_exact(in_hw, out_hw):
    # (exact scales only: torch's CPU bilinear is not bit-stable across hosts for the others - the kernel's arithmetic at those is
    #  checked against a separately rounded restatement in test_postprocess_kernel_host_emulation.py)
    g = torch.Generator().manual_seed(6)
    low = torch.randn(6, 256, 256, generator=g) * 3
    low = torch.nn.functional.avg_pool2d(low[None], 5, stride=1, padding=2)[0] * 4
    low[4] = -5.0
    low[5] = 5.0
    PP.test_postprocess_and_rle_bit_exact(low, in_hw, out_hw)
