"""Short workload for rocprofv3 --pmc passes: one encoder batch of 2 tiles + 3 x (AMG initialize + generate) on cuda:0.
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d gpurun_out/pmc_x -- python tools/pmc_tile.py
tools/pmc_summary.py turns the counter_collection csv files into a per-kernel table."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_sam_amd import util  # noqa: E402
from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator  # noqa: E402
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile  # noqa: E402

dev = torch.device("cuda", 0)
sd = synthetic_state_dict("vit_b", 0, variant="cells")
predictor = util.get_sam_model("vit_b", device=dev, state_dict=sd)
amg = AutomaticMaskGenerator(predictor, device_chunk=1024)
tiles_np = [synthetic_tile(1000 + i) for i in range(2)]
tiles_u8 = torch.stack([torch.as_tensor(util._to_image(t)) for t in tiles_np]).to(dev)
n_enc = int(os.environ.get("PMC_ENC_BATCH", "2"))
feats = predictor.model.image_encoder.forward_u8(tiles_u8[:n_enc]).unsqueeze(1)
emb = {"features": feats, "input_size": (1024, 1024), "original_size": (1024, 1024)}
for i in range(3):
    amg.initialize(tiles_np[i % n_enc], emb, i=i % n_enc)
    lab, flag = amg.generate_device()
torch.cuda.synchronize()
print("instances", int(lab.max().item()))
