import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from micro_sam_amd import util
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
for mt in ("vit_b", "vit_l"):
    p = util.get_sam_model(mt, device="cuda", state_dict=synthetic_state_dict(mt, 0, variant="cells"))
    tiles = torch.stack([torch.as_tensor(util._to_image(synthetic_tile(1000 + i))) for i in range(16)]).cuda()
    enc = p.model.image_encoder
    ref = enc.forward_u8(tiles).clone()
    for b in (1, 2, 4, 5, 6, 8, 12):
        out = torch.cat([enc.forward_u8(tiles[s:s + b]) for s in range(0, 16, b)][: 16 // b])
        n = out.shape[0]
        print(mt, "batch", b, "bit-identical to batch 16:", bool(torch.equal(out, ref[:n])), "max diff", float((out - ref[:n]).abs().max()), flush=True)
