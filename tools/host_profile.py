"""Where the HOST time of the API-level paths goes (cProfile, top functions by cumulative time) - a measuring tool for the GPU box:

    python tools/host_profile.py api        config 2 through the drop-in API (precompute_image_embeddings + initialize + generate)
    python tools/host_profile.py config3    one 2048 x 2048 slice, vit_l, tiled embeddings + TiledAutomaticMaskGenerator
    python tools/host_profile.py train      one SamTrainer.train_iteration (vit_b, decoder only)
"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_sam_amd import util  # noqa: E402
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "api"
dev = torch.device("cuda", 0)


def report(pr, n=28):
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(n)


if what == "api":
    from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator
    p = util.get_sam_model("vit_b", device=dev, state_dict=synthetic_state_dict("vit_b", 0, variant="cells"))
    amg = AutomaticMaskGenerator(p)
    stack = np.stack([synthetic_tile(1000 + i) for i in range(16)])

    def run():
        emb = util.precompute_image_embeddings(p, stack, ndim=3, batch_size=16, verbose=False)
        for z in range(len(stack)):
            amg.initialize(stack[z], emb, i=z)
            amg.generate()
        torch.cuda.synchronize()
    run()
    t0 = time.perf_counter(); run(); print("api path: %.2f ms per tile" % ((time.perf_counter() - t0) / len(stack) * 1e3))
    pr = cProfile.Profile(); pr.enable(); run(); pr.disable(); report(pr)
elif what == "config3":
    from micro_sam_amd import multi_dimensional_segmentation as mds
    from micro_sam_amd.instance_segmentation import TiledAutomaticMaskGenerator
    p = util.get_sam_model("vit_l", device=dev, state_dict=synthetic_state_dict("vit_l", 0, variant="cells"))
    vol = np.stack([synthetic_tile(3000, (2048, 2048))])

    def run():
        mds.segment_slices(vol, p, TiledAutomaticMaskGenerator(p), tile_shape=(768, 768), halo=(128, 128), batch_size=9)
        torch.cuda.synchronize()
    run()
    t0 = time.perf_counter(); run(); print("config3: %.1f ms per slice (9 tiles)" % ((time.perf_counter() - t0) * 1e3))
    pr = cProfile.Profile(); pr.enable(); run(); pr.disable(); report(pr)
else:
    sys.argv = [sys.argv[0]]
    from tools.train_bench import synthetic_batch
    from micro_sam_amd.training import ConvertToSamInputs, SamTrainer, get_trainable_sam_model
    model = get_trainable_sam_model("vit_b", device=dev, state_dict=synthetic_state_dict("vit_b", 0), freeze=["image_encoder", "prompt_encoder"])
    params = [q for q in model.parameters() if q.requires_grad]
    tr = SamTrainer(model, torch.optim.AdamW(params, lr=1e-5), ConvertToSamInputs(transform=model.transform), n_sub_iteration=8,
                    n_objects_per_batch=25, mask_prob=0.5, device=dev)
    rng = np.random.default_rng(0)
    batch = synthetic_batch(rng, 2, (520, 704), 30)

    def run():
        tr.train_iteration(*batch)
        torch.cuda.synchronize()
    run()
    t0 = time.perf_counter(); run(); print("train step: %.0f ms" % ((time.perf_counter() - t0) * 1e3))
    pr = cProfile.Profile(); pr.enable(); run(); pr.disable(); report(pr, 45)
