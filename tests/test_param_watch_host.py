"""modeling._ParamWatch and Sam.lane_view (round 4; ADVICE r3 "cache invalidation by sum(p._version) misses ..."): the key behind the cached
16-bit operand copies moves for every way a parameter can change that torch lets us see, and a lane view shares the parameters but not the
per-call scratch.  CPU only (nothing here touches the library)."""
import torch

from micro_sam_amd import modeling


def _model():
    torch.manual_seed(0)
    return modeling.build_sam("vit_b")


def test_watch_sees_in_place_updates_replaced_parameters_and_data_assignment():
    sam = _model()
    w = modeling._ParamWatch(sam.prompt_encoder, sam.mask_decoder)
    k0 = w.key()
    assert w.key() == k0                                                 # stable while nothing changes
    lin = sam.mask_decoder.iou_prediction_head.layers[0]
    with torch.no_grad():
        lin.weight.mul_(1.5)                                             # optimizer-style in-place update: version counter
    k1 = w.key()
    assert k1 != k0
    lin.weight = torch.nn.Parameter(lin.weight.detach().clone())         # a replaced Parameter object (the module's _parameters dict)
    k2 = w.key()
    assert k2 != k1
    lin.weight.data = lin.weight.data.clone()                            # param.data = ...: same object, same counter, other storage
    k3 = w.key()
    assert k3 != k2
    # NOT visible (documented): a write through .data that keeps the storage - such writers call invalidate()
    lin.weight.data.mul_(0.5)
    assert w.key() == k3
    # a module added later is picked up after reset() (what invalidate() does)
    sam.mask_decoder.extra = torch.nn.Linear(4, 4)
    assert w.key() == k3
    w.reset()
    assert w.key() != k3


def test_encoder_and_decoder_have_their_own_watches_and_invalidate_resets_them():
    sam = _model()
    ke, kd = sam.image_encoder._watch.key(), sam._watch.key()
    with torch.no_grad():
        sam.image_encoder.pos_embed.add_(1.0)
    assert sam.image_encoder._watch.key() != ke and sam._watch.key() == kd
    sam.invalidate()
    assert sam._watch._slots is None and sam.image_encoder._watch._slots is None


def test_lane_view_shares_parameters_and_not_the_scratch():
    sam = _model()
    sam._dec_ws = torch.zeros(4)
    sam._img_state = ("key", None, None, None)
    view = sam.lane_view()
    assert view is not sam and view._dec_ws is None and view._img_state is None
    assert sam._dec_ws is not None and sam._img_state is not None                      # the original keeps its own
    assert view.mask_decoder is sam.mask_decoder and view.image_encoder is sam.image_encoder
    assert all(a is b for a, b in zip(view.parameters(), sam.parameters()))
    view._dec_ws = torch.ones(2)                                                       # writing the view's scratch leaves the model's alone
    assert sam._dec_ws.numel() == 4
    assert view.split_token_mlp is sam.split_token_mlp and view.amg_low_res_dtype == sam.amg_low_res_dtype
