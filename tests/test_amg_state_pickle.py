"""amg_state.pickle is the reference's file: a stock install (segment_anything present, micro_sam_amd absent) unpickles it
(VERDICT r1 'pickle contract'; micro_sam/precompute_state.py:75-85)."""
import os
import pickle
import subprocess
import sys
import textwrap

import numpy as np
import torch


def _state():
    from micro_sam_amd import amg_utils
    from oracle import amg_ref as A
    rng = np.random.default_rng(0)
    masks = torch.as_tensor(rng.random((5, 40, 56)) > 0.6)
    md = amg_utils.MaskData(iou_preds=torch.rand(5), points=torch.rand(5, 2), stability_score=torch.rand(5),
                            boxes=A.batched_mask_to_box(masks), rles=A.mask_to_rle(masks))
    md["area"] = torch.arange(5)                      # product-only column: must not leak into the file
    return {"crop_list": [md], "crop_boxes": [[0, 0, 56, 40]], "original_size": (40, 56)}, masks


def test_round_trip_without_segment_anything(tmp_path):
    from micro_sam_amd import precompute_state as PS
    from oracle import amg_ref as A
    state, masks = _state()
    path = tmp_path / "amg_state.pickle"
    PS.save_amg_state(state, path)
    assert "segment_anything" not in sys.modules or hasattr(sys.modules["segment_anything"], "__file__")   # stubs removed
    raw = path.read_bytes()
    assert b"segment_anything.utils.amg" in raw and b"micro_sam_amd" not in raw
    back = PS.load_amg_state(path)
    d = back["crop_list"][0]
    assert set(d._stats) == {"iou_preds", "points", "stability_score", "boxes", "rles"}
    assert torch.equal(d["iou_preds"], state["crop_list"][0]["iou_preds"])
    for r, m in zip(d["rles"], masks):
        assert isinstance(r["counts"], list) and np.array_equal(A.rle_to_mask(r), m.numpy())
    assert back["original_size"] == (40, 56)


def test_stock_install_can_unpickle(tmp_path):
    """A process that has a `segment_anything.utils.amg.MaskData` (restated minimal upstream class) and NO micro_sam_amd on its
    path loads the file with plain pickle.load and finds the reference's columns."""
    from micro_sam_amd import precompute_state as PS
    state, _ = _state()
    path = tmp_path / "amg_state.pickle"
    PS.save_amg_state(state, path)
    pkg = tmp_path / "site" / "segment_anything" / "utils"
    pkg.mkdir(parents=True)
    (tmp_path / "site" / "segment_anything" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "amg.py").write_text(textwrap.dedent('''
        class MaskData:
            def __init__(self, **kwargs):
                self._stats = dict(**kwargs)
            def __getitem__(self, key):
                return self._stats[key]
    '''))
    code = textwrap.dedent(f'''
        import pickle, sys
        assert not any("micro_sam_amd" in m for m in sys.modules)
        with open({str(path)!r}, "rb") as f:
            st = pickle.load(f)
        md = st["crop_list"][0]
        assert type(md).__module__ == "segment_anything.utils.amg" and type(md).__name__ == "MaskData"
        assert sorted(md._stats) == ["boxes", "iou_preds", "points", "rles", "stability_score"], sorted(md._stats)
        assert md["rles"][0]["size"] == [40, 56] and sum(md["rles"][0]["counts"]) == 40 * 56
        print("stock-ok")
    ''')
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = str(tmp_path / "site")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0 and "stock-ok" in r.stdout, r.stderr[-2000:]
