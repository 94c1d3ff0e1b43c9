"""CPU: the product's DressCode agnostic mask (idm_vton_amd/dresscode.py, SURVEY.md 8 f2 "without cv2") against the REFERENCE'S OWN
get_agnostic (/root/reference/inference_dc.py:231-352) -- through the committed fixture tests/golden/dresscode_agnostic.npz everywhere,
and re-executed live where the reference checkout exists (oracle/make_golden_agnostic.py).  Bit-exact: the mask is boolean."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "dresscode_agnostic.npz")


def _cases():
    from oracle import make_golden_agnostic as mg
    gold = np.load(GOLD)
    assert list(gold["cases"]) == [f"{s},{w},{h},{c},{'/'.join(map(str, m))}" for s, w, h, c, m in mg.CASES], "fixture / CASES out of step"
    for n, (seed, w, h, cat, miss) in enumerate(mg.CASES):
        parse, pose = mg.synth_case(seed, w, h, miss)
        yield n, parse, pose, cat, (w, h), np.unpackbits(gold[f"mask_{n}"])[:h * w].reshape(1, h, w).astype(bool)


def test_get_agnostic_equals_the_reference_on_every_fixture_case():
    import torch
    from idm_vton_amd.dresscode import get_agnostic
    seen = 0
    for n, parse, pose, cat, size, want in _cases():
        got = get_agnostic(parse, pose, cat, size)
        assert isinstance(got, torch.Tensor) and got.dtype == torch.bool and tuple(got.shape) == want.shape
        diff = int((got.numpy() != want).sum())
        assert diff == 0, f"case {n} ({cat}, {size}): {diff} pixels differ from the reference's mask"
        assert 0.02 < want.mean() < 0.98                    # a real mask, not a constant
        seen += 1
    assert seen >= 11


def test_fixture_is_what_the_reference_computes_now():
    from oracle import make_golden_agnostic as mg
    if not os.path.isfile(os.path.join(mg.REF, "inference_dc.py")):
        pytest.skip("no reference checkout on this box: the committed fixture stands in")
    fresh, gold = mg.build(mg.reference_get_agnostic()), np.load(GOLD)
    assert sorted(fresh) == sorted(gold.files)
    for k in fresh:
        assert np.array_equal(fresh[k], gold[k]), k


def test_dilate_is_opencv_rect_dilation():
    """cv2.dilate with a k x k rectangle of ones, default anchor k // 2, n iterations == n passes of a maximum filter over
    [x - k // 2, x + k - 1 - k // 2] (scipy.ndimage, an independent implementation), borders never winning."""
    from scipy import ndimage
    from idm_vton_amd.dresscode import dilate
    g = np.random.default_rng(0)
    for (h, w), (kh, kw), it in (((37, 53), (5, 5), 5), ((64, 48), (10, 10), 5), ((90, 70), (20, 20), 5), ((9, 7), (4, 3), 2),
                                 ((5, 5), (2, 2), 1), ((12, 40), (1, 6), 3), ((33, 4), (20, 20), 5)):
        a = (g.random((h, w)) * (g.random((h, w)) > 0.9)).astype(np.float32)
        ref = a
        for _ in range(it):
            ref = ndimage.maximum_filter(ref, size=(kh, kw), mode="constant", cval=-np.inf, origin=0)
        assert np.array_equal(dilate(a, np.ones((kh, kw), np.uint16), iterations=it), ref), ((h, w), (kh, kw), it)
    # known answer: one pixel, even kernel -> the block reaches one further up / left than down / right
    a = np.zeros((9, 9), np.float32)
    a[4, 4] = 1
    d = dilate(a, np.ones((4, 4)), 1)
    ys, xs = np.nonzero(d)
    assert (ys.min(), ys.max(), xs.min(), xs.max()) == (3, 6, 3, 6)      # source at 4: output y sees [y - 2, y + 1] -> y in 3..6
    with pytest.raises(ValueError):
        dilate(a, np.array([[1, 0], [1, 1]]))
    with pytest.raises(ValueError, match="category"):
        from idm_vton_amd.dresscode import get_agnostic
        get_agnostic(np.zeros((8, 8), np.uint8), np.zeros((18, 4)), "shoes", (8, 8))
