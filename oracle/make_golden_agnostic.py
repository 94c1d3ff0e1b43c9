"""Golden vectors for the DressCode agnostic mask: the REFERENCE'S OWN `get_agnostic` executed on seeded synthetic inputs.
TEST INFRASTRUCTURE (see oracle/__init__.py); runs only where /root/reference exists (the build container).

What is executed: the `label_map` table (/root/reference/inference_dc.py:49-68) and the method `DresscodeTestDataset.get_agnostic`
(:231-352), taken from the file's syntax tree and compiled as they stand -- NOT the module: importing inference_dc.py pulls accelerate,
diffusers and a CUDA-era torchvision, none of which the mask needs.  The names the method uses resolve to what the script imports them
as: np, torch, PIL.Image / ImageDraw, numpy.linalg.lstsq -- and `cv2`, which this image does not have: its one call, cv2.dilate with a
rectangle of ones, is supplied here by scipy.ndimage.maximum_filter (window [x - k // 2, x + k - 1 - k // 2] = OpenCV's default anchor;
outside pixels never win), an implementation independent of the product's idm_vton_amd/dresscode.py:dilate.  So the fixture pins
everything first-party in the mask; OpenCV's own dilate stays unpinned (absent here) and is restated from its documented definition.

  python oracle/make_golden_agnostic.py [tests/golden/dresscode_agnostic.npz]
"""
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("IDMVTON_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "dresscode_agnostic.npz")


def _cv2_standin():
    from scipy import ndimage

    def dilate(src, kernel, iterations=1):
        a = np.asarray(src, dtype=np.float32)
        k = np.asarray(kernel)
        assert k.ndim == 2 and np.all(k != 0), "the reference only dilates with rectangles of ones"
        for _ in range(iterations):
            a = ndimage.maximum_filter(a, size=k.shape, mode="constant", cval=-np.inf, origin=0)
        return a
    return types.SimpleNamespace(dilate=dilate)


def reference_get_agnostic(path=None):
    """-> the reference's get_agnostic as a plain function (parse_array, pose_data, category, size), compiled from its source file."""
    import torch
    from numpy.linalg import lstsq
    from PIL import Image, ImageDraw
    path = path or os.path.join(REF, "inference_dc.py")
    tree = ast.parse(open(path).read(), path)
    keep = []
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "label_map" for t in node.targets):
            keep.append(node)
        if isinstance(node, ast.ClassDef) and node.name == "DresscodeTestDataset":
            keep += [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name == "get_agnostic"]
    assert len(keep) == 2, "inference_dc.py: label_map / DresscodeTestDataset.get_agnostic not found"
    ns = dict(np=np, torch=torch, Image=Image, ImageDraw=ImageDraw, lstsq=lstsq, cv2=_cv2_standin())
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    fn = ns["get_agnostic"]
    return lambda parse_array, pose_data, category, size: fn(None, parse_array, pose_data, category, size)


def synth_case(seed, width, height, missing=()):
    """A person-shaped label map (blocks of the 18 DressCode labels, jittered by the seed) and OpenPose-style keypoints in the 384 x 512
    annotation frame; `missing` joints sit at (0, 0) like undetected ones."""
    g = np.random.default_rng(seed)
    parse = np.zeros((height, width), dtype=np.uint8)
    sx, sy = width / 384.0, height / 512.0
    j = lambda lo, hi: int(g.integers(lo, hi + 1))

    def box(label, x0, y0, x1, y1):
        parse[int(max(0, y0) * sy):int(min(512, y1) * sy), int(max(0, x0) * sx):int(min(384, x1) * sx)] = label
    cx = 192 + j(-15, 15)
    box(2, cx - 40, 20, cx + 40, 70)                     # hair
    box(1, cx - 30, 10 + j(0, 5), cx + 30, 30)           # hat
    box(11, cx - 30, 60, cx + 30, 125 + j(0, 10))        # head (face + neck)
    box(3, cx - 20, 75, cx + 20, 85)                     # sunglasses
    box(4, cx - 75, 125, cx + 75, 290 + j(-10, 10))      # upper clothes
    box(14, cx - 115 - j(0, 10), 135, cx - 75, 330)      # left arm
    box(15, cx + 75, 135, cx + 115 + j(0, 10), 330)      # right arm
    box(5 if seed % 2 else 6, cx - 70, 290, cx + 70, 420)    # skirt / pants
    box(7, cx - 20, 200, cx + 20, 240) if seed % 3 == 0 else None   # a patch of 'dress'
    box(12, cx - 60, 420, cx - 10, 490)                  # legs
    box(13, cx + 10, 420, cx + 60, 490)
    box(9, cx - 65, 490, cx - 5, 508)                    # shoes
    box(10, cx + 5, 490, cx + 65, 508)
    box(16, cx + 120, 250, cx + 170, 330)                # bag
    box(17, cx - 35, 118, cx + 35, 132)                  # scarf
    box(8, cx - 70, 285, cx + 70, 295)                   # belt
    pose = np.zeros((18, 4), dtype=np.float64)
    pts = {0: (cx, 80), 1: (cx, 130), 2: (cx - 70 + j(-5, 5), 135 + j(-6, 6)), 5: (cx + 70 + j(-5, 5), 135 + j(-6, 6)),
           3: (cx - 95, 230 + j(-10, 10)), 6: (cx + 95, 230 + j(-10, 10)), 4: (cx - 100 + j(-8, 8), 320), 7: (cx + 100 + j(-8, 8), 320),
           8: (cx - 35, 300), 11: (cx + 35, 300), 9: (cx - 35, 420), 12: (cx + 35, 420), 10: (cx - 35, 495), 13: (cx + 35, 495)}
    for i, (x, y) in pts.items():
        pose[i, :2] = (x, y)
        pose[i, 2] = 0.9
    for i in missing:
        pose[i] = 0.0
    return parse, pose


# (seed, width, height, category, joints OpenPose "missed"): all three dilation branches (height > 512, > 256, else), all categories, every
# branch of the arm-stroke chain (:299-313)
CASES = [(1, 384, 512, "upper_body", ()), (2, 384, 512, "dresses", ()), (3, 384, 512, "lower_body", ()),
         (4, 768, 1024, "upper_body", ()), (5, 768, 1024, "dresses", (4,)), (6, 768, 1024, "lower_body", ()),
         (7, 192, 256, "upper_body", ()), (8, 384, 512, "upper_body", (4, 3)), (9, 384, 512, "upper_body", (7,)),
         (10, 384, 512, "dresses", (7, 6)), (11, 576, 768, "upper_body", (4,))]


def build(fn):
    out = {"cases": np.array([f"{s},{w},{h},{c},{'/'.join(map(str, m))}" for s, w, h, c, m in CASES])}
    for n, (seed, w, h, cat, miss) in enumerate(CASES):
        parse, pose = synth_case(seed, w, h, miss)
        mask = fn(parse, pose, cat, (w, h))
        mask = np.asarray(mask).astype(bool)
        assert mask.shape == (1, h, w), mask.shape
        out[f"mask_{n}"] = np.packbits(mask.reshape(-1))
    return out


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else OUT
    np.savez_compressed(out, **build(reference_get_agnostic()))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
