"""Writes a synthetic VITON-HD-layout test set (read by VitonHDTestDataset, inference.py:75-196):
test_pairs.txt, test/vitonhd_test_tagged.json, test/{image,cloth,agnostic-mask,image-densepose}/.  Seeded noise images.
  python tools/make_synth_vitonhd.py <out_dir> [--n 2] [--width 768 --height 1024] [--seed 42]"""
import argparse
import json
import os

import numpy as np
from PIL import Image


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=42)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    D, W, H = a.out, a.width, a.height
    for sub in ("image", "cloth", "agnostic-mask", "image-densepose"):
        os.makedirs(os.path.join(D, "test", sub), exist_ok=True)
    names, tagged = [], []
    for i in range(a.n):
        name = f"{i:05d}_00.jpg"
        names.append(name)
        smooth = lambda: np.kron(rng.integers(0, 256, (H // 32, W // 32, 3), dtype=np.uint8), np.ones((32, 32, 1), dtype=np.uint8))
        Image.fromarray(smooth()).save(os.path.join(D, "test", "image", name), quality=95)
        Image.fromarray(smooth()).save(os.path.join(D, "test", "cloth", name), quality=95)
        Image.fromarray(smooth()).save(os.path.join(D, "test", "image-densepose", name), quality=95)
        m = np.zeros((H, W), dtype=np.uint8)
        m[H // 5: 7 * H // 10, W // 5: 4 * W // 5] = 255                          # white = region to inpaint (torso)
        Image.fromarray(m).convert("RGB").save(os.path.join(D, "test", "agnostic-mask", name.replace(".jpg", "_mask.png")))
        tagged.append(dict(file_name=name, tag_info=[dict(tag_name="sleeveLength", tag_category="short sleeve"),
                                                      dict(tag_name="neckLine", tag_category=None), dict(tag_name="item", tag_category="t-shirts")]))
    json.dump({"data": tagged}, open(os.path.join(D, "test", "vitonhd_test_tagged.json"), "w"))
    with open(os.path.join(D, "test_pairs.txt"), "w") as f:
        for i, n in enumerate(names):
            f.write(f"{n} {names[(i + 1) % len(names)]}\n")
    print("wrote", D, a.n, "pairs", (W, H))


if __name__ == "__main__":
    main()
