"""Writes a synthetic DressCode-layout test set for one category (read by DresscodeTestDataset, /root/reference/inference_dc.py:96-226):
<category>/test_pairs_{paired,unpaired}.txt, dc_caption.txt, images/ (<id>_0.jpg person, <id>_1.jpg garment), skeletons/<id>_5.jpg,
label_maps/<id>_4.png (class ids of inference_dc.py:49-68), keypoints/<id>_2.json (18 x 4, 384x512 coordinates), image-densepose/.
  python tools/make_synth_dresscode.py <out_dir> [--category upper_body] [--n 2] [--width 768 --height 1024]"""
import argparse
import json
import os

import numpy as np
from PIL import Image


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--category", default="upper_body", choices=["upper_body", "lower_body", "dresses"])
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=42)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    W, H = a.width, a.height
    D = os.path.join(a.out, a.category)
    for sub in ("images", "skeletons", "label_maps", "keypoints", "image-densepose"):
        os.makedirs(os.path.join(D, sub), exist_ok=True)
    smooth = lambda: np.kron(rng.integers(0, 256, (H // 32, W // 32, 3), dtype=np.uint8), np.ones((32, 32, 1), dtype=np.uint8))
    pairs = []
    for i in range(a.n):
        pid = f"{i:06d}"
        Image.fromarray(smooth()).save(os.path.join(D, "images", pid + "_0.jpg"), quality=95)
        Image.fromarray(smooth()).save(os.path.join(D, "images", pid + "_1.jpg"), quality=95)
        Image.fromarray(smooth()).save(os.path.join(D, "skeletons", pid + "_5.jpg"), quality=95)
        Image.fromarray(smooth()).save(os.path.join(D, "image-densepose", pid + "_0.jpg"), quality=95)
        lab = np.zeros((H, W), dtype=np.uint8)                               # a crude figure: head, hair, torso, arms, legs, shoes
        cx = W // 2
        lab[H // 16: H // 8, cx - W // 12: cx + W // 12] = 2                  # hair
        lab[H // 8: H // 5, cx - W // 12: cx + W // 12] = 11                  # head
        lab[H // 5: H // 2, cx - W // 5: cx + W // 5] = 4                     # upper_clothes
        lab[H // 5: H // 2, cx - W // 3: cx - W // 5] = 14                    # left arm
        lab[H // 5: H // 2, cx + W // 5: cx + W // 3] = 15                    # right arm
        lab[H // 2: 9 * H // 10, cx - W // 6: cx + W // 6] = 6                # pants
        lab[9 * H // 10: 19 * H // 20, cx - W // 6: cx - W // 24] = 9
        lab[9 * H // 10: 19 * H // 20, cx + W // 24: cx + W // 6] = 10
        Image.fromarray(lab, mode="L").save(os.path.join(D, "label_maps", pid + "_4.png"))
        # 18 OpenPose keypoints in the 384x512 frame the script rescales from; (x, y, score, id)
        k = np.zeros((18, 4), dtype=np.float32)
        sx, sy = 384.0, 512.0
        pts = {0: (0.5, 0.15), 1: (0.5, 0.22), 2: (0.36, 0.24), 3: (0.30, 0.36), 4: (0.27, 0.48), 5: (0.64, 0.24), 6: (0.70, 0.36),
               7: (0.73, 0.48), 8: (0.42, 0.52), 9: (0.42, 0.72), 10: (0.42, 0.90), 11: (0.58, 0.52), 12: (0.58, 0.72), 13: (0.58, 0.90)}
        for j, (x, y) in pts.items():
            k[j] = (x * sx, y * sy, 0.9, j)
        json.dump({"keypoints": k.tolist()}, open(os.path.join(D, "keypoints", pid + "_2.json"), "w"))
        pairs.append((pid + "_0.jpg", pid + "_1.jpg"))
    for order in ("paired", "unpaired"):
        with open(os.path.join(D, f"test_pairs_{order}.txt"), "w") as f:
            for j, (im, c) in enumerate(pairs):
                f.write(f"{im} {c if order == 'paired' else pairs[(j + 1) % len(pairs)][1]}\n")
    with open(os.path.join(D, "dc_caption.txt"), "w") as f:
        for _, c in pairs:
            f.write(f"{c} short sleeve round neck t-shirts\n")
    print("wrote", D, a.n, "pairs", (W, H))


if __name__ == "__main__":
    main()
