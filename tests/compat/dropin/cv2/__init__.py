"""`cv2` name shim for the one OpenCV call of the reference's DressCode script (`cv2.dilate`, inference_dc.py:315-344: mask dilation in
`get_agnostic`).  Test / demo environment only (no opencv wheel in this image)."""
import numpy as np
import torch
import torch.nn.functional as F


def dilate(src, kernel, iterations=1):
    """Grayscale dilation with a rectangular structuring element of ones, OpenCV's anchor convention (anchor = k // 2: for an
    even k the window reaches one pixel further up/left than down/right)."""
    a = np.asarray(src, dtype=np.float32)
    kh, kw = np.asarray(kernel).shape
    t = torch.from_numpy(a)[None, None]
    for _ in range(iterations):
        t = F.pad(t, (kw // 2, kw - 1 - kw // 2, kh // 2, kh - 1 - kh // 2), value=float("-inf"))
        t = F.max_pool2d(t, (kh, kw), stride=1)
    return t[0, 0].numpy()
