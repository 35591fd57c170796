"""Image folder -> (image, mask) training samples; SURVEY.md section 8f row 3.

Mirror of the reference's `src.datasets.eval_dataset.Dataset` (eval_dataset.py:13-52) and of its reader
`cv2_read_rgba` (src/utils/preprocess.py:5-20) without OpenCV: RGBA PNGs are decoded with Pillow, resized on the
uint8 data with a restatement of `cv2.resize(..., interpolation=cv2.INTER_LINEAR)` for 8-bit images (half-pixel
centres, edge clamp, 11-bit fixed-point weights, no anti-aliasing), the mask is `alpha >= 128`, and
`__getitem__` composites over a fresh uniform-random background colour exactly like the reference
(`load_bg_color_fn('random')`, src/utils/prior.py:11-28: one `np.random.uniform(size=(1, 3))` draw per item).

PARITY UNPINNED for the resize: cv2 is not available where the fixtures were generated, so the fixed-point
restatement below follows OpenCV's documented generic (non-SIMD) path, `(b0*S0 + b1*S1 + 2^21) >> 22` on
horizontally pre-multiplied rows; OpenCV's SIMD path rounds in a slightly different order and can differ by one
grey level on isolated pixels.  `tests/test_host_cpu.py` pins it against a float64 bilinear within 1/255.
"""
import glob
import logging
import os

import numpy as np
import torch

logger = logging.getLogger(__name__)

_COEF_BITS = 11
_COEF_SCALE = 1 << _COEF_BITS


def _linear_taps(n_src, n_dst):
    """Source index pairs and 11-bit integer weights of cv2's INTER_LINEAR along one axis."""
    scale = n_src / n_dst
    x = (np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5
    i0 = np.floor(x).astype(np.int64)
    f = (x - i0).astype(np.float32)
    lo = i0 < 0
    i0[lo], f[lo] = 0, 0.0
    hi = i0 >= n_src - 1
    i0[hi], f[hi] = n_src - 1, 0.0
    i1 = np.minimum(i0 + 1, n_src - 1)
    w1 = np.rint(f * _COEF_SCALE).astype(np.int64)  # saturate_cast<short>(f * 2048)
    w0 = _COEF_SCALE - w1
    return i0, i1, w0, w1


def resize_linear_u8(arr, size):
    """arr uint8 (h, w, c), size (w, h) as in cv2.resize -> uint8 (h', w', c)."""
    assert arr.dtype == np.uint8 and arr.ndim == 3, (arr.dtype, arr.shape)
    wd, hd = size
    h, w, _ = arr.shape
    if (h, w) == (hd, wd):
        return arr.copy()
    x0, x1, a0, a1 = _linear_taps(w, wd)
    y0, y1, b0, b1 = _linear_taps(h, hd)
    src = arr.astype(np.int64)
    rows = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]  # (h, w', c), scaled by 2^11
    out = rows[y0] * b0[:, None, None] + rows[y1] * b1[:, None, None]             # scaled by 2^22
    out = (out + (1 << (2 * _COEF_BITS - 1))) >> (2 * _COEF_BITS)
    return np.clip(out, 0, 255).astype(np.uint8)


def read_rgba(path, mask_threshold=128, assert_binary=True, size=None):
    """`cv2_read_rgba` (preprocess.py:5-20): returns (rgba uint8 (h,w,4), rgb uint8 (h,w,3), mask bool (h,w))."""
    from PIL import Image
    try:
        with Image.open(path) as im:
            im.load()
            if im.mode != "RGBA":
                raise AssertionError((np.asarray(im).shape, im.mode))
            arr = np.array(im, dtype=np.uint8)
    except (FileNotFoundError, OSError) as ex:
        raise ValueError(f"failed to read {path}") from ex
    assert arr.shape[2] == 4, arr.shape
    if assert_binary:
        assert np.logical_or(arr[:, :, 3] == 0, arr[:, :, 3] == 255).all(), (np.unique(arr[:, :, 3]), path)
    if size is not None:
        arr = resize_linear_u8(arr, size)
    return arr, arr[:, :, :3], arr[:, :, 3] >= mask_threshold


class Dataset(torch.utils.data.Dataset):
    """Same constructor, fields and item dict as the reference Dataset (eval_dataset.py:13-52)."""

    def __init__(self, resolution, dataset_folder):
        super().__init__()
        self.resolution = resolution
        self.dataset_folder = dataset_folder
        paths = list(sorted(glob.glob(os.path.join(dataset_folder, "*.png"))))
        logger.info(f"found {len(paths)} images in {dataset_folder}")
        self.num_images = len(paths)
        rgb_list, mask_list = [], []
        for path in paths:
            _, rgb, mask = read_rgba(path, size=(self.resolution, self.resolution), assert_binary=False)
            rgb_list.append(rgb)
            mask_list.append(mask)
        shape = (0, self.resolution, self.resolution)
        self.data = {
            "rgb": torch.tensor(np.stack(rgb_list, axis=0) if paths else np.zeros(shape + (3,), np.uint8),
                                dtype=torch.float32).permute(0, 3, 1, 2) / 255.0,       # (n, 3, h, w)
            "alpha": torch.tensor(np.stack(mask_list, axis=0) if paths else np.zeros(shape, bool),
                                  dtype=torch.float32)[:, None, :, :],                   # (n, 1, h, w)
            "path": paths,
        }

    def bg_color_fn(self):
        arr = torch.tensor(np.random.uniform(low=0, high=1, size=(1, 3)), dtype=torch.float32)
        return arr[0, :, None, None].expand(3, self.resolution, self.resolution)

    def __getitem__(self, index):
        rgb = self.data["rgb"][index]
        alpha = self.data["alpha"][index]
        rgb = rgb * alpha + self.bg_color_fn() * (1 - alpha)
        return {"image": rgb, "mask": alpha, "image_path": self.data["path"][index], "pose_indices": index}

    def __len__(self):
        return self.num_images
