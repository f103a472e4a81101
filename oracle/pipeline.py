"""CPU oracle (test infrastructure) of the input pipeline.

Reference: modules/dataloaders_v0623.py:22-37 (384 model: Resize(448) -> RandomCrop(384) -> RandomRotation(5) -> ToTensor ->
Normalize; eval Resize(448) -> CenterCrop(384)), modules/dataloaders_v0401.py:25-37 (224 model: Resize(256) -> RandomCrop(224) ->
RandomHorizontalFlip; eval Resize((224, 224))), the collate contract dataloaders_v0623.py:60-116 and the Multi-view-CXR
anchor collate modules/multiview/dataloaders.py:50-106.

The transform arithmetic lives in un-vendored torchvision (pinned 0.16.2, README.md:122; not installed here) whose PIL
backend forwards to Pillow: F.resize -> Image.resize(size, BILINEAR), F.rotate -> Image.rotate(angle, NEAREST, expand=False,
fillcolor=0), F.to_tensor -> uint8 / 255, F.normalize -> (x - mean) / std.  `transform_pil` restates that stack on
Pillow itself (installed: the same library the reference would call), with torchvision's size rule
(_compute_resized_output_size) restated; `resize_u8` / `rotate_nearest_u8` additionally restate Pillow's integer
algorithms in numpy and are pinned against Pillow in tests/test_pipeline_cpu.py.  Parity w.r.t. torchvision itself:
unpinned (package absent); w.r.t. Pillow: pinned.
"""
import math

import numpy as np

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
PRECISION_BITS = 32 - 8 - 2


def resized_size(w, h, size):
    """torchvision.transforms.functional._compute_resized_output_size for an int `size` (shorter side -> size)."""
    if isinstance(size, (tuple, list)) and len(size) == 2:
        return int(size[1]), int(size[0])                  # (h, w) given explicitly -> (w, h)
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_short, new_long) if w <= h else (new_long, new_short)


def center_crop_origin(w, h, size):
    """torchvision F.center_crop: top = int(round((h - th) / 2.0)), left likewise."""
    return int(round((h - size) / 2.0)), int(round((w - size) / 2.0))


def transform_pil(img_u8, resize, crop_top, crop_left, out_size, flip=False, angle=None, mean=MEAN, std=STD):
    """img_u8: (H, W, 3) or (H, W) uint8 -> float32 (3, S, S), through Pillow like the reference's transform stack."""
    from PIL import Image
    im = Image.fromarray(img_u8).convert('RGB')
    w, h = im.size
    rw, rh = resized_size(w, h, resize)
    if (rw, rh) != (w, h):
        im = im.resize((rw, rh), Image.BILINEAR)
    im = im.crop((crop_left, crop_top, crop_left + out_size, crop_top + out_size))
    if flip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    if angle is not None:
        im = im.rotate(angle, Image.NEAREST, False, None, fillcolor=0)
    x = np.asarray(im, dtype=np.uint8).astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


# ---------------------------------------------------------------------------------------------------------------
# numpy restatement of Pillow's integer algorithms (Resample.c precompute_coeffs / normalize_coeffs_8bpc /
# ImagingResampleHorizontal_8bpc / ..Vertical_8bpc; Image.rotate + Geometry.c affine_fixed)
# ---------------------------------------------------------------------------------------------------------------
def bilinear_coeffs(in_size, out_size):
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = np.zeros(ksize)
        ww = 0.0
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            k[x] = 1.0 - t if t < 1.0 else 0.0
            ww += k[x]
        if ww != 0.0:
            k[:xmax] = k[:xmax] / ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + k[x] * (1 << PRECISION_BITS)) if k[x] < 0 else int(0.5 + k[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    img = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.zeros((len(bounds),) + img.shape[1:], np.int64)
    for i, (lo, n) in enumerate(bounds):
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for j in range(n):
            acc = acc + img[lo + j] * kk[i, j]
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def resize_u8(img_u8, rw, rh):
    """Pillow Image.resize((rw, rh), BILINEAR) of an (H, W, C) uint8 image: horizontal pass, then vertical pass."""
    h, w = img_u8.shape[:2]
    out = img_u8
    if rw != w:
        out = _pass(out, *bilinear_coeffs(w, rw), axis=1)
    if rh != h:
        out = _pass(out, *bilinear_coeffs(h, rh), axis=0)
    return out


def rotate_affine_fixed(angle, w, h):
    """Image.rotate(angle, NEAREST, expand=False, center=None) -> the six 16.16 integers of Geometry.c affine_fixed, or None
    when Pillow takes a shortcut (angle % 360 == 0: plain copy)."""
    angle = angle % 360.0
    if angle == 0:
        return None
    if angle in (90, 180, 270):
        raise NotImplementedError('Pillow transposes for multiples of 90 degrees; not part of RandomRotation(5)')
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2] + cx
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5] + cy

    def fix(v):
        return int(math.floor(v * 65536.0 + 0.5))
    return [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def rotate_nearest_u8(img_u8, angle):
    h, w = img_u8.shape[:2]
    aff = rotate_affine_fixed(angle, w, h)
    if aff is None:
        return img_u8.copy()
    a0, a1, a2, a3, a4, a5 = aff
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
    sx = (a2 + xs * a0 + ys * a1) >> 16
    sy = (a5 + xs * a3 + ys * a4) >> 16
    ok = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
    out = np.zeros_like(img_u8)
    out[ok] = img_u8[sy[ok], sx[ok]]
    return out


def transform_numpy(img_u8, resize, crop_top, crop_left, out_size, flip=False, angle=None, mean=MEAN, std=STD):
    """Same contract as transform_pil, through the numpy restatements only."""
    if img_u8.ndim == 2:
        img_u8 = np.repeat(img_u8[:, :, None], 3, axis=2)
    h, w = img_u8.shape[:2]
    rw, rh = resized_size(w, h, resize)
    im = resize_u8(img_u8, rw, rh)
    im = im[crop_top:crop_top + out_size, crop_left:crop_left + out_size]
    if flip:
        im = im[:, ::-1]
    if angle is not None:
        im = rotate_nearest_u8(np.ascontiguousarray(im), angle)
    x = im.astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


# ---------------------------------------------------------------------------------------------------------------
# collate contracts
# ---------------------------------------------------------------------------------------------------------------
def collate_order(batch_images_list, multiview_images_list, is_multiview_learning=True):
    """dataloaders_v0623.py:76-113: anchors first, then every not-yet-seen other view -> (image paths, patient ids)."""
    paths, pids, info = [], [], []
    for p in batch_images_list:
        sp = p.split('/')
        assert len(sp) == 4
        info.append('_'.join(sp[1:]))
        pids.append('_'.join(sp[1:3]))
        paths.append(p)
    if is_multiview_learning:
        for mvs in multiview_images_list:
            for p in mvs:
                sp = p.split('/')
                assert len(sp) == 4
                key = '_'.join(sp[1:])
                if key not in info:
                    info.append(key)
                    pids.append('_'.join(sp[1:3]))
                    paths.append(p)
    return paths, pids


def multiview_anchor_index(view_position, randint):
    """modules/multiview/dataloaders.py:70-90: index of the anchor view of one study; randint(lo, hi) like np.random.randint."""
    anchors = ['AP', 'PA', 'PAO', 'LAO']
    if any(vp in view_position for vp in anchors):
        idx = [k for k, vp in enumerate(view_position) if vp in anchors]
        return idx[randint(0, len(idx))]
    if all(vp == 'unk' for vp in view_position):
        return 0
    cand = [k for k, vp in enumerate(view_position) if vp not in ['LATERAL', 'LL']]
    if len(cand) == 0:
        return randint(0, len(view_position))
    return cand[randint(0, len(cand))]


def multiview_collate_order(image_ids, batch_view_position, batch_images_list, randint):
    """modules/multiview/dataloaders.py:67-105: all anchors (one per study, in study order), then all other views."""
    a_paths, a_ids, o_paths, o_ids = [], [], [], []
    for ids, vps, paths in zip(image_ids, batch_view_position, batch_images_list):
        r = multiview_anchor_index(list(vps), randint)
        for j, p in enumerate(paths):
            if j != r:
                o_paths.append(p)
                o_ids.append(ids)
            else:
                a_paths.append(p)
                a_ids.append(ids)
    return a_paths + o_paths, a_ids + o_ids


def pad_tokens(ids_list, masks_list):
    """dataloaders_v0623.py:63-73: right-pad with zeros to the longest sequence of the batch."""
    n = max(len(x) for x in ids_list)
    ids = np.zeros((len(ids_list), n), dtype=np.int64)
    masks = np.zeros((len(ids_list), n), dtype=np.int64)
    for i, (a, m) in enumerate(zip(ids_list, masks_list)):
        ids[i, :len(a)] = a
        masks[i, :len(m)] = m
    return ids, masks
