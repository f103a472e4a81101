"""Input pipeline on the HIP engine: the collate contract of the reference's loaders and the
torchvision transform stack as a GPU pre-processing step reading uint8 pixels.

Mirrors: modules/dataloaders_v0623.py:22-37 (384 model transforms), :60-116 (collate: anchors first, then every
not-yet-seen other view; patient id = subject_study), modules/dataloaders_v0401.py:25-37 (224 model transforms),
modules/multiview/dataloaders.py:50-106 (Multi-view-CXR anchor selection).

The reference decodes and transforms every image with PIL inside collate_fn on the CPU; here the decoded uint8 pixels
are uploaded once and `preprocess_batch` runs resize / crop / flip / rotate / normalise on the GPU
(csrc/preproc.hip, bit-exact with Pillow's integer resampling), writing straight into the (N, 3, S, S) f32 batch tensor
that FineTune / Pretrain consume.  Random parameters are drawn on the host exactly where torchvision draws them
(RandomCrop.get_params, RandomRotation.get_params, RandomHorizontalFlip)."""
import ctypes as C
import math

import numpy as np
import torch

from . import hip as H

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


# ----------------------------------------------------------------------------------------------------
# transform parameters (host)
# ----------------------------------------------------------------------------------------------------
def resized_size(w, h, size):
    """torchvision _compute_resized_output_size: int -> shorter side becomes `size`; (h, w) -> exactly that."""
    if isinstance(size, (tuple, list)) and len(size) == 2:
        return int(size[1]), int(size[0])
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_short, new_long) if w <= h else (new_long, new_short)


def rotate_affine_fixed(angle, w, h):
    """PIL Image.rotate(angle, NEAREST, expand=False) -> the 16.16 fixed-point inverse map of Geometry.c affine_fixed
    (None for angle % 360 == 0, which Pillow turns into a copy)."""
    angle = angle % 360.0
    if angle == 0:
        return None
    if angle in (90, 180, 270):
        raise NotImplementedError('multiples of 90 degrees are transposes in Pillow; RandomRotation(5) never draws them')
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2] + cx
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5] + cy

    def fix(v):
        return int(math.floor(v * 65536.0 + 0.5))
    return [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


class Transform:
    """One of the reference's four transform stacks; `params(w, h)` draws what torchvision draws for one image."""

    def __init__(self, resize, out_size, random_crop, rotate_degrees=0.0, random_flip=False, generator=None):
        self.resize, self.out_size, self.random_crop = resize, out_size, random_crop
        self.rotate_degrees, self.random_flip, self.generator = rotate_degrees, random_flip, generator

    @classmethod
    def for_model(cls, resolution, split, generator=None):
        train = split == 'train'
        if resolution == 384:       # dataloaders_v0623.py:22-37
            return cls(448, 384, train, 5.0 if train else 0.0, False, generator)
        if resolution == 224:       # dataloaders_v0401.py:25-37
            return cls(256, 224, True, 0.0, True, generator) if train else cls((224, 224), 224, False, 0.0, False, generator)
        raise ValueError('resolution must be 224 or 384')

    def params(self, w, h):
        rw, rh = resized_size(w, h, self.resize)
        S = self.out_size
        if rw < S or rh < S:
            raise ValueError('Required crop size %s is larger than input image size %s' % ((S, S), (rh, rw)))
        if self.random_crop:        # RandomCrop.get_params: nothing drawn when the sizes match, else i then j
            top = left = 0
            if not (rw == S and rh == S):
                top = int(torch.randint(0, rh - S + 1, (1,), generator=self.generator))
                left = int(torch.randint(0, rw - S + 1, (1,), generator=self.generator))
        else:                       # CenterCrop
            top, left = int(round((rh - S) / 2.0)), int(round((rw - S) / 2.0))
        flip = bool(self.random_flip and float(torch.rand(1, generator=self.generator)) < 0.5)
        angle = None
        if self.rotate_degrees:
            angle = float(torch.empty(1).uniform_(-self.rotate_degrees, self.rotate_degrees, generator=self.generator))
        return dict(resize_w=rw, resize_h=rh, crop_top=top, crop_left=left, out_size=S, flip=flip, angle=angle)


# ----------------------------------------------------------------------------------------------------
# GPU pre-processing
# ----------------------------------------------------------------------------------------------------
def preprocess_into(out_chw, img_u8, prm, mean=MEAN, std=STD):
    """img_u8: CUDA uint8 (H, W, 3) or (H, W); out_chw: CUDA f32 (3, S, S) view to fill."""
    assert img_u8.is_cuda and img_u8.dtype == torch.uint8 and img_u8.is_contiguous()
    assert out_chw.is_cuda and out_chw.dtype == torch.float32 and out_chw.is_contiguous()
    d = H.PreprocDesc()
    d.src = img_u8.data_ptr()
    d.src_h, d.src_w = img_u8.shape[0], img_u8.shape[1]
    d.channels = 1 if img_u8.dim() == 2 else img_u8.shape[2]
    d.resize_h, d.resize_w = prm['resize_h'], prm['resize_w']
    d.crop_top, d.crop_left, d.out_size = prm['crop_top'], prm['crop_left'], prm['out_size']
    d.flip = int(bool(prm.get('flip', False)))
    aff = rotate_affine_fixed(prm['angle'], prm['out_size'], prm['out_size']) if prm.get('angle') is not None else None
    d.rotate = int(aff is not None)
    for i in range(6):
        d.affine[i] = aff[i] if aff is not None else 0
    for i in range(3):
        d.mean[i], d.std[i] = mean[i], std[i]
    nb = H.lib.evk_preprocess_ws_bytes(C.byref(d))
    if nb < 0:
        raise ValueError('evk_preprocess_ws_bytes: bad descriptor')
    ws = torch.empty(nb, dtype=torch.uint8, device=img_u8.device)
    H.check(H.lib.evk_preprocess_image(C.byref(d), H.ptr(ws), nb, H.ptr(out_chw), H.stream()), 'preprocess_image')


def preprocess_batch(images_u8, transform, device='cuda'):
    """images_u8: list of uint8 arrays/tensors (H, W[, 3]) as decoded (.convert('RGB') not needed for grey inputs).
    Returns the (N, 3, S, S) f32 batch tensor and the parameters drawn per image."""
    S = transform.out_size
    out = torch.empty(len(images_u8), 3, S, S, dtype=torch.float32, device=device)
    prms = []
    for i, im in enumerate(images_u8):
        t = torch.as_tensor(np.ascontiguousarray(im) if isinstance(im, np.ndarray) else im)
        t = t.to(device, non_blocking=True).contiguous()
        prm = transform.params(t.shape[1], t.shape[0])
        preprocess_into(out[i], t, prm)
        prms.append(prm)
    return out, prms


# ----------------------------------------------------------------------------------------------------
# collate contracts (host)
# ----------------------------------------------------------------------------------------------------
def collate_order(batch_images_list, multiview_images_list, is_multiview_learning=True):
    """dataloaders_v0623.py:76-113 -> (image paths in batch order, patient ids): the anchor image of every sample first,
    then each other view that is not in the batch yet; patient id = '<subject>_<study>' (path components 1-2)."""
    paths, pids, seen = [], [], []
    for p in batch_images_list:
        sp = p.split('/')
        if len(sp) != 4:
            raise AssertionError('image path must have exactly 4 components: %s' % p)
        seen.append('_'.join(sp[1:]))
        pids.append('_'.join(sp[1:3]))
        paths.append(p)
    if is_multiview_learning:
        for views in multiview_images_list:
            for p in views:
                sp = p.split('/')
                if len(sp) != 4:
                    raise AssertionError('image path must have exactly 4 components: %s' % p)
                key = '_'.join(sp[1:])
                if key not in seen:
                    seen.append(key)
                    pids.append('_'.join(sp[1:3]))
                    paths.append(p)
    return paths, np.array(pids)


ANCHOR_VIEW_POSITIONS = ('AP', 'PA', 'PAO', 'LAO')


def multiview_collate_order(image_ids, batch_view_position, batch_images_list, randint=None):
    """modules/multiview/dataloaders.py:67-105 -> (image paths, patient ids): one anchor per study (frontal views preferred,
    laterals avoided), all anchors first, then all remaining views.  randint(lo, hi) defaults to np.random.randint."""
    randint = randint or np.random.randint
    a_paths, a_ids, o_paths, o_ids = [], [], [], []
    for ids, vps, paths in zip(image_ids, batch_view_position, batch_images_list):
        vps = list(vps)
        if any(vp in vps for vp in ANCHOR_VIEW_POSITIONS):
            idx = [k for k, vp in enumerate(vps) if vp in ANCHOR_VIEW_POSITIONS]
            r = idx[randint(0, len(idx))]
        elif all(vp == 'unk' for vp in vps):
            r = 0
        else:
            cand = [k for k, vp in enumerate(vps) if vp not in ('LATERAL', 'LL')]
            r = randint(0, len(vps)) if len(cand) == 0 else cand[randint(0, len(cand))]
        for j, p in enumerate(paths):
            if j == r:
                a_paths.append(p)
                a_ids.append(ids)
            else:
                o_paths.append(p)
                o_ids.append(ids)
    return a_paths + o_paths, np.array(a_ids + o_ids)


def pad_tokens(ids_list, masks_list):
    """dataloaders_v0623.py:63-73: zero right-padding to the longest sequence -> LongTensors."""
    n = max(len(x) for x in ids_list)
    ids = np.zeros((len(ids_list), n), dtype=np.int64)
    masks = np.zeros((len(ids_list), n), dtype=np.int64)
    for i, (a, m) in enumerate(zip(ids_list, masks_list)):
        ids[i, :len(a)] = a
        masks[i, :len(m)] = m
    return torch.from_numpy(ids), torch.from_numpy(masks)
