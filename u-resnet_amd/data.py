"""The step in front of the hot path, on the GPU: what the reference's Cityscapes dataset class does per sample on the
CPU (Segmentation/Cityscapes/DataInstance/Cityscapes.py) for a whole uint8 batch already in device memory.

  label remap   valid_classes / class_map / ignore_index                       Cityscapes.py:191-194, 220-223
  normalise     A.Normalize(mean=(0.485,0.456,0.406), std=(0.229,0.224,0.225)) Cityscapes.py:103, 256
                image / 255.0 when no transform is given                       Cityscapes.py:87, 238
  ToTensorV2    HWC uint8 -> CHW float: here straight to the NHWC activation the kernels consume
"""
import ctypes as C

import numpy as np
import torch

from . import ops
from ._lib import call

VALID_CLASSES = (7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33)    # Cityscapes.py:191-192
IGNORE_INDEX = 255                                                                            # Cityscapes.py:194
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def class_lut(valid_classes=VALID_CLASSES, ignore_index=IGNORE_INDEX):
    """256-entry table of Cityscapes.py:220-222: valid id k -> its position in valid_classes, every other id -> ignore."""
    lut = np.full(256, ignore_index, dtype=np.uint8)
    for i, k in enumerate(valid_classes):
        lut[k] = i
    return lut


def normalize_consts(mean=IMAGENET_MEAN, std=IMAGENET_STD, max_pixel_value=255.0):
    """(sub, mul, divide) in float32 the way albumentations' Normalize derives them: mean * max, reciprocal(std * max).
    mean=None gives the reference's no-transform branch, image / 255.0 (a true division)."""
    if mean is None:
        n = 3 if std is None else len(std)
        return np.zeros(n, dtype=np.float32), np.full(n, max_pixel_value, dtype=np.float32), 1
    sub = np.array(mean, dtype=np.float32) * np.float32(max_pixel_value)
    mul = np.reciprocal(np.array(std, dtype=np.float32) * np.float32(max_pixel_value), dtype=np.float32)
    return sub, mul, 0


def pack_images(images_u8, dtype=torch.bfloat16, mean=IMAGENET_MEAN, std=IMAGENET_STD, pad_to=None):
    """[N, H, W, C<=4] uint8 on the GPU -> normalised activation of logical shape [N, C(or pad_to), H, W] in NHWC memory
    (extra channels zero).  One pass: normalise, convert and lay out for the first convolutions."""
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or not images_u8.is_cuda:
        raise RuntimeError("pack_images: expected a uint8 [N, H, W, C] tensor on the GPU")
    images_u8 = images_u8.contiguous()
    n, h, w, c = images_u8.shape
    oc = c if pad_to is None else int(pad_to)
    sub, mul, divide = normalize_consts(mean, std)
    if len(sub) != c or c > 4 or oc < c:
        raise RuntimeError("pack_images: %d channels, %d normalisation constants, pad_to=%r" % (c, len(sub), pad_to))
    out = ops.empty_nhwc(n, oc, h, w, dtype, images_u8.device)
    call("ur_pack_image_u8", images_u8.data_ptr(), out.data_ptr(), ops._dt(out), n * h * w, c, oc,
         sub.ctypes.data_as(C.c_void_p), mul.ctypes.data_as(C.c_void_p), divide, ops._stream())
    return out


def remap_labels(label_ids_u8, lut=None):
    """[N, H, W] uint8 label ids on the GPU -> int64 training targets (19 classes, 255 = ignore)."""
    if label_ids_u8.dtype != torch.uint8 or not label_ids_u8.is_cuda:
        raise RuntimeError("remap_labels: expected a uint8 tensor on the GPU")
    label_ids_u8 = label_ids_u8.contiguous()
    lut = class_lut() if lut is None else np.ascontiguousarray(lut, dtype=np.uint8)
    if lut.size != 256:
        raise RuntimeError("remap_labels: the table must have 256 entries")
    out = torch.empty(label_ids_u8.shape, dtype=torch.int64, device=label_ids_u8.device)
    call("ur_remap_labels_u8", label_ids_u8.data_ptr(), out.data_ptr(), label_ids_u8.numel(), lut.ctypes.data_as(C.c_void_p),
         ops._stream())
    return out
