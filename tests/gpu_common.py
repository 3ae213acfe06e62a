"""Shared helpers for the -m gpu parity tests (the HIP path against the CPU oracle, through the C-ABI)."""
import os
import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CACHE = {}


def seeded_weights(scale="l", nc=5):
    from caesar_yolo_amd import weights as W
    key = ("w", scale, nc)
    if key not in _CACHE:
        path = os.path.join("/tmp", "cy_test_seeded_%s_%d.cyw" % (scale, nc))
        if not os.path.exists(path):
            W.make_seeded_file(path, scale, nc)
        _CACHE[key] = (path,) + W.read_cyw(path)[:3]
    return _CACHE[key]            # path, scale, names, weight dict


def detector(precision, max_batch=4, max_imgsz=640, scale="l", nc=5):
    from caesar_yolo_amd.model import HipDetector
    key = ("d", precision, max_batch, max_imgsz, scale, nc)
    if key not in _CACHE:
        _CACHE[key] = HipDetector(seeded_weights(scale, nc)[0], device=0, precision=precision, max_batch=max_batch,
                                  max_imgsz=max_imgsz)
    return _CACHE[key]


def oracle_model(scale="l", nc=5):
    from oracle import yolov8_ref as Y
    key = ("o", scale, nc)
    if key not in _CACHE:
        _, sc, names, w = seeded_weights(scale, nc)
        _CACHE[key] = Y.OracleYOLO(w, names, sc)
    return _CACHE[key]


def netin_from_chw(x_bchw, dtype):
    """oracle network input [B,3,H,W] fp32 (already flipped and /255) -> NHWC4 device tensor."""
    b, _, h, w = x_bchw.shape
    t = torch.zeros((b, h, w, 4), dtype=torch.float32)
    t[..., :3] = x_bchw.permute(0, 2, 3, 1)
    return t.to(dtype).cuda()
