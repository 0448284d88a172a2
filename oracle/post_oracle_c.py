"""ORACLE (test infrastructure): ctypes binding of oracle/libpost_oracle.so (post_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libpost_oracle.so")
        if not os.path.exists(path):
            build()
        _lib = ctypes.CDLL(path)
        _lib.yf_oracle_post.restype = ctypes.c_int
    return _lib


def post_process(head_large, head_small, anchors, input_shape, conf_thres=0.5, nms_thres=0.2, num_cls=3, kmax=None, num_anchors=3):
    """One frame. head_*: float32 [num_anchors*(5+num_cls),h,w]. Returns dict(box,conf,score,cls,src,count,n_candidates)."""
    hl = np.ascontiguousarray(head_large, np.float32)
    hs = np.ascontiguousarray(head_small, np.float32)
    anc = np.ascontiguousarray(np.asarray(anchors, np.float64)[:2, :num_anchors]).reshape(-1)
    if kmax is None:
        kmax = num_anchors * (hl.shape[1] * hl.shape[2] + hs.shape[1] * hs.shape[2])
    box = np.zeros((kmax, 4), np.int32); conf = np.zeros(kmax); score = np.zeros(kmax)
    cls = np.zeros(kmax, np.int32); src = np.zeros(kmax, np.int32); ncand = ctypes.c_int32(0)
    P = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
    n = lib().yf_oracle_post(P(hl, ctypes.c_float), hl.shape[1], hl.shape[2], P(hs, ctypes.c_float), hs.shape[1],
                             hs.shape[2], P(anc, ctypes.c_double), int(input_shape[0]), int(input_shape[1]),
                             ctypes.c_double(conf_thres), ctypes.c_double(nms_thres), num_cls, num_anchors, kmax,
                             P(box, ctypes.c_int32), P(conf, ctypes.c_double), P(score, ctypes.c_double),
                             P(cls, ctypes.c_int32), P(src, ctypes.c_int32), ctypes.byref(ncand))
    if n == -2:
        raise ZeroDivisionError("division by zero")  # detect.py:39
    if n < 0:
        raise RuntimeError("kmax too small")
    return dict(box=box[:n], conf=conf[:n], score=score[:n], cls=cls[:n], src=src[:n], count=n,
                n_candidates=ncand.value)
