"""MI355X-native YOLO-Fastest inference path (backbone -> 2 heads -> decode -> per-class NMS).

Python surface mirrors the reference (JunFenngZhi/YOLO-Fastest-and-Embedded-deployment):
`YoloFastest(io_params)`, `YOLO_post_process(...)`, `Detect_YOLO(...)`, `config_params`.
All compute is in libyolo_fastest_hip.so (hand-written HIP for gfx950, C ABI in include/yolo_fastest_hip.h);
there is no CPU fallback -- importing works without a GPU, computing does not.
"""
from . import _lib, packer
from .config import config_params, io_params_for
from .model import YoloFastest
from .post_process import YOLO_post_process
from .detect import Detect_YOLO, preprocess_u8
from . import validation
from .pipeline import BatchPipeline

__all__ = ["YoloFastest", "YOLO_post_process", "Detect_YOLO", "preprocess_u8", "config_params", "io_params_for",
           "packer", "BatchPipeline"]
