"""`config_params["io_params"]` of the reference (src/model_training/_config.py:1-20): the inference-side
keys only (anchors, shapes, thresholds, class names), with the same names and meaning."""
import copy

config_params = {
    "io_params": {
        # three groups of three [w, h] anchors; heads use anchors[0], anchors[1] (detect.py:51).
        # 256x320 uses the first two groups, 512x640 the last two (_config.py:9).
        "anchors": [
            [[10, 13], [16, 30], [33, 23]],
            [[150, 75], [100, 100], [75, 150]],
            [[300, 150], [200, 200], [150, 300]],
        ],
        "input_channel": 1,
        "input_shape": [256, 320, 1],       # net input [rows, cols, channels], rows/cols multiples of 32
        "origin_img_shape": [512, 640, 3],  # dataset image [rows, cols, channels]
        "input_tensor_shape": (1, 1, 256, 320),
        "num_cls": 3,
        "num_anchors": 3,
        "anchor_mask": [[0, 1, 2], [3, 4, 5]],
        "strides": [16, 32],
        "conf_thre": 0.5,
        "nms_thre": 0.2,
        "class_names": ["carrier", "defender", "destroyer"],
    },
}


def io_params_for(rows):
    """io_params for the two shipped checkpoints: rows=256 (256x320) or rows=512 (512x640, anchor groups 1..2,
    no resize)."""
    io = copy.deepcopy(config_params["io_params"])
    if rows == 512:
        io["input_shape"] = [512, 640, 1]
        io["input_tensor_shape"] = (1, 1, 512, 640)
        io["anchors"] = io["anchors"][1:]
    elif rows != 256:
        raise ValueError("shipped checkpoints are 256x320 and 512x640")
    return io
