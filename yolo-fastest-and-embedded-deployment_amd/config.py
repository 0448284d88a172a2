"""`config_params` of the reference (src/model_training/_config.py): `io_params` (:2-20: anchors, shapes, thresholds, class names;
save_path / log_path relative instead of the author's home directory) and `train_params` (:38-50), same names and meaning.
`augment_params` belongs to the dataset code, which is not part of this package."""
import copy

config_params = {
    "io_params": {
        "save_path": "./models/",           # where train() writes YOLO-Fastest_epoch_N.pth
        "log_path": "./logs/",
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
    "train_params": {
        "pretrained_pth": "",               # a .pth to start from; otherwise initialize_weights()
        "total_epochs": 30,
        "batch_size": 16,
        "lr0": 0.001,                       # initial learning rate (Adam)
        "momentum": 0.937,                  # unused by the reference's Adam call (train.py:84 passes betas explicitly)
        "weight_decay": 0.0005,             # unused by the reference (no weight_decay argument at train.py:84)
        "branch_weight": [1.0, 1.0],
        "IOU_loss_thre": 0.5,               # anchors whose shape IoU with a target exceeds this are not punished as background
        "IOU_val_thre": 0.5,                # match threshold of the mAP computation
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
