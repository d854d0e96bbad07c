"""Configuration objects with the attribute names and values of the reference
(config.py:15-224, heart_main.py:26-174, LiTS_2017/LiTS_main.py:62-142).

Only the attributes the hot path reads are kept; names, values and the derived
IMAGE_SHAPE / MASK_SHAPE rules are the contract.  A reference
``HeartConfig`` instance can be passed anywhere a ``cfun_amd`` config is expected.
"""
import numpy as np


class Config:
    NAME = None
    GPU_COUNT = 1
    IMAGES_PER_GPU = 1
    BACKBONE = "P3D19"
    BACKBONE_LAYERS = (2, 3)            # P3D19 = [2, 3] bottlenecks (backbone.py:161-164)
    BACKBONE_STEM_KD = 3                # stem kernel (3,7,7), pad (1,3,3) (backbone.py:124)
    BACKBONE_STRIDES = [8, 16]
    BACKBONE_CHANNELS = [16, 32]
    FPN_CLASSIFY_FC_LAYERS_SIZE = 128
    UNET_MASK_BRANCH_CHANNEL = 20
    TOP_DOWN_PYRAMID_SIZE = 128
    RPN_CONV_CHANNELS = 256
    NUM_CLASSES = 8
    RPN_ANCHOR_SCALES = (64, 128)
    RPN_ANCHOR_RATIOS = [1]
    RPN_ANCHOR_STRIDE = 1
    RPN_NMS_THRESHOLD = 0.7
    RPN_TRAIN_ANCHORS_PER_IMAGE = 128
    PRE_NMS_LIMIT = 1000
    POST_NMS_ROIS_TRAINING = 500
    POST_NMS_ROIS_INFERENCE = 64
    IMAGE_RESIZE_MODE = "self"
    IMAGE_MIN_DIM = 192
    IMAGE_MAX_DIM = 320
    TRAIN_ROIS_PER_IMAGE = 15
    ROI_POSITIVE_RATIO = 0.33
    POOL_SIZE = [12, 12, 12]
    MASK_POOL_SIZE = [96, 96, 96]
    MAX_GT_INSTANCES = 32
    RPN_BBOX_STD_DEV = np.array([0.1, 0.1, 0.1, 0.2, 0.2, 0.2])
    BBOX_STD_DEV = np.array([0.1, 0.1, 0.1, 0.2, 0.2, 0.2])
    DETECTION_MAX_INSTANCES = 32
    DETECTION_MIN_CONFIDENCE = 0.7
    DETECTION_NMS_THRESHOLD = 0.3
    DETECTION_TARGET_IOU_THRESHOLD = 0.5
    LEARNING_RATE = 0.001
    LEARNING_MOMENTUM = 0.9
    WEIGHT_DECAY = 0.0001
    LOSS_WEIGHTS = {"rpn_class_loss": 100., "rpn_bbox_loss": 50., "mrcnn_class_loss": 1., "mrcnn_bbox_loss": 20.,
                    "mrcnn_mask_loss": 1., "mrcnn_mask_edge_loss": 1.}
    TRAIN_BN = False
    GRADIENT_CLIP_NORM = 5.0
    UNET_DROPOUT = 0.6                  # nn.Dropout3d(p=0.6), mask_branch.py:19 (0 in the LiTS fork)

    def __init__(self, stage="beginning"):
        self.BATCH_SIZE = self.IMAGES_PER_GPU * self.GPU_COUNT
        if self.IMAGE_RESIZE_MODE == "crop":
            dims = [self.IMAGE_MIN_DIM] * 3
        elif self.IMAGE_RESIZE_MODE == "self":   # [H, W, D] = [max, max, min]
            dims = [self.IMAGE_MAX_DIM, self.IMAGE_MAX_DIM, self.IMAGE_MIN_DIM]
        else:
            dims = [self.IMAGE_MAX_DIM] * 3
        self.IMAGE_SHAPE = np.array(dims + [1])
        self.STAGE = stage
        side = 192 if stage == "finetune" else 96
        self.MASK_SHAPE = self.MINI_MASK_SHAPE = (side, side, side)

    @property
    def image_dhw(self):
        h, w, d = [int(v) for v in self.IMAGE_SHAPE[:3]]
        return d, h, w


class HeartConfig(Config):
    NAME = "heart"


def heart_config(stage, height, width, depth):
    """HeartConfig for an explicit volume size (BASELINE.json configs: 64x64x32 ... 512x512x256)."""
    if height != width:
        raise ValueError("IMAGE_RESIZE_MODE 'self' makes H == W (config.py:208-209)")
    cls = type("HeartConfig%dx%dx%d" % (height, width, depth), (HeartConfig,),
               dict(IMAGE_MAX_DIM=height, IMAGE_MIN_DIM=depth))
    return cls(stage)


class LiTSConfig(Config):
    """Shapes of the LiTS_2017 fork (BASELINE.json configs[4])."""
    NAME = "LiTS"
    BACKBONE = "P3D35"
    BACKBONE_LAYERS = (4, 5)            # LiTS_2017/backbone.py:172-176
    BACKBONE_STEM_KD = 5                # stem k(5,7,7) p(2,3,3), LiTS_2017/backbone.py:124
    BACKBONE_CHANNELS = [24, 48]
    TOP_DOWN_PYRAMID_SIZE = 160
    RPN_CONV_CHANNELS = 320
    FPN_CLASSIFY_FC_LAYERS_SIZE = 320
    UNET_MASK_BRANCH_CHANNEL = 32
    NUM_CLASSES = 3
    IMAGE_MAX_DIM = 320
    IMAGE_MIN_DIM = 256
    MASK_POOL_SIZE = [32, 80, 80]
    UNET_DROPOUT = 0.0
    UNMOLD_OVERLAP_TILE = True          # utils.unmold_mask averages ALL detections (LiTS_2017/utils.py:383-408)
    # the fork trains in two phases (LiTS_2017/model.py:985-1001, 1518-1545, 1282-1296): 'beginning' = detector only
    # (FPN, RPN, classifier; no mask head, mask losses 0), any other stage = mask branch only (everything else frozen,
    # classifier not run, detection losses 0).  False: the heart pipeline (all heads, all six losses) at LiTS shapes.
    STAGE_SPLIT = True
    ROI_COUNT_ROUND = True              # RoI counts by int(round()) (LiTS_2017/model.py:448, 496; heart truncates)
    MASK_CE_CLASS_WEIGHTS = (1.0, 1.0, 100.0)   # nn.CrossEntropyLoss(weight=...) of the mask loss (LiTS_2017/model.py:926)
    EDGE_LOSS_RAW_SOBEL = True          # edge loss = MSE on the raw 3 Sobel responses, no magnitude (LiTS_2017/model.py:959-972)
    LOSS_WEIGHTS = {"rpn_class_loss": 50., "rpn_bbox_loss": 5., "mrcnn_class_loss": 50., "mrcnn_bbox_loss": 5.,
                    "mrcnn_mask_loss": 2., "mrcnn_mask_edge_loss": 0.25}     # LiTS_2017/LiTS_main.py:162-169
    PAD_IMAGE_SHAPE = [646, 646, 536]   # LiTS_2017/LiTS_main.py:121: every volume is centred in this zero frame, then resized
    POST_NMS_ROIS_INFERENCE = 50        # LiTS_2017/LiTS_main.py:107
    DETECTION_NMS_THRESHOLD = 0.7       # LiTS_2017/LiTS_main.py:147

    def __init__(self, stage="beginning"):
        super().__init__(stage)
        self.MASK_SHAPE = self.MINI_MASK_SHAPE = (64, 160, 160) if stage == "finetune" else (32, 80, 80)
        # LiTS_2017/config.py:216-226: 50 RoIs at 33 % positives for the detector phase, 4 positives for the mask phase
        self.TRAIN_ROIS_PER_IMAGE, self.ROI_POSITIVE_RATIO = (50, 0.33) if stage == "beginning" else (4, 1.0)
