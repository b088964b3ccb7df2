"""flair_amd — MI355X-native segmentation hot path of FLAIR-1 (U-Net/ResNet34 fwd+bwd, CE/argmax/IoU head).

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every FLOP of the path
runs in hand-written HIP kernels inside ``libflair_hip.so`` (C ABI: include/flair_hip.h).
"""
from . import _lib
from .unet import Unet, create_model
from .model import FLAIR_ModelFactory, MetadataMLP
from .segformer import SegformerForSemanticSegmentation
from .head import FusedCrossEntropyLoss, MulticlassJaccardIndex, MeanMetric
from .task_module import segmentation_task_training, segmentation_task_predict
from .train import SegTrainer, bucket_ranges, allreduce_buckets, shard_indices
from .data_feed import TileFeed, draw_d4
from . import checkpoint, metrics, tasks_utils, writer, zone_detect

__all__ = ["Unet", "create_model", "FLAIR_ModelFactory", "MetadataMLP", "FusedCrossEntropyLoss", "SegformerForSemanticSegmentation",
           "MulticlassJaccardIndex", "MeanMetric", "segmentation_task_training", "segmentation_task_predict",
           "SegTrainer", "bucket_ranges", "allreduce_buckets", "shard_indices", "TileFeed", "draw_d4", "checkpoint", "metrics", "tasks_utils",
           "writer", "zone_detect"]
