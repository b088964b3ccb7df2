"""Drop-in name shim: ``import segmentation_models_pytorch as smp`` resolves to the MI355X-native U-Net.

The reference imports smp at /root/reference/src/flair/model.py:3 and src/zone_detect/model.py and calls
``smp.create_model(arch=..., encoder_name=..., classes=..., in_channels=...)`` (model.py:37-41).  Putting
``flair-1_amd/`` on PYTHONPATH makes those files run unchanged on the HIP path.
"""
from flair_amd.unet import Unet, create_model  # noqa: F401

__version__ = "0.3.3+flair_amd"
