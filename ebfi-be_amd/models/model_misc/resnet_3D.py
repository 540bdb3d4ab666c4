"""Drop-in for the parts of `models.model_misc.resnet_3D` the detail branch uses."""
from ebfi_amd.model import BasicBlock, Conv_3d, SEGating, VideoResNet, identity, r3d_18, upConv3D  # noqa: F401
