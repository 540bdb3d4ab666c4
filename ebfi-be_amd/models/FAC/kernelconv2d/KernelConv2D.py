"""Drop-in for `models.FAC.kernelconv2d.KernelConv2D`."""
from ebfi_amd.fac import KernelConv2D, KernelConv2DFunction  # noqa: F401
