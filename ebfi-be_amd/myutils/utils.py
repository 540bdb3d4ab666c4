"""Drop-in for the hot-path helpers of `myutils.utils`."""
from ebfi_amd.blur import Frame2DCP, Frame2Lap  # noqa: F401
from ebfi_amd.dp import reduce_tensor  # noqa: F401
