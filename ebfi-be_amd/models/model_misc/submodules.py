"""Drop-in for the one class of `models.model_misc.submodules` the model uses."""
from ebfi_amd.model import ConvLayer  # noqa: F401
