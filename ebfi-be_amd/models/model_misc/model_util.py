"""Drop-in for the two helpers of `models.model_misc.model_util` the model uses."""
from ebfi_amd.model import CropSize, initialize_weights  # noqa: F401
