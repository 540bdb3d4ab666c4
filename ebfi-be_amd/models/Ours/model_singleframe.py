"""Drop-in for the reference's `models.Ours.model_singleframe` (the names train_ours.py:20 and
infer_ours.py import / eval()): re-exports the MI355X-native implementation."""
from ebfi_amd.model import (EVFIAutoEx, ExposureDecision, Modification, ResidualControl,  # noqa: F401
                            UNet3d_18)
from ebfi_amd.dcn import DCN_sep  # noqa: F401  (imported, never instantiated, by the reference model file)
from ebfi_amd.fac import KernelConv2D  # noqa: F401
