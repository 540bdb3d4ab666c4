"""Drop-in for `models.DCNv2.dcn_v2` (conv part; PS-ROI pooling is out of scope)."""
from ebfi_amd.dcn import DCN, DCN_sep, DCNv2, _DCNv2, dcn_v2_conv  # noqa: F401
