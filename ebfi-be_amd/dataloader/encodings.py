"""Drop-in for the hot-path function of `dataloader.encodings`."""
from ebfi_amd.encodings import events_to_stack  # noqa: F401
