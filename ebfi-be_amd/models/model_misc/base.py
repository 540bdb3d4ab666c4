from ebfi_amd.model import BaseModel  # noqa: F401
