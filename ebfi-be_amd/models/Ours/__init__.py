"""Import-path shim: keeps the reference module layout importable on top of ebfi_amd."""
