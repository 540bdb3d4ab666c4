"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/ebfi_hip.h declares, the ctypes table covers all of them, and the product ops fail loudly
(never fall back) when handed CPU tensors.  No compute calls: there is no GPU here."""
import ctypes
import os

import pytest
import torch

from ebfi_amd import _native as N


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(N.LIB_PATH):
        N.build()
    return N.LIB_PATH


def test_header_symbols_all_bound():
    declared = N.declared_symbols()
    assert len(declared) >= 16
    assert set(declared) == set(N.SIGNATURES), (set(declared) ^ set(N.SIGNATURES))


def test_library_exports_every_declared_symbol(built_lib):
    h = ctypes.CDLL(built_lib)
    for name in N.declared_symbols():
        assert hasattr(h, name), name


def test_library_loads_and_reports_version(built_lib):
    lib = N.lib()
    # the binding, the header and the library agree on the ABI generation (14 since the two-part image writer of round 6)
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ebfi_hip.h")).read()
    assert "#define EBFI_ABI_VERSION 14" in header
    assert lib.ebfi_abi_version() == 14 == N.ABI_VERSION
    assert lib.ebfi_events_workspace(16) == (2 * 16 + 1) * 8
    # workspace query is pure host arithmetic: 512 slabs max, here 2*ceil(16/64)=2 tiles -> 2 slabs
    need = lib.ebfi_dcn_backward_workspace(2, 2, 4, 4, 2, 3, 3, 1, 1, 1, 1, 1, 1, 1, 0)
    assert need == 2 * (2 * 2 * 9 + 2) * 4


def test_argument_errors_do_not_touch_the_gpu(built_lib):
    lib = N.lib()
    bad = N.i64x4((1, 1, 4, 4))
    rc = lib.ebfi_fac_forward(None, bad, bad, None, bad, bad, 3, None, bad, bad, 0, None)
    assert rc == -1 and b"null" in lib.ebfi_last_error()
    rc = lib.ebfi_dcn_forward(*([ctypes.c_void_p(8)] * 6), 1, 3, 4, 4, 2, 3, 3, 1, 1, 1, 1, 1, 1, 2, 0, None)
    assert rc == -1 and b"divisible" in lib.ebfi_last_error()
    rc = lib.ebfi_dcn_forward(*([ctypes.c_void_p(8)] * 6), 1, 2, 4, 4, 2, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, None)
    assert rc == -3    # a dtype code other than EBFI_F32 -> loud, not silent


def test_ops_refuse_cpu_tensors():
    from ebfi_amd.dcn import dcn_v2_conv
    from ebfi_amd.encodings import events_to_stack
    from ebfi_amd.fac import KernelConv2D
    from ebfi_amd.blur import Frame2Lap
    with pytest.raises(NotImplementedError):
        KernelConv2D(3)(torch.randn(1, 2, 4, 4), torch.randn(1, 18, 4, 4))
    with pytest.raises(NotImplementedError):
        dcn_v2_conv(torch.randn(1, 2, 4, 4), torch.zeros(1, 18, 4, 4), torch.ones(1, 9, 4, 4),
                    torch.randn(2, 2, 3, 3), torch.zeros(2), 1, 1, 1, 1)
    with pytest.raises(NotImplementedError):
        events_to_stack(torch.zeros(5), torch.zeros(5), torch.zeros(5), torch.zeros(5), 4, (4, 4))
    with pytest.raises(NotImplementedError):
        Frame2Lap(torch.rand(1, 3, 4, 4))


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(N.EbfiNativeError):
        N.lib()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ebfi-be_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".sh")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, os.path.join(root, f)
                assert "liboracle" not in text, os.path.join(root, f)
