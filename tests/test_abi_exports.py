"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every symbol that
include/p4s_gpu.h declares, and its POD structs have the documented layout.  No compute calls."""
import ctypes as C
import os
import re

from parquet4seastar_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "p4s_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(p4s_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(capi.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), "libp4s_gpu.so lacks " + name
    lib.p4s_version.restype = C.c_char_p
    assert b"gfx950" in lib.p4s_version()


def test_struct_layout():
    assert C.sizeof(capi.PageDesc) == 48 and C.sizeof(capi.PageStatus) == 48
    assert capi.PageDesc.num_values.offset == 32 and capi.PageDesc.flags.offset == 44
    assert capi.PageStatus.values_decoded.offset == 8 and capi.PageStatus.out_bytes.offset == 40


def test_error_text_matches_reference_messages():
    from oracle import oracle as orc
    for code, tmpl in orc.MESSAGES.items():
        st = capi.PageStatus(code, 0, 0, 7, 9, 0, 0)
        buf = C.create_string_buffer(512)
        capi.lib().p4s_format_error(C.byref(st), buf, 512)
        assert buf.value.decode() == tmpl.format(7, 9), code


def test_no_gpu_means_loud_failure():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.P4SError):
        capi.Context(0)


def test_host_library_exports():
    """The C++ host layer's flat C library loads without a GPU and exports what its binding names."""
    from parquet4seastar_amd import hostapi
    lib = C.CDLL(hostapi.LIB_PATH)
    for name in hostapi.EXPORTS:
        assert hasattr(lib, name), "libp4s_host.so lacks " + name
    assert os.path.exists(os.path.join(os.path.dirname(hostapi.LIB_PATH), "host_selftest"))
