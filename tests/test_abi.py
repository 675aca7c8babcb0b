"""CPU: the C-ABI shared library loads, exports every symbol include/ccp_gs.h declares, and
fails loudly (no CPU fallback) when no HIP device is usable.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ccp_gs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ccp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from coursecomputationalphotography_amd import capi
    lib = capi.load()
    declared = header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"libccp_gs.so does not export {name}"
    assert sorted(capi.ABI_SYMBOLS) == declared, "capi.ABI_SYMBOLS out of sync with include/ccp_gs.h"
    assert lib.ccp_abi_version() == 6


def test_status_strings():
    from coursecomputationalphotography_amd import capi
    assert capi.status_string(0) == "ok"
    assert "fallback" in capi.status_string(2)


def test_no_device_fails_loudly():
    """Without a GPU every compute entry point refuses; nothing is computed on the host."""
    from coursecomputationalphotography_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(capi.CcpError) as e:
        capi.Grid(8, 8)
    assert e.value.status == 2
    with pytest.raises(capi.CcpError) as e:
        capi.CsrMatrix()
    assert e.value.status == 2


def test_bad_arguments_rejected_before_touching_the_device():
    from coursecomputationalphotography_amd import capi
    lib = capi.load()
    h = ctypes.c_void_p()
    assert lib.ccp_grid_create(None, ctypes.byref(h)) == 1
    d = capi.GridDesc(0, 8, 1, 0, 8, 0, 0, 0)
    assert lib.ccp_grid_create(ctypes.byref(d), ctypes.byref(h)) == 1
    d = capi.GridDesc(8, 8, 99, 0, 8, 0, 0, 0)
    assert lib.ccp_grid_create(ctypes.byref(d), ctypes.byref(h)) == 1
    assert lib.ccp_csr_create(0, None) == 1
    assert lib.ccp_grid_destroy(None) == 0 and lib.ccp_csr_destroy(None) == 0
    # communicator handles: argument errors come before any RCCL / device work
    buf = (ctypes.c_uint8 * capi.COMM_ID_BYTES)()
    assert lib.ccp_comm_create(buf, 0, 1, 0, None) == 1
    assert lib.ccp_comm_create(None, 0, 1, 0, ctypes.byref(h)) == 1
    assert lib.ccp_comm_create(buf, 3, 2, 0, ctypes.byref(h)) == 1
    assert lib.ccp_comm_destroy(None) == 0 and lib.ccp_comm_unique_id(None) == 1
    assert lib.ccp_grid_attach_comm(None, None) == 1
    assert "RCCL" in capi.status_string(7)


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: no product source may reference it."""
    pkg = os.path.join(ROOT, "coursecomputationalphotography_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "ccp_oracle" not in text and "orc_" not in text, f


def test_header_is_plain_c_and_cxx(tmp_path):
    """include/ccp_gs.h is the whole boundary: it must compile on its own as C11 and as C++17 (no torch, no HIP types)."""
    import shutil
    import subprocess
    hdr = os.path.join(ROOT, "include", "ccp_gs.h")
    for cc, std, lang in (("gcc", "-std=c11", "c"), ("g++", "-std=c++17", "c++")):
        if shutil.which(cc) is None:
            pytest.skip(f"no {cc}")
        src = tmp_path / f"use_header.{ 'c' if lang == 'c' else 'cc'}"
        src.write_text('#include "ccp_gs.h"\nint main(void) { return ccp_abi_version() == 0; }\n')
        subprocess.check_call([cc, std, "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.dirname(hdr), str(src)])
