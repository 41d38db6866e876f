"""CPU: the C-ABI library loads and exports every symbol include/openeat_hip.h
declares (no compute is launched here)."""
import ctypes
import os
import re

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "openeat_hip.h")
LIB = os.path.join(ROOT, "openeat_amd", "lib", "libopeneat_hip.so")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(oe_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(LIB)
    names = declared_functions()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/openeat_hip.h but not exported"


def test_python_binding_covers_the_header():
    from openeat_amd import hip
    assert sorted(hip.exported_symbols()) == declared_functions()
    lib = hip.lib()
    assert lib.oe_abi_version() >= 1


def test_invalid_arguments_are_reported_not_launched():
    from openeat_amd import hip
    lib = hip.lib()
    rc = lib.oe_layernorm_fwd(None, None, None, 1e-5, 4, 32, None, 0, None, None, None)
    assert rc != 0 and b"null" in lib.oe_last_error()
