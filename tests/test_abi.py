"""The C-ABI library loads and exports every symbol include/duckhts_amd.h declares (no compute calls)."""
import ctypes
import os
import re

import duckhts_amd
from conftest import ROOT


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dhts_[a-z0-9_]+|duckhts_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(duckhts_amd.LIB_PATH)
    names = declared_functions("duckhts_amd.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/duckhts_amd.h but not exported"
    assert sorted(duckhts_amd.EXPORTS) == names


def test_abi_version_and_no_device_fails_loudly():
    L = duckhts_amd.lib()
    assert L.dhts_abi_version() == 1
    if L.dhts_device_count() == 0:
        # no GPU here: creating a context must fail (no CPU fallback)
        assert not L.dhts_create(0)
        try:
            duckhts_amd.Context(0)
        except duckhts_amd.DhtsError as e:
            assert "no CPU fallback" in str(e)
        else:
            raise AssertionError("Context() must raise without a device")
        assert b"no context" in L.dhts_error(None)
