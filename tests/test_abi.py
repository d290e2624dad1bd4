"""The C-ABI library loads without a GPU and exports every symbol include/soapdenovo2_amd.h declares."""
import ctypes
import os
import re

from conftest import ROOT
from soapdenovo2_amd import api


def _declared():
    src = open(os.path.join(ROOT, "include", "soapdenovo2_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:pg_|call_)[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    L = ctypes.CDLL(api.LIB_PATH)
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), n
    assert sorted(api.EXPORTED_SYMBOLS) == names


def test_device_ops_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    L = api.lib()
    h = L.pg_create(0, 31, 0, 8, 20)
    assert not h
    assert b"HIP device" in L.pg_last_error()


def test_pack_read_layout():
    import numpy as np
    L = api.lib()
    codes = np.array([3, 2, 1, 0] * 10, dtype=np.uint8)      # 40 bases -> 2 words
    out = np.zeros(2, dtype=np.uint64)
    L.pg_pack_read(codes.ctypes.data, 40, out.ctypes.data)
    want0 = 0
    for i in range(32):
        want0 |= int(codes[i]) << (62 - 2 * i)
    want1 = 0
    for i in range(32, 40):
        want1 |= int(codes[i]) << (62 - 2 * (i - 32))
    assert int(out[0]) == want0 and int(out[1]) == want1
    assert (api.pack_reads_uniform(codes[None, :])[:2] == out).all()
