"""The C-ABI library loads and exports every symbol include/hivemind_amd.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hivemind_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    path = os.path.join(ROOT, "hivemind_amd", "csrc", "libhivemind_amd.so")
    assert os.path.exists(path), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(path)
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in hivemind_amd.h but not exported"


def test_binding_covers_header():
    import hivemind_amd._lib as L
    assert set(declared_symbols()) <= set(L.EXPORTED_SYMBOLS) | {"hm_last_error"}
    assert L.lib.hm_abi_version() >= 1


def test_host_side_helpers_without_gpu():
    import numpy as np
    import hivemind_amd as hm
    import oracle_py as O
    sp = hm.startpos()
    assert sp.tobytes() == O.Board().compact(0, False).tobytes()      # incl. the Zobrist key
    normal, drop = O.policy_tables("ora")
    for c in (0, 1):
        for f, t in ((12, 28), (6, 21), (52, 60), (4, 7)):
            assert hm.policy_index((f << 6) | t, c) == normal[c, f, t, 0]
    assert hm.policy_index(0, 0) == 0


def test_calls_fail_loudly_without_device():
    import torch
    import pytest
    import hivemind_amd as hm
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hm.HivemindError):
        hm.init(0)
    assert hm.lib.hm_init(0) != 0 and b"no HIP device" in hm.lib.hm_last_error()
