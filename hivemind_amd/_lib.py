"""ctypes binding of libhivemind_amd.so (the C ABI declared in include/hivemind_amd.h).

There is no CPU fallback: if the HIP extension is missing, importing this module raises, and
every call that needs a GPU raises HivemindError when no device is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HIVEMIND_AMD_LIB selects another build of the SAME HIP library (e.g. the -DHM_SEARCH_PROF diagnostic build)
LIB_PATH = os.environ.get("HIVEMIND_AMD_LIB") or os.path.join(_HERE, "csrc", "libhivemind_amd.so")

POS_DTYPE = np.dtype([
    ("by_type", "<u8", (6,)), ("by_color", "<u8", (2,)), ("promoted", "<u8"), ("key", "<u8"),
    ("hand", "u1", (2, 5)), ("castling", "u1"), ("ep", "u1"), ("stm", "u1"), ("rule50", "u1"),
    ("game_ply", "<u2"),
])
BOARD_DTYPE = np.dtype([
    ("pos", POS_DTYPE, (2,)), ("last_move", "<u4", (2,)), ("rep_count", "u1", (2,)),
    ("team", "u1"), ("time_adv", "u1"), ("reserved", "<u4"),
])
assert POS_DTYPE.itemsize == 96 and BOARD_DTYPE.itemsize == 208

NB_PLANES, PLANE_VALUES, POLICY_VALUES, MAX_MOVES = 74, 4736, 4672, 512
DT_F16, DT_F32, DT_U8 = 0, 1, 2


class HivemindError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")

lib = C.CDLL(LIB_PATH)

_vp, _sz, _i, _u64p = C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)
_SIGS = {
    "hm_init": (_i, [_i]),
    "hm_last_error": (C.c_char_p, []),
    "hm_device_available": (_i, []),
    "hm_abi_version": (_i, []),
    "hm_board_startpos": (_i, [_vp]),
    "hm_policy_index": (_i, [C.c_uint32, _i]),
    "hm_encode_planes": (_i, [_vp, _sz, _i, _vp, _vp]),
    "hm_legal_moves": (_i, [_vp, _sz, _vp, _vp, _vp]),
    "hm_legal_moves_wave": (_i, [_vp, _sz, _vp, _vp, _vp]),
    "hm_count_moves": (_i, [_vp, _sz, _vp, _vp]),
    "hm_make_moves": (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "hm_perft": (_i, [_vp, _i, _i, _i, _u64p, C.POINTER(C.c_double)]),
    "hm_hash_evaluator": (_i, [_vp, _i, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hm_net_create": (_i, [_vp, _sz, _vp, _vp, C.POINTER(_vp)]),
    "hm_net_create_host": (_i, [_vp, _sz, _vp, _sz, _vp, _sz, C.POINTER(_vp)]),
    "hm_net_destroy": (_i, [_vp]),
    "hm_net_forward": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hm_net_forward_groups": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hm_net_forward_groups_timed": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hm_net_profile": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hm_net_save_file": (_i, [C.c_char_p, _vp, _sz, _vp, _sz, _vp, _sz]),
    "hm_engine_create": (_i, [_i, _i, C.POINTER(_vp)]),
    "hm_engine_destroy": (_i, [_vp]),
    "hm_engine_load_network": (_i, [_vp, _vp, _sz, _vp, _sz, _vp, _sz]),
    "hm_engine_load_network_file": (_i, [_vp, C.c_char_p]),
    "hm_engine_enqueue_half": (_i, [_vp, _vp, _sz]),
    "hm_engine_sync_half": (_i, [_vp, _vp, _sz]),
    "hm_engine_run_half": (_i, [_vp, _vp, _vp, _sz]),
    "hm_engine_run_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "hm_engine_batch_size": (_i, [_vp]),
    "hm_engine_net": (_vp, [_vp]),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)      # AttributeError here = header/library mismatch
    _fn.restype, _fn.argtypes = _res, _args

EXPORTED_SYMBOLS = tuple(_SIGS)


def check(rc):
    if rc != 0:
        raise HivemindError(f"hivemind_amd error {rc}: {lib.hm_last_error().decode()}")
