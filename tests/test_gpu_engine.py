"""Engine-shaped evaluator seam (hm_engine_*, include/hivemind_amd.h) against the rules of class Engine
(nn/engine.h:43-81, nn/engine.cc:537-679): batch size, one request in flight per worker (second enqueue / sync without a
pending request fail), four independent workers, pinned outputs valid until the worker's next enqueue, f32 wrapper,
host or device observations, plan-file round trip; results equal the direct fused forward bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class HalfOutputs(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_uint16)) for n in ("value", "policy_a", "policy_b", "wdl", "moves_left")]


def _outs(o, B):
    g = lambda p, n: np.ctypeslib.as_array(p, shape=(n,)).copy()
    return g(o.value, B), g(o.policy_a, B * 4672), g(o.policy_b, B * 4672), g(o.wdl, B * 3), g(o.moves_left, B)


def test_engine_seam_rules(hm, tmp_path):
    import oracle_py as O
    from hivemind_amd import net as N
    lib = hm.lib
    torch.manual_seed(0)
    fused = N.FusedNet(N.rise_v3_small())
    B = 8
    e = C.c_void_p()
    hm.check(lib.hm_engine_create(0, B, C.byref(e)))
    assert lib.hm_engine_batch_size(e) == B                                   # getBatchSize
    obs = [hm.board_to_planes(hm.to_device(O.random_positions(40 + k, B, 80)), "f16") for k in range(5)]
    host = [o.cpu().numpy().view(np.uint16).copy() for o in obs]
    want = [tuple(t.cpu().numpy().view(np.uint16).reshape(-1) for t in fused(o)) for o in obs]
    out = HalfOutputs()
    # no network yet
    assert lib.hm_engine_enqueue_half(e, host[0].ctypes.data, 0) == -5 and b"no network" in lib.hm_last_error()
    path = str(tmp_path / "small.hmnp")
    fused.save(path)
    hm.check(lib.hm_engine_load_network_file(e, path.encode()))                # loadNetwork (plan-file branch)
    assert lib.hm_engine_load_network_file(e, (path + ".missing").encode()) == -1
    # sync without a pending request; bad worker index
    assert lib.hm_engine_sync_half(e, C.byref(out), 0) == -5 and b"no inference pending" in lib.hm_last_error()
    assert lib.hm_engine_enqueue_half(e, host[0].ctypes.data, 4) == -1
    # four workers in flight at once, host and device observations
    for w in range(4):
        src = host[w].ctypes.data if w % 2 == 0 else obs[w].data_ptr()
        hm.check(lib.hm_engine_enqueue_half(e, src, w))
    assert lib.hm_engine_enqueue_half(e, host[4].ctypes.data, 2) == -5 and b"already has an inference pending" in lib.hm_last_error()
    for w in (3, 1, 0, 2):                                                     # any order
        hm.check(lib.hm_engine_sync_half(e, C.byref(out), w))
        for got, exp in zip(_outs(out, B), want[w]):
            assert np.array_equal(got, exp), w
    assert lib.hm_engine_sync_half(e, C.byref(out), 1) == -5                   # nothing pending any more
    # outputs stay valid until the next enqueue on THAT worker
    hm.check(lib.hm_engine_run_half(e, host[4].ctypes.data, C.byref(out), 1))
    keep = HalfOutputs()
    hm.check(lib.hm_engine_run_half(e, host[0].ctypes.data, C.byref(keep), 2))
    for got, exp in zip(_outs(out, B), want[4]):
        assert np.array_equal(got, exp)
    # runInference: f32 in / f32 out through the same fp16 path (engine.cc:537-564)
    f32 = obs[3].float().cpu().numpy()
    v = np.zeros(B, np.float32); a = np.zeros(B * 4672, np.float32); b = np.zeros(B * 4672, np.float32)
    w_ = np.zeros(B * 3, np.float32); m = np.zeros(B, np.float32)
    hm.check(lib.hm_engine_run_f32(e, f32.ctypes.data, v.ctypes.data, a.ctypes.data, b.ctypes.data, w_.ctypes.data, m.ctypes.data, 3))
    for got, exp in zip((v, a, b, w_, m), want[3]):
        assert np.array_equal(got, exp.view(np.float16).astype(np.float32))
    assert lib.hm_engine_run_f32(e, None, v.ctypes.data, a.ctypes.data, b.ctypes.data, w_.ctypes.data, m.ctypes.data, 0) == -1
    # the loaded network serves the self-play driver's device-side seam too
    assert lib.hm_engine_net(e)
    hm.check(lib.hm_engine_destroy(e))


def test_checkpoint_to_fused_forward(hm, tmp_path):
    """(f2) a reference-shaped checkpoint drives the fused forward: identical outputs before / after the file round trip."""
    import oracle_py as O
    from hivemind_amd import net as N
    torch.manual_seed(5)
    model = N.rise_v33().eval()
    planes = hm.board_to_planes(hm.to_device(O.random_positions(9, 16, 60)), "f16")
    before = [t.clone() for t in N.FusedNet(model)(planes)]
    torch.save({"model_state_dict": model.state_dict(), "optimizer_state_dict": {}}, tmp_path / "model.tar")
    after = N.FusedNet(N.load_checkpoint(str(tmp_path / "model.tar")))(planes)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(before, after))
