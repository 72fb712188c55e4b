"""CPU suite for the search restatement (oracle/search.hpp): the reference's own known answers that do
not need the unbuildable search sources — joint-action sit rules and generator cases
(engine/tests/test_move_gen.cc:159-278), progressive-widening schedule and cpuct
(search/search_params.h:307-317 through the reference build, tests/golden/pw_schedule.json) — plus
self-consistency of the single-thread schedule."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_py as O

L = O.lib
L.ora_gen_enumerate.restype = C.c_int
L.ora_gen_enumerate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
L.ora_joint_prior.restype = C.c_float
L.ora_joint_prior.argtypes = [C.c_uint32, C.c_float, C.c_uint32, C.c_float] + [C.c_int] * 7
L.ora_is_double_sit_legal.restype = C.c_int
L.ora_is_single_pass_legal.restype = C.c_int
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def enumerate_gen(a, b, pa, pb, adv, a_on, b_on, ca=None, cb=None, tie=0):
    a = np.asarray(a, np.uint32); b = np.asarray(b, np.uint32)
    pa = np.asarray(pa, np.float32); pb = np.asarray(pb, np.float32)
    ca = None if ca is None else np.asarray(ca, np.uint8)
    cb = None if cb is None else np.asarray(cb, np.uint8)
    oa = np.zeros(4096, np.uint32); ob = np.zeros(4096, np.uint32); op = np.zeros(4096, np.float32)
    n = L.ora_gen_enumerate(a.ctypes.data, len(a), b.ctypes.data, len(b), pa.ctypes.data, pb.ctypes.data, int(adv), int(a_on), int(b_on),
                            None if ca is None else ca.ctypes.data, None if cb is None else cb.ctypes.data, tie,
                            oa.ctypes.data, ob.ctypes.data, op.ctypes.data, 4096)
    return list(zip(oa[:n].tolist(), ob[:n].tolist())), op[:n]


def test_double_sit_rule():                       # test_move_gen.cc:159-182
    assert L.ora_is_double_sit_legal(1, 1, 0) and L.ora_is_double_sit_legal(1, 0, 1)
    assert not L.ora_is_double_sit_legal(0, 1, 0) and not L.ora_is_double_sit_legal(1, 1, 1) and not L.ora_is_double_sit_legal(1, 0, 0)
    assert L.ora_joint_prior(0, 0.5, 0, 0.5, 1, 0, 0, 0, 0, 0, 0) < 0            # disadvantaged
    assert L.ora_joint_prior(0, 0.5, 0, 0.5, 1, 0, 1, 0, 0, 0, 0) == np.float32(0.25)   # advantaged, one idle board
    assert L.ora_joint_prior(0, 0.5, 0, 0.5, 1, 1, 1, 0, 0, 0, 0) < 0            # both boards on turn


def test_single_pass_rule():                      # test_move_gen.cc:184-211
    assert not L.ora_is_single_pass_legal(0, 1, 1, 0) and L.ora_is_single_pass_legal(0, 1, 1, 1)
    assert L.ora_is_single_pass_legal(1, 1, 1, 0) and L.ora_is_single_pass_legal(0, 1, 0, 0)
    assert L.ora_joint_prior(0, 0.5, 7, 0.5, 1, 1, 0, 1, 1, 0, 0) < 0            # quiet partner
    assert L.ora_joint_prior(0, 0.5, 7, 0.5, 1, 1, 0, 1, 1, 0, 1) == np.float32(0.25)   # capturing partner
    assert L.ora_joint_prior(0, 0.5, 7, 0.5, 1, 1, 0, 0, 1, 0, 0) == np.float32(0.25)   # forced pass


@pytest.mark.parametrize("tie", [0, 1])
def test_generator_skips_quiet_pass_pairs(tie):   # test_move_gen.cc:213-242
    quiet_a, cap_b, quiet_b = 1, 2, 3
    pairs, _ = enumerate_gen([quiet_a, 0], [quiet_b, cap_b, 0], [0.6, 0.4], [0.5, 0.3, 0.2], False, True, True, [0, 0], [0, 1, 0], tie)
    assert (0, cap_b) in pairs
    assert (0, quiet_b) not in pairs and (quiet_a, 0) not in pairs and (0, 0) not in pairs


@pytest.mark.parametrize("tie", [0, 1])
def test_generator_follows_prior_ordering(tie):   # test_move_gen.cc:244-262, 264-278
    pairs, pr = enumerate_gen([1, 2], [3, 4], [0.9, 0.01], [0.8, 0.2], False, True, True, tie=tie)
    assert pairs[0] == (1, 3) and np.isclose(pr[0], 0.72) and pairs[1] == (1, 4) and np.isclose(pr[1], 0.18)
    pairs, _ = enumerate_gen([1, 2, 3], [0], [0.9, 0.02, 0.01], [1.0], False, True, False, tie=tie)
    assert [p[0] for p in pairs] == [1, 2, 3]


def test_generator_is_exhaustive_and_sorted():
    rng = np.random.RandomState(5)
    for _ in range(50):
        na, nb = rng.randint(1, 12), rng.randint(1, 12)
        pa = rng.dirichlet(np.ones(na)).astype(np.float32); pb = rng.dirichlet(np.ones(nb)).astype(np.float32)
        a = list(range(1, na)) + [0]; b = list(range(101, 100 + nb)) + [0]
        adv, a_on, b_on = [bool(x) for x in rng.randint(0, 2, 3)]
        ca = rng.randint(0, 2, na).astype(np.uint8); cb = rng.randint(0, 2, nb).astype(np.uint8)
        for tie in (0, 1):
            pairs, pr = enumerate_gen(a, b, pa, pb, adv, a_on, b_on, ca, cb, tie)
            assert len(set(pairs)) == len(pairs)
            assert np.all(np.diff(pr) <= 1e-12)                   # descending joint prior
            # every valid pair is produced exactly once
            a_can = a_on and na > 1; b_can = b_on and nb > 1
            valid = 0
            for i, ma in enumerate(a):
                for j, mb in enumerate(b):
                    valid += L.ora_joint_prior(ma, float(pa[i]), mb, float(pb[j]), a_on, b_on, adv, a_can, b_can, int(ca[i]), int(cb[j])) >= 0
            assert valid == len(pairs)
        p0, _ = enumerate_gen(a, b, pa, pb, adv, a_on, b_on, ca, cb, 0)
        p1, _ = enumerate_gen(a, b, pa, pb, adv, a_on, b_on, ca, cb, 1)
        assert p0 == p1                                           # distinct priors: both tie modes agree


def test_pw_schedule_and_cpuct_match_reference_build():
    pw = json.load(open(os.path.join(G, "pw_schedule.json")))
    for v in range(0, 2001):
        assert L.ora_pw_allowed_children(v, 0) == pw["nonroot"][v] and L.ora_pw_allowed_children(v, 1) == pw["root"][v]
    # reference test values (engine/tests/test_move_gen.cc:774-781): ceil(2 * N^0.4)
    assert [L.ora_pw_allowed_children(v, 0) for v in (0, 1, 2, 10, 100)] == [1, 2, 3, 6, 13]
    if O.ref is not None:
        for v in (0.0, 1.0, 7.0, 400.0, 1600.0, 20000.0):
            assert L.ora_get_cpuct(v) == O.ref.ref_get_cpuct(v)


def test_portable_exp_close_to_libm():
    xs = np.concatenate([np.linspace(-86.9, 0, 20001), -np.logspace(-6, 1.9, 2000)]).astype(np.float32)
    got = np.array([L.ora_portable_expf(float(x)) for x in xs], np.float32)
    want = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(got - want) / want) < 3e-7


def test_search_schedule_properties():
    """400-node budget from the start position: 412 nodes / 411 root-edge visits (the figure SURVEY §6
    measured on the reference with one search thread), deterministic, budget respected within one batch."""
    b = O.Board()
    runs = []
    for _ in range(2):
        s = O.Search(0, 0)
        assert s.run(b, 0, False, 400)
        e, i = s.edges(), s.info()
        runs.append((e["visits"].tolist(), e["move_a"].tolist(), i["nodes"]))
        assert i["nodes"] == 412 and int(e["visits"].sum()) == 411 and i["root_visits"] == 412
        assert 400 <= i["nodes"] < 400 + 16 and i["same_batch"] == 0
    assert runs[0] == runs[1]
    # terminal roots give "bestmove (none)"
    m = O.Board(fen="r1bqkb1r/pppp1Qpp/2n2n2/4p3/2B1P3/8/PPPP1PPP/RNB1K1NR b KQkq - 0 4 | 4k3/8/8/8/8/8/8/4K3 w - - 0 1")
    assert not O.Search(1, 1).run(m, 1, False, 100)


def test_search_finds_immediate_root_mate():
    # white to play Qxf7# on board A (scholar's mate), partner board idle
    b = O.Board(fen="r1bqkb1r/pppp1ppp/2n2n2/4p2Q/2B1P3/8/PPPP1PPP/RNB1K1NR w KQkq - 4 4 | 4k3/8/8/8/8/8/8/4K3 b - - 0 1")
    s = O.Search(1, 1)
    assert s.run(b, 0, False, 400)
    e, i = s.edges(), s.info()
    assert len(e["visits"]) == 1 and e["visits"][0] == 1 and b.uci(0, e["move_a"][0]) == "h5f7" and e["move_b"][0] == 0
    assert i["root_type"] == 1 and s.root_q() == 1.0          # NodeType::WIN
