"""Host-only parts of the UCI front end (no GPU): the reference's known answer for the early-stopping rule
(engine/tests/test_move_gen.cc:1298-1302, SearchParams::has_insurmountable_visit_lead) and the movetime controller of
Agent::run_search's polling loop (agent.cc:561-713; SearchInfo::update_nps / try_extend_time, searchinfo.h:129-180):
early exit on a solved root or a forced mate, early stopping, time extension on a falling evaluation and on a late change of
the best move, at most two extensions."""
import ctypes as C

import numpy as np

import hivemind_amd as hm

lib = hm.lib


def test_reference_known_answer_visit_lead():
    assert lib.hm_insurmountable_visit_lead(100.0, 40.0, 2.0) == 1
    assert lib.hm_insurmountable_visit_lead(100.0, 60.0, 2.0) == 0
    assert lib.hm_insurmountable_visit_lead(100.0, 50.0, 2.0) == 0


class TM:
    def __init__(self, ms):
        self.h = lib.hm_time_manager_create(ms)

    def poll(self, elapsed, nodes, visits, q, root_type=0, child_type=None, child_end=None):
        v = np.ascontiguousarray(visits, np.int32); qq = np.ascontiguousarray(q, np.float32)
        ct = np.ascontiguousarray(child_type if child_type is not None else np.zeros(len(v)), np.int32)
        ce = np.ascontiguousarray(child_end if child_end is not None else np.zeros(len(v)), np.int32)
        eff = C.c_double(0)
        log = C.create_string_buffer(512)
        stop = lib.hm_time_manager_poll(self.h, elapsed, nodes, len(v), v.ctypes.data, qq.ctypes.data, root_type, ct.ctypes.data, ce.ctypes.data,
                                        C.byref(eff), log, 512)
        return bool(stop), eff.value, log.value.decode()

    def __del__(self):
        lib.hm_time_manager_destroy(self.h)


def test_runs_to_the_deadline_without_a_reason_to_stop():
    t = TM(1000)
    for ms in (50, 300, 700, 990):
        stop, eff, log = t.poll(ms, ms * 10, [60, 50, 40], [0.1, 0.1, 0.0])
        assert not stop and eff == 1000 and log == ""
    assert t.poll(1000.5, 10000, [600, 500, 400], [0.1, 0.1, 0.0])[0]


def test_early_stopping_needs_nps_visit_lead_and_the_better_q():
    t = TM(1000)
    # before 100 ms there is no NPS estimate (searchinfo.h:132): no early stop however large the lead
    assert not t.poll(50, 500, [400, 10], [0.3, 0.0])[0]
    # 10 000 nodes/s, 800 ms left: the runner-up can still reach 50 + 8000 visits
    assert not t.poll(200, 2000, [1500, 50], [0.3, 0.0])[0]
    # 100 ms left: projected 50 + 0.1 s x nps (~10 000) = ~1050; best has 9000 > 2 x 1050 and the better Q: stop
    stop, _, log = t.poll(900, 9000, [9000, 50], [0.3, 0.0])
    assert stop and log == "info string Early stopping: saved 100ms\n"
    # same lead but the runner-up has the better Q: go on
    t2 = TM(1000)
    t2.poll(200, 2000, [1500, 50], [0.0, 0.3])
    assert not t2.poll(900, 9000, [9000, 50], [0.0, 0.3])[0]


def test_time_extension_on_falling_eval_and_late_best_move_change():
    t = TM(1000)
    t.poll(100.5, 1000, [60, 50], [0.20, 0.1])
    stop, eff, log = t.poll(300, 3000, [600, 590], [0.10, 0.1])               # best Q fell by 0.10 > 0.05
    assert not stop and eff == 1000 + int(700 * 0.5) and log == "info string Extending search time (eval dropped by 10 cp)\n"
    stop, eff2, log = t.poll(700, 7000, [2000, 2100], [0.10, 0.1])            # the best move changes after 40 % of the move time
    assert not stop and eff2 == eff + int((eff - 700) * 0.5) and log == "info string Extending search time (best move changed to 1)\n"
    stop, eff3, log = t.poll(800, 8000, [2500, 2400], [0.02, 0.1])            # third trigger: MAX_TIME_EXTENSIONS = 2
    assert eff3 == eff2 and log == ""


def test_early_exit_on_solved_root_and_forced_mate():
    t = TM(5000)
    stop, _, log = t.poll(20, 200, [10, 5], [1.0, 0.0], root_type=1)
    assert stop and log == "info string Early exit: root position is proven WIN\n"
    t = TM(5000)
    stop, _, log = t.poll(20, 200, [10, 5], [1.0, 0.0], child_type=[2, 0], child_end=[2, 0])     # the most visited child is a proven loss for the opponent
    assert stop and log == "info string Early exit: forced mate in 1 found\n"
    t = TM(5000)
    assert not t.poll(20, 200, [10, 5], [1.0, 0.0], child_type=[0, 2], child_end=[0, 2])[0]      # only the best child counts (agent.cc:114-127)
