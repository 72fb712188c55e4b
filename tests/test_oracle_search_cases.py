"""CPU suite: the search restatement (oracle/search.hpp) against the reference's own gtest known answers
for Node / SearchThread / classify_terminal_position / policy normalisation / hash_key properties
(engine/tests/test_move_gen.cc:280-326, 446-1296, 1595-1690, 1779-1791), transcribed as data in
tests/golden/search_cases.json by tests/golden/make_search_cases.py.  This is what pins the search oracle:
the reference's search sources cannot be built in this image."""
import pytest

import search_lab as SL

CASES = SL.load_cases()


@pytest.mark.parametrize("exp_mode,tie_mode", [(0, 0), (1, 1)], ids=["libm+std::sort", "portable_exp+total_order"])
@pytest.mark.parametrize("case", CASES, ids=[c["test"] for c in CASES])
def test_reference_search_known_answer(case, exp_mode, tie_mode):
    """Both configurations of the oracle must satisfy every reference assertion: (0,0) = the reference's own
    std::exp / std::sort behaviour, (1,1) = the implementation-independent forms the GPU engine uses."""
    r = SL.Runner(tie_mode, exp_mode)
    try:
        r.run(case["steps"])
    finally:
        r.close()


def test_every_reference_search_test_is_accounted_for():
    import json
    d = json.load(open(SL.GOLDEN))
    names = {c["test"] for c in d["cases"]}
    assert len(names) == len(d["cases"]) >= 45
    assert all(c["ref"].startswith("test_move_gen.cc:") for c in d["cases"])
    assert d["skipped"]          # the not-restated TESTs carry a reason each
