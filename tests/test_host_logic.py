"""CPU suite for host logic that needs no GPU: HVM4 chunk writer/reader (record format of
tools/selfplay.cc:69-158 as read by src/preprocessing/convert_selfplay_data.py:24-28), the record gather
with world_size-2 gloo, config defaults and argument validation of the C ABI."""
import ctypes as C
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sample_bytes(game_id, nodes, mply, team, planes_byte, pol_a, pol_b):
    b = struct.pack("<QIHHBBbBf", game_id, nodes, mply, 3, team, 0, 1, 2, 0.25)
    b += bytes([planes_byte]) * 4736
    for pol in (pol_a, pol_b):
        b += struct.pack("<H", len(pol))
        for idx, p in pol:
            b += struct.pack("<Hf", idx, p)
    return b


def test_hvm4_roundtrip(tmp_path):
    import hivemind_amd as hm
    recs = _sample_bytes(7, 411, 12, 1, 255, [(0, 0.5), (100, 0.5)], [(0, 1.0)]) + _sample_bytes(8, 399, 13, 0, 0, [(5, 1.0)], [(9, 0.25), (10, 0.75)])
    path = str(tmp_path / "chunk_1_000000.hvm")
    hm.write_chunk(path, np.frombuffer(recs, np.uint8), 2)
    raw = open(path, "rb").read()
    # header exactly as the reference reader expects ('<4sIHHQ': magic, version 4, 74 channels, 4672 policy, count)
    assert struct.unpack_from("<4sIHHQ", raw, 0) == (b"HVM4", 4, 74, 4672, 2)
    assert raw[struct.calcsize("<4sIHHQ"):] == recs
    assert not os.path.exists(path + ".tmp")                      # published atomically
    s = hm.read_hvm4(path)
    assert [x["game_id"] for x in s] == [7, 8] and s[0]["nodes"] == 411 and s[0]["planes"].shape == (74, 8, 8)
    assert np.all(s[0]["planes"] == 255) and s[1]["policy_b"]["index"].tolist() == [9, 10]
    assert abs(float(s[1]["policy_b"]["prob"].sum()) - 1.0) < 1e-6 and s[0]["outcome"] == 1 and s[0]["wdl"] == 2
    with pytest.raises(ValueError):
        open(path, "wb").write(b"HVM3" + raw[4:])
        hm.read_hvm4(path)


def test_selfplay_config_defaults_match_reference():
    """SelfPlayConfig defaults (tools/selfplay.h:10-31)."""
    import hivemind_amd as hm
    c = hm.default_selfplay_config()
    assert (c.games, c.nodes, c.max_macro_plies, c.chunk_samples) == (1, 800, 400, 16384)
    assert (c.raw_policy_mean_macro_plies, c.raw_policy_max_macro_plies, c.raw_policy_high_temperature_probability) == (8.0, 30, 0.05)
    assert (c.mcts_temperature, c.mcts_temperature_decay, c.mcts_temperature_plies) == (1.0, 0.93, 20)
    assert abs(c.resign_threshold + 0.9) < 1e-7 and c.resign_consecutive_plies == 3 and c.resign_disable_fraction == 0.10
    assert c.node_random_factor == 0.05 and abs(c.dirichlet_alpha - 0.3) < 1e-7 and abs(c.dirichlet_epsilon - 0.25) < 1e-7
    s = hm.default_config()          # search_params.h:26-273
    assert (s.cpuct_init, s.cpuct_base, s.fpu_reduction) == (2.5, 19652.0, 1.0)
    assert (s.pw_coefficient, s.root_pw_coefficient) == (2.0, 4.0) and abs(s.pw_exponent - 0.4) < 1e-7
    assert (s.enable_transpositions, s.enable_dynamic_fpu, s.enable_wdl_eval) == (1, 1, 1)
    assert abs(s.wdl_value_weight - 0.25) < 1e-7 and abs(s.moves_left_discount - 0.005) < 1e-7 and s.draw_contempt == 0.0


def test_c_abi_argument_validation():
    import hivemind_amd as hm
    lib = hm.lib
    assert lib.hm_hvm4_write_chunk(None, None, 0, 0) != 0 and b"null" in lib.hm_last_error()
    assert lib.hm_encode_planes(None, 4, 0, None, None) != 0          # not initialised / null pointers
    assert lib.hm_policy_index(0, 1) == 0


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = {port!r}
rank = int(sys.argv[1])
dist.init_process_group("gloo", rank=rank, world_size=2)
from hivemind_amd.selfplay import gather_records
rec = np.full(1000 + 500 * rank, rank + 1, np.uint8)
out, cnt = gather_records(rec, 3 + rank, dist)
if rank == 0:
    assert cnt == 7 and out.size == 2500 and np.all(out[:1000] == 1) and np.all(out[1000:] == 2), (cnt, out.size)
else:
    assert cnt == 0 and out.size == 0
# chunk boundaries: a rank with nothing to send at this boundary, then the reverse
for turn in range(2):
    mine = np.full(300, 7 + rank, np.uint8) if rank == turn else np.zeros(0, np.uint8)
    out, cnt = gather_records(mine, len(mine) // 100, dist)
    if rank == 0:
        assert cnt == 3 and out.size == 300 and np.all(out == 7 + turn), (turn, cnt, out.size)
    else:
        assert cnt == 0 and out.size == 0
dist.destroy_process_group()
print("ok", rank)
"""


def test_record_gather_world2_gloo(tmp_path):
    """The only cross-GPU exchange of the path (finished HVM4 bytes -> rank 0), on CPU with gloo."""
    script = tmp_path / "w.py"
    script.write_text(WORKER.format(root=ROOT, port="29517"))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_reference_shaped_state_dict_and_checkpoint_load(tmp_path):
    """Trained-weight ingestion.  The key/shape list in tests/golden/risev33_state_dict.json spells out the
    reference module's attribute paths (rise_mobile_v3.py / builder_util.py, read as text: the Python reference cannot be
    imported here, timm is absent) and its parameter count measured by the survey on the real module (14 122 085); a
    checkpoint file with the layout of the reference trainer (trainer_agent.py:871-890) must load strictly."""
    import json
    import torch
    from hivemind_amd import net as N
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "risev33_state_dict.json")))
    assert gold["trainable_parameters"] == 14122085
    torch.manual_seed(3)
    src = N.rise_v33()
    sd = src.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == gold["entries"]
    # names the reference code spells out literally
    for k in ("body_spatial.0.body.0.weight", "body_spatial.6.se.body.0.bias", "body_spatial.15.body.7.running_var",
              "value_head.body_wdl.0.weight", "value_head.body_plys.0.bias", "value_head.body_final.2.weight",
              "policy_heads.shared_body.1.num_batches_tracked", "policy_heads.board_projections.1.weight"):
        assert k in sd, k
    opt = torch.optim.SGD(src.parameters(), lr=0.1, momentum=0.9)
    path = tmp_path / "model-0.12345-0.678-0042.tar"
    torch.save({"model_state_dict": {"module." + k: v for k, v in sd.items()}, "optimizer_state_dict": opt.state_dict(),
                "training_iteration": 42, "batch_steps": 1000}, path)
    got = N.load_checkpoint(str(path))
    assert not got.training and all(torch.equal(v, got.state_dict()[k]) for k, v in sd.items())
    bad = dict(sd)
    bad.pop("value_head.body_wdl.0.bias")
    torch.save({"model_state_dict": bad}, tmp_path / "bad.tar")
    with pytest.raises(RuntimeError, match="Missing key"):
        N.load_checkpoint(str(tmp_path / "bad.tar"))


def _bench(args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=120)


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` with no torchrun environment: the parent starts two ranks before anything touches a GPU and
    forwards rank 0's line (the device-list shape of the reference's main.cc:154-173); stub workload, gloo, no GPU."""
    import json
    r = _bench(["--gpus", "2", "--workload", "stub", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    # value is the whole-job aggregate: both ranks' units (10 and 11 per step) over the slowest rank's time
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 21.0) < 1e-6
    one = json.loads([l for l in _bench(["--workload", "stub"]).stdout.splitlines() if l.startswith("{")][0])
    assert one["n_gpus"] == 1


def test_bench_launcher_fails_when_a_rank_fails():
    r = _bench(["--gpus", "2", "--workload", "stub"], {"HM_BENCH_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0 and "rank 1 failed" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
