"""RISEv3 numerics: the fused single-launch HIP forward (hm_net_forward) and the folded fp16 torch
form against a plain PyTorch fp32 reference of the same weights.  Tolerance: 1e-3 absolute on value,
wdl, moves-left, and on policy logits relative to the logit scale (north_star: "value/policy logits
within 1e-3")."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _randomise_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1, generator=g)
            m.running_var.uniform_(0.5, 1.5, generator=g)
            m.weight.data.uniform_(0.5, 1.5, generator=g)
            m.bias.data.normal_(0, 0.1, generator=g)


@pytest.mark.parametrize("name", ["small", "full"])
def test_fused_forward_matches_fp32_reference(hm, name):
    import oracle_py as O
    from hivemind_amd import net as N
    torch.manual_seed(0)
    model = N.rise_v3_small() if name == "small" else N.rise_v33()
    _randomise_bn(model, 1)
    model.eval()
    boards = O.random_positions(3, 96, 120)
    planes = hm.board_to_planes(hm.to_device(boards), "f16")
    with torch.no_grad():
        ref = model.cuda().float()(planes.float())
    fused = N.FusedNet(model)
    got = fused(planes)
    torch.cuda.synchronize()
    lib = N.InferenceNet(model)(planes)
    names = ("value", "pi_a", "pi_b", "wdl", "moves_left")
    report = {}
    for nme, r, g, l in zip(names, ref, got, lib):
        r = r.float().reshape(g.shape)
        scale = max(1.0, float(r.abs().max()))
        err = float((g.float() - r).abs().max())
        err_lib = float((l.float().reshape(g.shape) - r).abs().max())
        # the fp16 output format alone costs half an ulp of the largest logit: 2^-11 * scale
        report[nme] = dict(max_abs_err=err, library_fp16_err=err_lib, logit_scale=scale, fp16_ulp_at_scale=scale * 2.0 ** -10)
        assert err / scale < 1e-3, (name, nme, err, err_lib, scale)          # north_star: value / policy logits within 1e-3
    print("\nnumerics[%s] fused HIP forward vs fp32 torch: %s" % (name, json.dumps(report)))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(report, open(f"gpurun_out/net_numerics_{name}.json", "w"), indent=1)
    # batch-size independence and ragged batches (1 row, odd counts)
    for n in (1, 7, 33):
        sub = fused(planes[:n].contiguous())
        for a, b in zip(sub, got):
            assert torch.equal(a, b[:n])


def test_ragged_groups_skip_unused_rows(hm):
    """hm_net_forward_groups: live rows equal the dense forward bit for bit, skipped rows stay untouched."""
    import oracle_py as O
    from hivemind_amd import net as N
    torch.manual_seed(0)
    model = N.rise_v3_small()
    model.eval()
    boards = O.random_positions(5, 64, 100)
    planes = hm.board_to_planes(hm.to_device(boards), "f16")
    fused = N.FusedNet(model)
    dense = [t.clone() for t in fused(planes)]
    counts = torch.tensor([8, 0, 3, 1, 7, 8, 0, 5], dtype=torch.int32, device="cuda")
    out = tuple(torch.full_like(t, 7.0) for t in dense)
    fused(planes, out=out, group_rows=counts, group=8)
    torch.cuda.synchronize()
    live = torch.zeros(64, dtype=torch.bool)
    for g, c in enumerate(counts.tolist()):
        live[g * 8:g * 8 + c] = True
    live = live.cuda()
    for d, o in zip(dense, out):
        assert torch.equal(o[live], d[live])
        assert bool((o[~live] == 7.0).all())
