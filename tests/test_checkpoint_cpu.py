"""tutel.checkpoint.gather / scatter: per-rank expert-parallel checkpoints <-> one full
checkpoint.  Pure host logic; compared against the reference's own CLI tools when /root/reference exists."""
import os
import subprocess
import sys

import pytest
import torch

REF = os.environ.get("TUTEL_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_states(size, E_global, H=6, M=4, seed=0):
    """state dicts as `size` ranks of one MoE layer would save them (moe_layer.py:119-143 shapes)."""
    g = torch.Generator().manual_seed(seed)
    if E_global % size == 0:
        E_loc, Hs = E_global // size, H
    else:
        E_loc, Hs = 1, H // (size // E_global)
    out = []
    shared = {"embed.weight": torch.randn(3, M, generator=g), "moe.gates.0.wg.weight": torch.randn(E_global, M, generator=g)}
    for _ in range(size):
        sd = {k: v.clone() for k, v in shared.items()}
        sd["moe._num_global_experts"] = torch.tensor(E_global)
        sd["moe.experts.batched_fc1_w"] = torch.randn(E_loc, Hs, M, generator=g)
        sd["moe.experts.batched_fc2_w"] = torch.randn(E_loc, Hs, M, generator=g)
        sd["moe.experts.batched_fc1_bias"] = torch.randn(E_loc, Hs, generator=g)
        sd["moe.experts.batched_fc2_bias"] = torch.randn(E_loc, M, generator=g)
        out.append(sd)
    return out


@pytest.mark.parametrize("size,E_global,namespace", [(4, 8, ""), (2, 2, "model/state"), (4, 2, "")])
def test_gather_scatter_round_trip_and_reference_tools(tmp_path, size, E_global, namespace):
    from tutel.checkpoint import gather, scatter
    states = _rank_states(size, E_global)

    def wrap(sd):
        for package in reversed([p for p in namespace.split("/") if p]):
            sd = {package: sd}
        return sd

    def unwrap(obj):
        for package in namespace.split("/"):
            if package:
                obj = obj[package]
        return obj

    inputs = str(tmp_path / "in" / "r{rank}-of-{size}.pt")
    for r, sd in enumerate(states):
        os.makedirs(tmp_path / "in", exist_ok=True)
        torch.save(wrap(sd), inputs.format(rank=r, size=size))
    full_path = str(tmp_path / "full.pt")
    gather(inputs, size, full_path, namespace)
    full = unwrap(torch.load(full_path))
    assert full["moe.experts.batched_fc1_w"].shape[0] == E_global
    if E_global % size == 0:
        assert torch.equal(full["moe.experts.batched_fc1_w"], torch.cat([s["moe.experts.batched_fc1_w"] for s in states]))
    else:  # each expert's hidden rows were spread over size/E ranks
        assert full["moe.experts.batched_fc1_w"].shape == (E_global, 6, 4)
        assert torch.equal(full["moe.experts.batched_fc1_w"][0, :3], states[0]["moe.experts.batched_fc1_w"][0])
    assert torch.equal(full["embed.weight"], states[-1]["embed.weight"]) and int(full["moe._num_global_experts"]) == E_global
    if E_global % size == 0:  # (biases shared by the shards of one expert do not survive a gather: as in the reference)
        outs = scatter(full_path, str(tmp_path / "out" / "r{rank}-of-{size}.pt"), size, namespace)
        for r, f in enumerate(outs):
            back = unwrap(torch.load(f))
            assert set(back) == set(states[r])
            for k in states[r]:
                assert torch.equal(back[k], states[r][k]), (r, k)
    # re-shard to another world size and gather again: same full checkpoint
    other = size // 2
    outs = scatter(full_path, str(tmp_path / "half" / "r{rank}-of-{size}.pt"), other, namespace)
    if E_global % other == 0:
        gather(str(tmp_path / "half" / "r{rank}-of-{size}.pt"), other, str(tmp_path / "full2.pt"), namespace)
        full2 = unwrap(torch.load(str(tmp_path / "full2.pt")))
        assert all(torch.equal(full[k], full2[k]) for k in full)

    if not os.path.isdir(os.path.join(REF, "tutel")):
        return
    # the reference's own tools on the same files
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([REF, os.path.join(ROOT, "oracle", "_ref")]))
    ref_full = str(tmp_path / "ref_full.pt")
    subprocess.check_call([sys.executable, "-m", "tutel.checkpoint.gather", f"--input_size={size}", f"--inputs={inputs}",
                           f"--output={ref_full}", f"--namespace={namespace}"], env=env, cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    rf = unwrap(torch.load(ref_full))
    assert set(rf) == set(full) and all(torch.equal(rf[k], full[k]) for k in full), "gather differs from the reference tool"
    if E_global % other == 0:
        subprocess.check_call([sys.executable, "-m", "tutel.checkpoint.scatter", f"--output_size={other}", f"--input={ref_full}",
                               f"--outputs={tmp_path}/refhalf/r{{rank}}-of-{{size}}.pt", f"--namespace={namespace}"],
                              env=env, cwd=str(tmp_path), stdout=subprocess.DEVNULL)
        for r, f in enumerate(outs):
            mine, ref = unwrap(torch.load(f)), unwrap(torch.load(f"{tmp_path}/refhalf/r{r}-of-{other}.pt"))
            assert set(mine) == set(ref) and all(torch.equal(mine[k], ref[k]) for k in mine), "scatter differs from the reference tool"


def test_gather_needs_the_expert_count(tmp_path):
    from tutel.checkpoint import gather
    sd = {"moe.experts.batched_fc1_w": torch.zeros(2, 3, 4)}
    torch.save(sd, tmp_path / "r0.pt")
    with pytest.raises(Exception, match="default_num_global_experts"):
        gather(str(tmp_path / "r{rank}.pt"), 1, str(tmp_path / "o.pt"))
    gather(str(tmp_path / "r{rank}.pt"), 1, str(tmp_path / "o.pt"), default_num_global_experts=2)
    assert int(torch.load(tmp_path / "o.pt")["moe._num_global_experts"]) == 2
