"""Pure host-side tensor bookkeeping: which state-dict entries are expert parameters (everything under
`<layer>.experts.` of a module that carries the `_num_global_experts` buffer, moe_layer.py:119) and how
their leading "expert" dimension maps onto ranks.

Two regimes, as in the layer itself (moe_layer.py:121-143):
  E % size == 0 : every rank owns E/size whole experts -> shards are dim-0 slices;
  size % E == 0 : every expert is sharded over size/E ranks along its hidden dimension (the first
                  dimension after dim 0 that is larger than 1: batched_fc1_w [1, H/s, M], fc1_bias [1, H/s],
                  batched_fc2_w [1, H/s, M_out]) -> shards are [1, H/s, ...] row groups.
(reference: gather.py:59-71 concatenates the per-rank pieces along dim 0 and re-views them; scatter.py:37-50
cuts the flat storage into `size` equal chunks -- the same bytes per rank as the row groups produced here.)
"""
import torch

from ..system import apply_rank_size_from_pattern


def _descend(obj, namespace):
    for package in namespace.split("/"):
        if package:
            obj = obj[package]
    return obj


def find_moe_layers(state_dict, default_num_global_experts=0):
    """{'<layer>.experts.': E_global} for every MoE layer in `state_dict`, plus the list of layers that
    had no `_num_global_experts` entry (legacy checkpoints; then `default_num_global_experts` is used)."""
    layers, legacy = {}, []
    for key, value in state_dict.items():
        if key.endswith("._num_global_experts"):
            layers[key[:key.rindex(".")] + ".experts."] = int(value)
    if not layers and default_num_global_experts > 0:
        for key in state_dict:
            if ".experts." in key:
                entry = key[:key.rindex(".experts.")]
                if entry + ".experts." not in layers:
                    layers[entry + ".experts."] = int(default_num_global_experts)
                    legacy.append(entry)
    return layers, legacy


def merge_expert_shards(pieces, num_global_experts, size):
    """Per-rank tensors (rank order) of one expert parameter -> the full [E_global, ...] tensor."""
    full = torch.cat(list(pieces), dim=0).contiguous()
    if num_global_experts % size == 0:
        assert full.size(0) == num_global_experts, \
            "Unexpected group size of expert with num_global_experts: %d v.s. %d. Maybe you set a wrong --size value." % (full.size(0), num_global_experts)
        return full.clone()
    if size % num_global_experts == 0:
        return full.view([num_global_experts, -1] + list(full.shape)[2:]).clone()
    raise Exception(f'Neither of "global_experts({num_global_experts}) / args.size({size})" nor '
                    f'"args.size({size}) / global_experts({num_global_experts})" is evenly divisible.')


def split_expert_param(full, size):
    """Full [E_global, ...] tensor -> list of `size` per-rank tensors (inverse of merge_expert_shards)."""
    shape = list(full.shape)
    if shape[0] % size == 0:
        return [t.contiguous().clone() for t in full.view([size, shape[0] // size] + shape[1:])]
    if size % shape[0] == 0:
        divisor = size // shape[0]
        dim = next((i for i in range(1, len(shape)) if shape[i] > 1), None)
        assert dim is not None and shape[dim] % divisor == 0, \
            f"The second non-squeezable dimension is to be sliced to {divisor} pieces from an parameter of shape {shape}, which isn't divisible evenly."
        out = []
        for e in range(shape[0]):
            out += [t.contiguous().clone() for t in full[e:e + 1].chunk(divisor, dim=dim)]
        return out
    raise Exception(f'Neither of "global_experts({shape[0]}) / args.size({size})" nor '
                    f'"args.size({size}) / global_experts({shape[0]})" is evenly divisible.')


def gather(inputs, input_size, output, namespace="", default_num_global_experts=0):
    """`input_size` per-rank checkpoints (file pattern with {rank} / {size}) -> one checkpoint whose expert
    parameters hold all E_global experts; every other entry is taken from the last rank's file."""
    first = _descend(torch.load(apply_rank_size_from_pattern(inputs, rank=0, size=input_size), map_location="cpu"), namespace)
    layers, legacy = find_moe_layers(first, default_num_global_experts)
    if not layers:
        raise Exception("Failed to detect Tutel MoE layer in the checkpoint,\n\tas the provided checkpoint may be in legacy format "
                        "with field `_num_global_experts` missing.\nPlease try again by manually providing the designed number "
                        "of total experts using: --default_num_global_experts=?")
    pieces, bundle, state = {}, None, None
    for rank in range(input_size):
        bundle = torch.load(apply_rank_size_from_pattern(inputs, rank=rank, size=input_size), map_location="cpu")
        state = _descend(bundle, namespace)
        for entry in legacy:
            state[entry + "._num_global_experts"] = default_num_global_experts
        for key, value in state.items():
            if any(key.startswith(prefix) for prefix in layers):
                pieces.setdefault(key, []).append(value)
    for key, parts in pieces.items():
        E = next(layers[prefix] for prefix in layers if key.startswith(prefix))
        state[key] = merge_expert_shards(parts, E, input_size)
    torch.save(bundle, output)
    return output


def scatter(input, outputs, output_size, namespace=""):  # noqa: A002
    """One gathered checkpoint -> `output_size` per-rank checkpoints (file pattern with {rank} / {size})."""
    bundle = torch.load(input, map_location="cpu")
    state = _descend(bundle, namespace)
    layers, _ = find_moe_layers(state)
    if not layers:
        raise Exception("No any Tutel MoE layer is found, as the provided checkpoint may be in legacy format. You need to reload this "
                        "legacy checkpoint by corresponding application, re-checkpoint model's state_dict and get the latest format.")
    shards = {key: split_expert_param(value, output_size) for key, value in state.items()
              if any(key.startswith(prefix) for prefix in layers)}
    full = dict(state)
    files = []
    for rank in range(output_size):
        for key in full:
            state[key] = shards[key][rank] if key in shards else full[key]
        files.append(apply_rank_size_from_pattern(outputs, rank=rank, size=output_size))
        torch.save(bundle, files[-1])
    return files
