"""Linear top-k gate, `{'type': 'top', 'k': .., 'fp32_gate': .., 'capacity_factor': .., 'gate_noise': ..}`
(reference: tutel/gates/top.py).

forward(x) returns the routing LOGITS [T, E]; softmax and the top-k selection are fused into the
routing kernel (tutel_amd_gate_topk).  The [T,M] x [M,E] projection itself is a small plain GEMM
and stays on the vendor library (hipBLASLt via F.linear).  With fp32_gate the projection runs in
fp32 whatever the expert dtype -- exact ties between scores then have measure zero, which makes
the token->expert assignment independent of any tie rule (hard part 1).
"""
import torch
import torch.nn.functional as F

_ALLOWED_EXTRA = frozenset(("capacity_factor", "gate_noise"))  # consumed by MOELayer, tolerated here


class LinearTopKGate(torch.nn.Module):
    def __init__(self, model_dim, num_global_experts, k=1, fp32_gate=False, **options):
        unknown = [name for name in options if name not in _ALLOWED_EXTRA]
        if unknown:
            raise Exception("Unrecognized argument provided to Gating module: %s" % unknown[0])
        super().__init__()
        self.fp32_gate = bool(fp32_gate)
        self.top_k = min(int(k), num_global_experts)
        # parameter name `wg.weight` [E, M] is part of the checkpoint format
        self.wg = torch.nn.Linear(model_dim, num_global_experts, bias=False,
                                  **({"dtype": torch.float32} if self.fp32_gate else {}))

    def forward(self, x):
        weight = self.wg.weight
        if self.fp32_gate and weight.dtype != torch.float32:  # module was cast as a whole (.half()/.bfloat16())
            weight = weight.float()
        return F.linear(x if x.dtype == weight.dtype else x.to(weight.dtype), weight)


Gate = LinearTopKGate
