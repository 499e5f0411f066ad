"""Cosine-similarity top-k gate, `{'type': 'cosine_top', 'k': .., 'proj_dim': 256, 'init_t': 0.5, ..}`
(reference: tutel/gates/cosine_top.py:7-34).

forward(x) returns the routing LOGITS [T, E]:
    logits = normalize(proj(x), dim=1) @ normalize(sim_matrix, dim=0) * exp(min(temperature, log 100))
Parameter names and shapes (`temperature` [1], `cosine_projector.{weight,bias}` [P,M]/[P],
`sim_matrix` [P,E]) are the checkpoint format.  The two small projections are plain library GEMMs;
softmax + top-k + locations run in the fused routing kernels exactly as for the linear gate.
"""
import torch
import torch.nn.functional as F

_ALLOWED_EXTRA = frozenset(("capacity_factor", "gate_noise"))


class CosineTopKGate(torch.nn.Module):
    def __init__(self, model_dim, num_global_experts, k=1, fp32_gate=False, proj_dim=256, init_t=0.5, **options):
        unknown = [name for name in options if name not in _ALLOWED_EXTRA]
        if unknown:
            raise Exception("Unrecognized argument provided to Gating module: %s" % unknown[0])
        super().__init__()
        self.top_k = min(num_global_experts, int(k))
        self.fp32_gate = bool(fp32_gate)
        # creation order = the reference's RNG order: temperature, projector, sim_matrix (randn, then normal_)
        self.temperature = torch.nn.Parameter(torch.log(torch.full([1], 1.0 / init_t)), requires_grad=True)
        self.cosine_projector = torch.nn.Linear(model_dim, proj_dim)
        self.sim_matrix = torch.nn.Parameter(torch.randn(size=(proj_dim, num_global_experts)), requires_grad=True)
        self.clamp_max = torch.log(torch.tensor(1. / 0.01)).item()  # fp32 log, as the reference computes it
        torch.nn.init.normal_(self.sim_matrix, 0, 0.01)

    def forward(self, x):
        w, b, sim, temp = self.cosine_projector.weight, self.cosine_projector.bias, self.sim_matrix, self.temperature
        if self.fp32_gate:
            x, w, b, sim = x.float(), w.float(), b.float(), sim.float()
        elif x.dtype != w.dtype:
            x = x.to(w.dtype)
        logits = torch.matmul(F.normalize(F.linear(x, w, b), dim=1), F.normalize(sim, dim=0))
        return logits * torch.clamp(temp, max=self.clamp_max).exp()


Gate = CosineTopKGate
