"""Linear top-k gate (reference: tutel/gates/top.py).  forward(x) -> logits [T, E]; the softmax
and the top-k selection happen in the fused routing kernel (tutel_amd_gate_topk).  The tiny
[T,M]x[M,E] projection is a plain library GEMM (hipBLASLt through torch.nn.functional.linear)."""
import torch


class LinearTopKGate(torch.nn.Module):
    def __init__(self, model_dim, num_global_experts, k=1, fp32_gate=False, **options):
        super().__init__()
        for opt in options:
            if opt not in ("capacity_factor", "gate_noise"):
                raise Exception("Unrecognized argument provided to Gating module: %s" % opt)
        self.wg = torch.nn.Linear(model_dim, num_global_experts, bias=False,
                                  dtype=torch.float32 if fp32_gate else None)
        self.top_k = min(num_global_experts, int(k))
        self.fp32_gate = fp32_gate

    def forward(self, x):
        wg = self.wg.float() if self.fp32_gate else self.wg
        return wg(x.to(dtype=wg.weight.dtype))


Gate = LinearTopKGate
