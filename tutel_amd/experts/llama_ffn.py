"""SwiGLU ("llama") per-expert FFN, `experts={'type': 'llama_ffn', 'hidden_size_per_expert': H, ..}`
(reference: tutel/experts/llama_ffn.py:7-44).

    y = (act(x @ W_fc1) * (x @ W_fc2)) @ W_fc3        per local expert, x [E_loc, R, M], no biases

Parameters keep the reference's names and its flat ZeRO-sharded storage (checkpoint compatible):
W_fc1 / W_fc2 hold ceil(E_loc*M*H / sharded_count) elements of the [E_loc, M, H] tensor, W_fc3 of
[E_loc, H, M]; all three are stored [K, N] row-major, i.e. the layout the grouped GEMM calls n-major.

Forward:
  * bf16 / fp16, no autograd, unsharded, recognised activation -> three launches of the MFMA
    grouped GEMM (tutel_amd_expert_gemm): the activation is fused into the W_fc1 launch and the
    gating product into the W_fc2 launch (its epilogue multiplies by the stored act(x @ W_fc1));
  * anything else -> ATen matmuls, op for op as the reference.
"""
import torch

from .. import net, ops
from .ffn import KMajorCache, classify_activation, _PREPACK


class LlamaFFNNetwork(torch.nn.Module):
    def _create_sharded_param(self, *full_shape, **kwargs):
        full_shape = torch.Size(full_shape)
        sharded = (full_shape.numel() + self.sharded_count - 1) // self.sharded_count
        return torch.nn.Parameter(torch.empty(sharded, **kwargs)), full_shape

    def _get_gathered_param(self, param, full_shape, parent_group):
        if self.sharded_count == 1:
            return param.view(full_shape)
        group = net.create_groups_from_world(group_count=-self.sharded_count, parent_group=parent_group).model_group
        return net.zero_gather(param, group=group).view(-1).narrow(0, 0, full_shape.numel()).view(full_shape)

    def __init__(self, model_dim, hidden_size_per_expert, num_experts_per_device, sharded_count,
                 activation_fn=torch.nn.functional.silu):
        super().__init__()
        self.sharded_count = sharded_count
        self.model_dim, self.output_dim = model_dim, model_dim
        self.W_fc1, self.W_fc1_full_shape = self._create_sharded_param(num_experts_per_device, model_dim, hidden_size_per_expert)
        self.W_fc2, self.W_fc2_full_shape = self._create_sharded_param(num_experts_per_device, model_dim, hidden_size_per_expert)
        self.W_fc3, self.W_fc3_full_shape = self._create_sharded_param(num_experts_per_device, hidden_size_per_expert, model_dim)
        self.activation_fn = activation_fn
        self._act_cache = {}
        self._kmajor = KMajorCache()
        self.reset_parameters()

    def reset_parameters(self):
        with torch.no_grad():
            self.W_fc1.normal_(0, 0.01)
            self.W_fc2.normal_(0, 0.01)
            self.W_fc3.normal_(0, 0.01)

    def extra_repr(self):
        return "model_dim=%d, hidden_size=%d, num_experts_per_device=%d, sharded_count=%d." % (
            self.W_fc1_full_shape[1], self.W_fc1_full_shape[2], self.W_fc1_full_shape[0], self.sharded_count)

    # -- fused path -------------------------------------------------------------------------
    def fused_activation(self):
        key = self.training
        if key not in self._act_cache:
            self._act_cache[key] = classify_activation(self.activation_fn)
        return self._act_cache[key]

    def can_fuse(self, x, ctx):
        if not x.is_cuda or self.sharded_count > 1 or torch.is_autocast_enabled():
            return False
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return False
        _, M, H = self.W_fc1_full_shape
        return (x.dim() == 3 and x.dtype == self.W_fc1.dtype and ops.gemm_supported(x.dtype, H, M)
                and ops.gemm_supported(x.dtype, M, H) and self.fused_activation() is not None)

    def forward_fused(self, x):
        w1, w2, w3 = (self.W_fc1.view(self.W_fc1_full_shape), self.W_fc2.view(self.W_fc2_full_shape),
                      self.W_fc3.view(self.W_fc3_full_shape))
        km = _PREPACK and not self.training  # eval: weights laid out k-major once (see KMajorCache)
        if km:
            w1, w2, w3 = self._kmajor.get("fc1", w1), self._kmajor.get("fc2", w2), self._kmajor.get("fc3", w3)
        g = ops.expert_gemm(x, w1, None, km, act=self.fused_activation())
        h = ops.expert_gemm(x, w2, None, km, mul=g)
        return ops.expert_gemm(h, w3, None, km)

    def invalidate_prepacked(self):
        self._kmajor.invalidate()

    def train(self, mode=True):
        self._kmajor.invalidate()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._kmajor.invalidate()
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self._kmajor.invalidate()
        return super()._load_from_state_dict(*args, **kwargs)

    # -- reference-equivalent ATen path -----------------------------------------------------
    def forward(self, x, ctx):
        if self.can_fuse(x, ctx):
            return self.forward_fused(x.contiguous())
        w1 = self._get_gathered_param(self.W_fc1, self.W_fc1_full_shape, ctx.group)
        w2 = self._get_gathered_param(self.W_fc2, self.W_fc2_full_shape, ctx.group)
        w3 = self._get_gathered_param(self.W_fc3, self.W_fc3_full_shape, ctx.group)
        y = self.activation_fn(torch.matmul(x, w1)) * torch.matmul(x, w2)
        return torch.matmul(y, w3)


ExpertModule = LlamaFFNNetwork
