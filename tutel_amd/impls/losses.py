"""Auxiliary losses of the MoE gate (reference: tutel/impls/losses.py).

On the forward hot path the gshard loss is produced by the routing kernels themselves
(tutel_amd_compute_location, l_aux output); the torch versions here are the differentiable
forms used when the scores require grad, and the callable identity `gshard_loss` that
`extract_critical(loss_fn=...)` recognises."""
import torch


def _one_hot_with_dtype(data, num_classes, dtype, hot_value=1):
    out = torch.zeros([data.size(0), num_classes], device=data.device, dtype=dtype)
    out.scatter_(1, data.unsqueeze(-1).long(), hot_value)
    return out


def gshard_loss(scores_w_noise, top_ids):
    """l_aux = (1/T) * sum_e (sum_t scores[t,e]) * (count_first_choice[e] * E / T)   (losses.py:12-19)."""
    T, E = int(scores_w_noise.size(0)), int(scores_w_noise.size(1))
    ce = torch.sum(_one_hot_with_dtype(top_ids[:, 0], E, dtype=scores_w_noise.dtype, hot_value=E / T), dim=0)
    me = torch.sum(scores_w_noise, dim=0)
    return torch.sum(me * ce) / T


def load_importance_loss(scores_wo_noise, topk_logits, num_global_experts, gate_noise):
    """Importance + load balancing loss (losses.py:21-42); training only, needs gate_noise > 0."""
    assert gate_noise > 0, "`gate_noise` must be > 0 for normalization in load_importance_loss()."
    s = scores_wo_noise.float()
    imp = s.sum(0)
    l_imp = imp.var() / (imp.mean() ** 2 + 1e-10)
    normal = torch.distributions.normal.Normal(
        torch.tensor([0.0], device=s.device), torch.tensor([gate_noise / num_global_experts], device=s.device))
    load = normal.cdf(s - topk_logits[:, -1].view(-1, 1).float()).sum(0)
    l_load = load.var() / (load.mean() ** 2 + 1e-10)
    return (l_imp + l_load) / 2.0
