"""tutel.net facade: groups + collectives (reference: tutel/net.py:6-12)."""
from .impls.communicate import (get_world_size, get_world_rank, create_groups_from_world,
                                create_standalone_group, barrier)
from .impls.communicate import (simple_all_reduce, simple_all_to_all, simple_split,
                                simple_reduce_scatter, simple_all_gather)
from .impls.communicate import (all_to_all, all_to_all_single, all_gather, zero_gather, zero_scatter,
                                spatial_split, reduce_scatter, allreduce_forward, allreduce_backward)
from .impls.communicate import batch_all_to_all_v, batch_all_gather_v


class TutelDistributedOptimizer:
    """ZeRO-1 style wrapper for the SHARED (non-expert) parameters (reference: tutel/net.py:15-60): every rank keeps and updates
    1/W of each shared parameter's flattened, zero-padded copy -- gradients arrive by reduce-scatter, the updated shards are
    all-gathered back into the full parameters -- while expert parameters (tagged `_tutel_expert` by MOELayer) are owned by
    their rank and stepped as they are.

        opt = TutelDistributedOptimizer(model.parameters(), group=None, average_shared=False).warp_local(torch.optim.SGD, lr=..)
        opt.zero_grad(); loss.backward(); opt.step()
    """

    def __init__(self, params, group=None, average_shared=False):
        params = list(params)
        self.params = [p for p in params if not hasattr(p, "_tutel_expert")]
        self.expert_params = [p for p in params if hasattr(p, "_tutel_expert")]
        self.shapes = [p.shape for p in self.params]
        self.group, self.average_shared = group, average_shared

    def chunk_param(self):
        self.virt_params = [zero_scatter(p.data, simple_split, group=self.group)[0] for p in self.params]

    def chunk_grad(self):
        W = get_world_size(self.group)
        for shard, p in zip(self.virt_params, self.params):
            if getattr(p, "grad", None) is not None:
                g = p.grad.view(-1) / W if self.average_shared else p.grad.view(-1)
                shard.grad, _ = zero_scatter(g, simple_reduce_scatter, group=self.group)

    def restore(self):
        for shard, p, shape in zip(self.virt_params, self.params, self.shapes):
            full = simple_all_gather(shard.data, group=self.group).view(-1)
            p.data = full[:shape.numel()].view(shape)

    def warp_local(self, local_optim, *args, **kwargs):
        self.chunk_param()
        self.local_optim = local_optim(self.virt_params + self.expert_params, *args, **kwargs)
        return self

    def zero_grad(self):
        for p in self.params + self.expert_params:
            if getattr(p, "grad", None) is not None:
                p.grad.detach_()
                p.grad.zero_()

    def step(self):
        self.chunk_grad()
        self.local_optim.step()
        self.restore()
