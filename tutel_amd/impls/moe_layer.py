"""MOELayer -- the drop-in for tutel.moe.moe_layer (reference: tutel/impls/moe_layer.py).

Forward on MI355X (one process per GPU):

    x[..., M] -> gate logits (library GEMM)                                   gates/top.py
      -> fused softmax + top-k + stable-rank locations + l_aux (2 HIP kernels)  fast_dispatch.extract_critical
      -> fast_encode: bucket-major scatter into [E, C, M] (1 HIP kernel)
      -> all_to_all_single over RCCL/xGMI (W > 1)            raw layout [W, E_loc, C, M]
      -> expert FFN: 2 MFMA grouped-GEMM launches that READ and WRITE the raw all-to-all
         layout directly (the reference's two permute+contiguous copies are folded into the
         GEMM's row addressing), bias/activation fused, dropless row counts on device
      -> all_to_all_single back -> fast_decode: k-way gather, fp32 combine (1 HIP kernel)

The constructor / forward signatures, attributes, env switches (SKIP_MOE, BATCH_PRIO, CAP_FACTOR)
and error behaviour follow the reference so existing user code runs unchanged.
"""
import importlib
import logging
import os
import re

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch.nn import ModuleList

from . import communicate as C
from . import losses
from .fast_dispatch import RoutingPlan, extract_critical, fast_decode, fast_encode, get_dispatch_count
from .overlap import a2a_ffn_overlap_forward, a2a_ffn_overlap_fused
from . import ep_native
from .. import ops
from ..experts.ffn import FusedExpertsNetwork
from ..gates.top import LinearTopKGate


# test hook: run the overlapped expert-parallel path even with a single rank (exercises its
# stream / event / allocator discipline on one GPU)
_FORCE_OVERLAP = int(os.environ.get("TUTEL_AMD_FORCE_OVERLAP", "0")) != 0
# A/B switch: gather fc1's rows from the tokens (fused fast_encode) on the single-rank path
_FUSE_ENCODE = int(os.environ.get("TUTEL_AMD_FUSE_ENCODE", "1")) != 0
# A/B switch: the projection of a 16-bit linear gate inside the native call (csrc/gate_proj.hip) instead of F.linear
_NATIVE_GATE = int(os.environ.get("TUTEL_AMD_NATIVE_GATE", "1")) != 0


def _autocast_dtype(t):
    return torch.get_autocast_dtype(t.device.type)


class MOELayer(torch.nn.Module):
    """Tutel-compatible Mixture-of-Experts layer."""

    @staticmethod
    def global_expert_count(num_local_experts, group=None):
        if not isinstance(num_local_experts, int):  # a fraction 1/n: n devices share one expert
            num_local_experts = -int(1 / (num_local_experts + 1e-5))
        world = C.get_world_size(group)
        if num_local_experts == 0:
            raise Exception("Invalid value of num_local_experts: %d" % num_local_experts)
        if num_local_experts > 0:
            return num_local_experts * world
        assert world % -num_local_experts == 0, \
            f"Excepting {-num_local_experts} devices to share an expert param, while global device count is {world}."
        return world // -num_local_experts

    @property
    def num_global_experts(self):
        # The buffer `_num_global_experts` (checkpoint format, reference moe_layer.py:119) lives on
        # the device after .cuda(); int() of it is a D2H copy + stream sync.  The forward path
        # reads this value several times, so it is mirrored in a host int.
        return self._num_global_experts_host

    def __init__(self, gate_type, model_dim: int, experts=None, scan_expert_func=None, result_func=None,
                 group=None, seeds=None, a2a_ffn_overlap_degree=1, is_postscore=True,
                 batch_prioritized_routing=False, normalize_gate=True, is_gshard_loss=True,
                 parallel_type="adaptive:1", use_2dh=False, **kwargs):
        super().__init__()
        assert model_dim % 2 == 0, "Model_dim (%s) must be even value, while this Model_dim mod 2 > 0." % model_dim
        if "pad_samples" in kwargs:
            logging.warning("`pad_samples` option in Tutel Moe-layer has been deprecated, as Tutel always assumes `pad_samples=False` for better efficiency.")
            kwargs.pop("pad_samples")
        for key in kwargs:
            raise Exception("Unrecognized argument provided to Tutel Moe-layer: %s" % key)

        self.group = group or (dist.group.WORLD if dist.is_available() and dist.is_initialized() else None)
        self.result_func = result_func
        self.skip_moe = int(os.environ.get("SKIP_MOE", "0")) != 0
        self.model_dim = model_dim
        self.is_postscore = is_postscore
        self.batch_prioritized_routing = batch_prioritized_routing or int(os.environ.get("BATCH_PRIO", 0)) != 0
        self.normalize_gate = normalize_gate
        self.is_gshard_loss = is_gshard_loss
        self.a2a_ffn_overlap_degree = a2a_ffn_overlap_degree
        self.use_2dh = use_2dh
        self.megablocks_size = 0
        self.protected_shape = None

        experts = dict(experts)
        n_local = experts.pop("count_per_node", 1) if "count_per_node" in experts else experts.pop("num_experts_per_device", 1)
        self.num_local_experts = 1 if n_local == -1 else n_local
        self._num_global_experts_host = int(MOELayer.global_expert_count(self.num_local_experts, self.group))
        self.register_buffer("_num_global_experts", torch.tensor(self._num_global_experts_host))
        self.world_size = C.get_world_size(self.group)
        if self.num_global_experts < self.world_size:
            self.sharded_count = self.world_size // self.num_global_experts
            self.num_local_experts = 1
        else:
            self.sharded_count = 1

        # how the ranks that share one expert split its work (moe_layer.py:150-166 upstream): r = 0 gathers the expert's parameters and
        # keeps every token local; 1 <= r <= sharded_count, a divisor of it, replicates a token bucket r times across hidden-dim shards
        self.auto_parallel, self.use_model_parallel = False, True
        self.valid_rs = [0] + [d for d in range(1, self.sharded_count + 1) if self.sharded_count % d == 0]
        self.adaptive_degree = self._degree_of_parallel_type(parallel_type)

        if seeds is not None and seeds[1] is not None:
            torch.manual_seed(seeds[1])
        self.experts = self._build_experts(experts)
        if scan_expert_func is not None:
            for n, p in self.experts.named_parameters():
                scan_expert_func(n, p)
        for _, p in self.experts.named_parameters():
            setattr(p, "_tutel_expert", True)

        if isinstance(gate_type, str):
            assert re.match(r"^Top[0-9]+Gate$", gate_type), "Unrecognized gate_type: %s" % gate_type
            k = int(gate_type[3:-4])
            logging.warning(f"gate_type value `{gate_type}` in Tutel Moe-layer has been deprecated, please use gate_type = {{'type': 'top', 'k': {k}}} instead.")
            gate_type = {"type": "top", "k": k}
        gate_specs = gate_type if isinstance(gate_type, list) else [gate_type]
        self.gates = ModuleList([self._build_gate(dict(spec), gi, seeds) for gi, spec in enumerate(gate_specs)])

        if seeds is not None and len(seeds) > 2 and seeds[2] is not None:
            torch.manual_seed(seeds[2])

    # ---- construction helpers -----------------------------------------------------------
    def _build_experts(self, spec):
        kind = spec.pop("type")
        if kind == "custom":
            module = spec.pop("module")
        else:
            assert re.match(r"[a-zA-Z0-9\_]+", kind), "Expert type must only include digits, letters and underline characters."
            try:
                module = importlib.import_module(f"...experts.{kind}", __name__).ExpertModule
            except ModuleNotFoundError:
                raise Exception("Builtin expert type is not recognized: %s" % kind)
            if kind == "ffn":
                assert "fused_custom_fn" not in spec, "`fused_custom_fn` option for Tutel Moe-layer has been deprecated, please follows helloworld_from_scratch.py for custom construction instead."
                assert "implicit_dropout_p" not in spec, "`implicit_dropout_p` option for Tutel Moe-layer has been deprecated, please use torch.nn.Dropout(p=implicit_dropout_p) on custom activation_fn (for fc1_dropout) and after Tutel Moe-layer (for fc2_dropout) instead."
        spec.update(model_dim=self.model_dim, num_experts_per_device=self.num_local_experts, sharded_count=self.sharded_count)
        try:
            return module(**spec)
        except TypeError:
            logging.warning("\nExpertModule.__init__(.., local_experts, ..) has been deprecated, please rename `local_experts` to `num_experts_per_device` in init methods.\n")
            spec["local_experts"] = spec.pop("num_experts_per_device")
            return module(**spec)

    def _build_gate(self, spec, gi, seeds):
        kind = spec.pop("type")
        assert re.match(r"[a-zA-Z0-9\_]+", kind), "Gate type must only include digits, letters and underline characters."
        if seeds is not None and seeds[0] is not None:
            torch.manual_seed(seeds[0] + gi)
        if kind == "custom":
            factory = spec.pop("module")
        else:
            try:
                factory = importlib.import_module(f"...gates.{kind}", __name__).Gate
            except ModuleNotFoundError:
                raise Exception("Unrecognized gate_type: %s" % kind)
        gate = factory(model_dim=self.model_dim, num_global_experts=self.num_global_experts, **spec)
        if not hasattr(gate, "gate_noise"):
            gate.gate_noise = spec.get("gate_noise", 0.0)
        if not hasattr(gate, "capacity_factor"):
            gate.capacity_factor = spec.get("capacity_factor", float(os.environ.get("CAP_FACTOR", 1.0)))
        return gate

    # ---- checkpoint compatibility (reference moe_layer.py:57-78) ---------------------------
    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        key = prefix + "_num_global_experts"
        if key not in state_dict:
            logging.warning("Loading a legacy Tutel checkpoint without `_num_global_experts`; it will be written in the new format next time.")
            state_dict[key] = self._num_global_experts
        else:
            have, want = int(state_dict[key]), self.num_global_experts
            assert have == want, "Failed to load state from checkpoint: the number of global experts mismatch (%s <- %s)" % (want, have)
        for name, param in self.experts.named_parameters():
            k = prefix + "experts." + name
            if k not in state_dict:
                logging.warning("Could not find parameter `%s` in state_dict, zero values will be filled into this parameter." % k)
                state_dict[k] = torch.zeros_like(param)
            if state_dict[k].numel() == param.numel():
                state_dict[k] = state_dict[k].view(param.shape)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def extra_repr(self):
        return "Top-K(s) = %s, Total-Experts = %d [managed by %d device(s)]," % (
            [f"k={g.top_k}, noise={g.gate_noise}" for g in self.gates], self.num_global_experts, self.world_size)

    def get_parameter_iterator(self, param_type):
        if param_type == "gate":
            return self.gates.named_parameters()
        if param_type == "local_experts":
            return self.experts.named_parameters()
        raise Exception("Specified parameter type is not recognized: %s. Valid `param_type` includes: gate, local_experts." % param_type)

    # ---- experts ------------------------------------------------------------------------
    def expert_local(self, x, reserve_shape):
        y = self.experts(x.view(x.size(0), x.size(1), *reserve_shape), self)
        self.protected_shape = y.shape
        return y.reshape(y.size(0), y.size(1), -1)

    def _experts_on_raw_a2a(self, y, reserve_shape):
        """W > 1 fast path: all-to-all the [E,C,M] buckets as they are and let the grouped GEMM
        address the raw [W,E_loc,C,M] exchange buffers on both sides (no permute copies)."""
        W, E_loc = self.world_size, self.num_local_experts
        E, Cc, M = y.shape
        recv = C.simple_all_to_all(y, group=self.group)  # [W(src), E_loc, C, M] flat
        Mo = self.experts.output_dim
        send = torch.empty([E, Cc, Mo], dtype=y.dtype, device=y.device)
        self.experts.forward_fused(recv, self, a_layout=(Cc * M, E_loc * Cc * M, Cc, M), R=W * Cc,
                                   out=send, d_layout=(Cc * Mo, E_loc * Cc * Mo, Cc, Mo))
        self.protected_shape = torch.Size([E_loc, W * Cc, Mo])
        return C.simple_all_to_all(send, group=self.group)

    # ---- which device path runs this forward: every predicate in ONE place -------------------------------------------
    # Paths (same results; tests assert their equality):
    #   "native_moe"     routing + encode + exchange(s) + expert FFN + exchange(s) + decode in ONE native call
    #                    (tutel_amd_moe_forward): inference with the capacity known up front, or dropless on a single rank
    #   "native_ep"      the same pipeline after a separate routing step (tutel_amd_ep_forward): batch-prioritised routing,
    #                    load-importance loss, gate noise, unequal token counts, autocast, a trainable router with frozen experts
    #   "python_overlap" the overlapped pipeline orchestrated from Python over torch.distributed (impls/overlap.py): gloo
    #                    rendezvous, or the library communicator could not be created
    #   "fused_encode"   single rank: fc1 gathers its rows from the tokens, no bucket array (what native_* does when the
    #                    native pipeline is switched off)
    #   "generic"        the reference's op sequence on the HIP ops: training (autograd), custom / fp32 experts, CPU tensors,
    #                    sharded experts, adaptive_r = 0, 2DH with overlap
    def _plan(self, stage, x, logits, gate, top_k, cf, degree, alignment, reserve_shape, inequivalent_tokens, megablocks_size,
              original_dtype, crit=None, allow_native=True):
        W = self.world_size
        fusable = (x.is_cuda and len(reserve_shape) == 1 and isinstance(self.experts, FusedExpertsNetwork) and not C.SKIP_A2A
                   and self.num_global_experts >= W and self.adaptive_degree != 0 and logits.dim() == 2
                   and (x.dtype == logits.dtype or (logits.dtype == torch.float32 and x.dtype == original_dtype))
                   and self.experts.can_fuse(x, self))
        if not fusable:
            return "generic"
        native = (allow_native and ep_native.ENABLED and not _FORCE_OVERLAP and degree <= 32 and (degree == 1 or not self.use_2dh)
                  and (W == 1 or ep_native.group_ok(self.group)))
        if stage == "before_routing":
            T, E = logits.shape
            k = min(top_k, E)
            dropless = cf <= 0
            ok = (native and ep_native.FAST_PATH and not torch.is_autocast_enabled() and not inequivalent_tokens
                  and not self.batch_prioritized_routing and self.is_gshard_loss and not (self.training and gate.gate_noise > 0)
                  and logits.dtype in (torch.float32, torch.bfloat16, torch.float16)
                  and not (torch.is_grad_enabled() and logits.requires_grad)
                  and k <= 16 and E <= 4096 and k * E <= 8192)
            if ok and dropless:   # capacity = max expert load: read back inside the native call, single rank only
                ok = W == 1 and (megablocks_size == 0 or self.is_postscore)
            elif ok:
                ok = megablocks_size == 0 and self._static_capacity(T, E, k, cf, alignment) > 0 \
                    and self._static_capacity(T, E, k, cf, alignment) % max(degree, 1) == 0
            return "native_moe" if ok else "routed"
        # after routing: crit is known
        plan_ok = isinstance(crit, RoutingPlan)
        if native and plan_ok and ep_native.usable(self, x, crit, degree):
            return "native_ep"
        if (degree > 1 and plan_ok and (W > 1 or _FORCE_OVERLAP) and crit[4] > 0 and crit[4] % degree == 0 and not self.use_2dh
                and crit.gates2d is not None):   # (a trainable router keeps its gates in crit[3] with autograd -> generic)
            return "python_overlap"
        if W == 1 and self.is_postscore and plan_ok and crit[4] > 0 and _FUSE_ENCODE:
            return "fused_encode"
        return "generic"

    @staticmethod
    def _static_capacity(T, E, k, cf, alignment):
        """capacity_factor > 0: k * int(cf * ceil(T / E)) rounded up to the alignment (fast_dispatch.py:188-199)"""
        capacity = k * int(cf * ((T + E - 1) // E))
        rem = capacity % alignment
        return capacity + ((alignment - rem) if rem > 0 else 0)

    @staticmethod
    def _native_gate_weight(gate, x):
        """wg.weight when the gate projection can run inside the native call, else None"""
        if not (_NATIVE_GATE and x.is_cuda and type(gate) is LinearTopKGate and not gate.fp32_gate and x.dim() == 2):
            return None
        w = gate.wg.weight
        if w.dtype != x.dtype or x.dtype not in (torch.bfloat16, torch.float16) or not w.is_contiguous() or w.device != x.device:
            return None
        if gate._forward_hooks or gate._forward_pre_hooks or gate.wg._forward_hooks or gate.wg._forward_pre_hooks:
            return None
        if torch.is_grad_enabled() and (w.requires_grad or x.requires_grad):
            return None
        if torch.is_autocast_enabled() or ops.gate_proj_splits(x.shape[0], x.shape[1], w.shape[0], x.dtype) == 0:
            return None
        # the split-K kernel fetches 16-byte vectors: a contiguous but offset view (a slice of a larger buffer) stays on F.linear
        # instead of raising from the forward.  Non-contiguous x is made contiguous (a fresh allocation) by the caller.
        if w.data_ptr() % 16 or (x.is_contiguous() and x.data_ptr() % 16):
            return None
        return w

    def _run_native_moe(self, x, logits, top_k, cf, degree, alignment, megablocks_size, gate_w=None):
        T, E = logits.shape
        k = min(top_k, E)
        spe = (T + E - 1) // E
        xc = x if x.is_contiguous() else x.contiguous()
        if cf <= 0:   # dropless: capacity = max expert load, read back inside the native call (fast_dispatch.py:191-199)
            guess = (k * spe * 3 // 2 + 31) // 32 * 32
            res = ep_native.forward_from_logits(self, xc, logits.contiguous(), k, guess, 1, self.normalize_gate, want_loss=True,
                                                dropless=(k * int(-cf * spe) if cf < 0 else 0, alignment), megablocks_size=megablocks_size,
                                                gate_w=gate_w)
        else:
            res = ep_native.forward_from_logits(self, xc, logits.contiguous(), k, self._static_capacity(T, E, k, cf, alignment), degree,
                                                self.normalize_gate, want_loss=True, gate_w=gate_w)
        if res is None:   # the library's communicator could not be created: every rank falls back together
            return None
        y, l_aux, cnt, capacity = res
        self.megablocks_size = megablocks_size
        self.dispatch_count = cnt
        if C.get_world_rank(self.group) == 0 and logging.getLogger().isEnabledFor(logging.INFO):
            logging.info("Capacity = %d, real-time capacity-factor for top-%d = %s", capacity, top_k, capacity / max(1, k * spe))
        return y, l_aux

    # ---- forward ------------------------------------------------------------------------
    def _degree_of_parallel_type(self, parallel_type):
        """`parallel_type` ("data" | "model" | "auto" | "adaptive:<r>") -> the initial adaptive_degree (same values and error texts as upstream)"""
        if parallel_type.startswith("adaptive:"):
            wanted = int(parallel_type.split(":", 1)[1])
            degree = sorted((0, wanted, self.sharded_count))[1]   # clamped into [0, sharded_count]
            if degree not in self.valid_rs:
                raise Exception("Unexpected value of adaptive_degree: %d, expecting a candidate within %s." % (degree, self.valid_rs))
            return degree
        if self.sharded_count == 1:
            return 1                      # nothing to split: any spelling is accepted, as upstream
        by_name = {"data": 1, "auto": 1, "model": self.sharded_count}
        if parallel_type not in by_name:
            raise Exception("Unrecognized parallel type specified: %s" % parallel_type)
        return by_name[parallel_type]

    def _spread_over_shards(self, y):
        """fewer experts than ranks: [E, C, M] buckets -> one [*, M] block per rank, each bucket replicated adaptive_degree times when the
        hidden dimension is what is split (every shard then sees every row), cut into world_size / E row blocks otherwise"""
        if self.num_global_experts >= self.world_size:
            return y
        if self.use_model_parallel:
            y = y.repeat(1, self.adaptive_degree, 1)
        return y.view(self.world_size, -1, y.size(2))

    def _collect_from_shards(self, y):
        """inverse of _spread_over_shards on the experts' outputs: the hidden-dim shards' partial products are summed"""
        if self.num_global_experts >= self.world_size:
            return y
        if self.use_model_parallel:
            return y.view(self.num_global_experts, self.adaptive_degree, -1, y.size(2)).sum(dim=1)
        return y.view(self.num_global_experts, -1, y.size(2))

    def forward(self, input, gate_index=0, capacity_factor=None, top_k=None, a2a_ffn_overlap_degree=None,
                reserve_dims=1, inequivalent_tokens=False, adaptive_r=None, megablocks_size=0):
        if self.skip_moe:
            out = input
            out.l_aux = None
            return self.result_func(out) if self.result_func is not None else out

        original_shape, original_dtype = input.shape, input.dtype
        assert len(original_shape) >= 2, "Input data must be at least 2D tensor: (s)amples, .., (m)odel_dim"
        reserve_shape = original_shape[-reserve_dims:]

        x = input.reshape(-1, reserve_shape.numel())
        if torch.is_autocast_enabled():
            x = x.to(_autocast_dtype(x))  # so the all-to-all moves low-precision bytes
        else:
            for p in self.experts.parameters():
                x = x.to(p.dtype)
                break
        if x.is_cuda and x.data_ptr() % 16:
            x = x.clone()   # an offset view of a larger buffer: the kernels fetch 16-byte vectors, upstream takes any tensor (ops._a16)
        gate = self.gates[gate_index]
        if a2a_ffn_overlap_degree is not None:
            self.a2a_ffn_overlap_degree = a2a_ffn_overlap_degree
        degree = self.a2a_ffn_overlap_degree
        top_k = top_k or gate.top_k
        cf = capacity_factor or gate.capacity_factor
        if megablocks_size > 0 and (self.num_local_experts <= 1 or torch.is_grad_enabled() or self.world_size > 1):
            megablocks_size = 0
        if adaptive_r is not None:
            self.adaptive_degree = adaptive_r

        mega = max(megablocks_size, 1)
        alignment = (self.sharded_count * degree + mega - 1) // mega * mega
        if alignment > 256:
            alignment = (alignment + 127) // 128 * 128

        def finish(y, l_aux):
            y = y.view(list(original_shape[:-reserve_dims]) + list(self.protected_shape[-reserve_dims:])).to(original_dtype)
            self.l_aux = y.l_aux = l_aux
            return self.result_func(y) if self.result_func is not None else y

        plan_args = (gate, top_k, cf, degree, alignment, reserve_shape, inequivalent_tokens, megablocks_size, original_dtype)
        # A plain 16-bit linear gate on the one-call path: the projection runs INSIDE the native call (csrc/gate_proj.hip, split-K
        # MFMA + the top-k kernel adding the partial sums) instead of F.linear here.  Decided on the logits' shape / dtype alone
        # (a meta tensor stands in for them); anything else -- hooks on the gate, a gradient, fp32_gate, an uncovered shape --
        # projects below as before.
        gate_w = self._native_gate_weight(gate, x)
        if gate_w is not None:
            spec = torch.empty([x.shape[0], gate_w.shape[0]], dtype=x.dtype, device="meta")
            if self._plan("before_routing", x, spec, *plan_args) == "native_moe":
                res = self._run_native_moe(x, spec, top_k, cf, degree, alignment, megablocks_size, gate_w=gate_w)
                if res is not None:
                    return finish(*res)
        # the gate projection: computed ONCE, autocast off (moe_layer.py:315-323), whatever path consumes it.  Where the in-call
        # projection would have applied, the other paths project with the same kernel (same partial sums, same order: same bits),
        # so that a batch is routed the same way whichever path the planner picks (bench.py's N > 1 parity canary compares the
        # native pipelines with the torch.distributed path and expects equal tokens, not merely equally valid roundings)
        logits = ops.gate_logits(x, gate_w) if gate_w is not None else None
        if logits is not None:
            pass
        elif x.is_cuda:
            with torch.autocast("cuda", enabled=False):
                logits = gate(x)
        else:
            logits = gate(x)
        if self._plan("before_routing", x, logits, *plan_args) == "native_moe":
            res = self._run_native_moe(x, logits, top_k, cf, degree, alignment, megablocks_size)
            if res is not None:
                return finish(*res)

        def routing():
            noisy = logits
            if self.training and gate.gate_noise > 0:
                noisy = logits + gate.gate_noise * torch.randn_like(logits) / self.num_global_experts
            if self.is_gshard_loss:
                loss_fn = losses.gshard_loss
            else:
                def loss_fn(scores, topk_ids):
                    return losses.load_importance_loss(F.softmax(logits, dim=1), noisy.gather(index=topk_ids, dim=1),
                                                       self.num_global_experts, gate.gate_noise)
            return logits.dtype, extract_critical(
                None, top_k=top_k, loss_fn=loss_fn, capacity_factor=cf,
                batch_prioritized_routing=self.batch_prioritized_routing, normalize_gate=self.normalize_gate,
                group=self.group, alignment=alignment, inequivalent_tokens=inequivalent_tokens, _logits=noisy)

        if x.is_cuda:
            with torch.autocast("cuda", enabled=False):
                logits_dtype, (crit, l_aux) = routing()
        else:
            logits_dtype, (crit, l_aux) = routing()

        self.megablocks_size = megablocks_size
        self.dispatch_count = get_dispatch_count(crit)
        if getattr(self, "_keep_routing", False):   # tests: the routing this forward used, element by element
            self.last_routing = (torch.stack([t.to(torch.int32) for t in crit[1]]), torch.stack([t.to(torch.int32) for t in crit[2]]))
        plan = self._plan("after_routing", x, logits, *plan_args, crit=crit)

        if plan == "native_ep":
            # the whole post-routing pipeline behind ONE native call (csrc/ep.hip): encode -> all-to-all -> expert FFN ->
            # all-to-all -> decode, `degree` stages pipelined over the caller's stream and the library's side stream
            y = ep_native.forward(self, x if x.is_contiguous() else x.contiguous(), crit, degree)
            if y is not None:
                return finish(y, l_aux)
            plan = self._plan("after_routing", x, logits, *plan_args, crit=crit, allow_native=False)   # no communicator: all ranks agree

        if plan == "python_overlap":   # overlapped expert parallelism driven from Python: one fused routine, no layout copies
            y = a2a_ffn_overlap_fused(self, x if x.is_contiguous() else x.contiguous(), crit, degree, self.is_postscore)
            return finish(y, l_aux)

        if plan == "fused_encode":
            # single rank, is_postscore: fast_encode is a pure row copy, so fc1 gathers its rows straight
            # from the tokens through the slot map -- the [E,C,M] bucket array is never materialised
            y = self.experts.forward_fused(x if x.is_contiguous() else x.contiguous(), self, R=crit[4], slot_map=crit.slot_map)
            self.protected_shape = y.shape
            return finish(fast_decode(y, crit, self.is_postscore), l_aux)

        # ---- generic: the reference's op sequence -----------------------------------------------------------------------
        # encode: with is_postscore the bucket rows are verbatim copies, so the reference's
        # round trip through logits_dtype (moe_layer.py:327) is value-preserving and skipped.
        if self.is_postscore or x.dtype == logits_dtype:
            y = fast_encode(x.contiguous(), crit, self.is_postscore)
        else:
            y = fast_encode(x.to(logits_dtype), crit, self.is_postscore).to(x.dtype)

        if self.adaptive_degree == 0:
            y = self.expert_local(y, reserve_shape)
        else:
            if self.auto_parallel:
                self.use_model_parallel = (y.numel() * (self.sharded_count - 1) * 2 < sum(p.numel() for p in self.experts.parameters()))
            y = self._spread_over_shards(y)

            fused = (isinstance(self.experts, FusedExpertsNetwork) and self.world_size > 1 and len(reserve_shape) == 1
                     and self.num_global_experts >= self.world_size and not C.SKIP_A2A and self.experts.can_fuse(y, self))
            if degree > 1 and y.is_cuda:
                # the stream pipeline carries no autograd graph: only when neither the buckets nor any
                # expert parameter can require grad (a no-grad INPUT says nothing about the experts)
                needs_autograd = torch.is_grad_enabled() and (
                    y.requires_grad or any(p.requires_grad for p in self.experts.parameters()))
                y = a2a_ffn_overlap_forward(y, expert_fn=lambda t: self.expert_local(t, reserve_shape),
                                            a2a_ffn_overlap_degree=degree, use_2dh=self.use_2dh, group=self.group,
                                            needs_autograd=needs_autograd)
            elif fused:
                y = self._experts_on_raw_a2a(y, reserve_shape)
            else:
                y = C.all_to_all(y, 1, 0, use_2dh=self.use_2dh, group=self.group)
                y = self.expert_local(y, reserve_shape)
                y = C.all_to_all(y, 0, 1, use_2dh=self.use_2dh, group=self.group)

            y = self._collect_from_shards(y)

        # decode: the kernel multiplies/accumulates in fp32 and rounds ONCE to its output dtype.  With an
        # fp32 gate and low-precision experts the reference casts the buckets to fp32, combines in fp32 and
        # rounds the result to the input dtype (moe_layer.py:359-361) -- the same single rounding -- so when
        # the layer's output dtype equals the bucket dtype the fp32 round trip is skipped (bit-identical).
        if y.dtype == logits_dtype or (logits_dtype == torch.float32 and y.dtype == original_dtype):
            y = fast_decode(y if y.is_contiguous() else y.contiguous(), crit, self.is_postscore)
        else:
            y = fast_decode(y.to(logits_dtype), crit, self.is_postscore)
        return finish(y, l_aux)


moe_layer = MOELayer
