"""Fused per-expert FFN (reference: tutel/experts/ffn.py, FusedExpertsNetwork).

Parameters keep the reference's names and shapes (checkpoint compatible):
    batched_fc1_w    [E_loc, H/s, M]        batched_fc1_bias [E_loc, H/s]
    batched_fc2_w    [E_loc, H/s, M_out]    batched_fc2_bias [E_loc, ceil(M_out/s)]

Forward, y = act(x @ W1^T + b1) @ W2 + b2 per local expert, x [E_loc, R, M]:
  * bf16 / fp16, no autograd, recognised activation -> the MFMA grouped GEMM kernels: two launches of
    tutel_amd_expert_gemm -- bias + activation fused into the first, bias into the second, dropless row counts honoured
    on device (or, opt-in with TUTEL_AMD_FFN_FUSED=1, ONE persistent launch for fc1 -> activation -> fc2,
    tutel_amd_expert_ffn: bit-identical, measured slower at the headline shape).  W2 is stored [H, M_out] (checkpoint format); the kernel takes that layout as is
    (training-mode modules) or, in eval mode, a k-major copy laid out once (KMajorCache below);
  * the gathered-weight modes -- adaptive_r = 0 (every rank's experts all-gathered, ffn.py:83-89) and sharded experts
    (num_local_experts < 0, ffn.py:91-109) -- run on the SAME MFMA grouped GEMM, fed the gathered [E, H', M] tensors
    (forward(), below the gathers);
  * only fp32 / fp64 experts, weight gradients, and arbitrary python activations stay on ATen batched matmul (rocBLAS /
    hipBLASLt), op for op as the reference.
"""
import os

import torch
import torch.nn.functional as F

from .. import net, ops

# eval-mode weight pre-layout: keep a [E, N, K] (k-major) copy of weights stored [E, K, N]; 0 disables
_PREPACK = int(os.environ.get("TUTEL_AMD_PREPACK", "1")) != 0
# training: forward + data gradients of the bf16 / fp16 ReLU FFN on the MFMA grouped GEMM (weight gradients stay on ATen); 0 disables
_TRAIN_FUSED = int(os.environ.get("TUTEL_AMD_TRAIN_FUSED", "1")) != 0


def _FFN_FUSED_ON():
    from .. import _lib
    return ops.get_option(_lib.OPT_FFN_FUSED) > 0


class _FFNTrain(torch.autograd.Function):
    """y = relu(x @ W1^T + b1) @ W2 + b2 per expert WITH autograd, on the grouped GEMM kernels (round 3).

    forward : two launches (bias + ReLU fused into the first, bias into the second); the hidden activation is kept for backward.
    backward: d hid = (gy @ W2^T) * [hid > 0] -- one launch, the ReLU mask rides in the GEMM's gating epilogue (W2 [E, H, M_out] is
              the k-major operand of that product as stored); d x = d hid @ W1 -- one launch (W1 [E, H, M] is its [K, N] operand as
              stored, consumed through the transposing LDS read).  The weight gradients contract over the ROW dimension
              (hid^T @ gy, d hid^T @ x): library batched GEMMs, as in the reference (ffn.py:114-120 under autograd)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        hid = ops.expert_gemm(x, w1, b1, True, act="relu")
        y = ops.expert_gemm(hid, w2, b2, False)
        ctx.save_for_backward(x, w1, w2, hid)
        ctx.has_bias = (b1 is not None, b2 is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w1, w2, hid = ctx.saved_tensors
        gy = gy.contiguous()
        need_x, need_w1, need_b1, need_w2, need_b2 = ctx.needs_input_grad
        gx = gw1 = gb1 = gw2 = gb2 = None
        if need_w2:
            gw2 = torch.matmul(hid.transpose(1, 2), gy)
        if need_b2 and ctx.has_bias[1]:
            gb2 = gy.sum(dim=1)
        if need_x or need_w1 or (need_b1 and ctx.has_bias[0]):
            ghid = ops.expert_gemm(gy, w2, None, True, mul=(hid > 0).to(hid.dtype))
            if need_x:
                gx = ops.expert_gemm(ghid, w1, None, False)
            if need_w1:
                gw1 = torch.matmul(ghid.transpose(1, 2), x)
            if need_b1 and ctx.has_bias[0]:
                gb1 = ghid.sum(dim=1)
        return gx, gw1, gb1, gw2, gb2


class KMajorCache:
    """k-major ([E, N, K] contiguous) copies of weights the checkpoint format stores [E, K, N].

    The MFMA operand layout wants 8 consecutive k per lane: a k-major tile is read from LDS with one
    ds_read_b128 per fragment, an n-major tile needs two transposing ds_read_b64_tr_b16 -- measured
    +3 us of 118 at the headline shape and +20 % at >= 256 rows per expert.  Inference weights are
    static, so the fused (no-autograd, eval) path lays them out once.  A copy is rebuilt whenever the
    parameter's storage pointer, autograd version counter (bumped by every in-place op on the
    parameter: optimizer steps, copy_, load_state_dict), dtype, device or shape changes.  Writes
    made through `param.data` bypass the version counter: call `invalidate()` (or
    `module.train()`; or set TUTEL_AMD_PREPACK=0) after such a write."""

    def __init__(self):
        self._store = {}

    def get(self, name, w, transpose=True, dtype=None):
        """cached derived copy of parameter `w`: transposed to k-major ([E, K, N] -> [E, N, K]) and / or cast to `dtype`
        (autocast: fp32 master weights, low-precision compute copies)"""
        try:
            version = w._version
        except RuntimeError:  # inference tensors (module built under torch.inference_mode()) carry no version counter
            version = -1
        key = (w.data_ptr(), version, w.dtype, w.device, tuple(w.shape), bool(transpose), dtype)
        hit = self._store.get(name)
        if hit is None or hit[0] != key:
            t = w.detach()
            if dtype is not None and t.dtype != dtype:
                t = t.to(dtype)
            t = t.transpose(1, 2).contiguous() if transpose else t.contiguous()
            hit = (key, t)
            self._store[name] = hit
        return hit[1]

    def invalidate(self):
        self._store.clear()


def _probe_points():
    """fp32 probe: dense around the origin, then every magnitude a bf16 / fp16 / fp32 hidden activation can take."""
    lin = torch.linspace(-16.0, 16.0, 2049)
    mag = torch.logspace(-30, 38, 545)
    big = torch.tensor([60000.0, 65504.0, 1e5, 3.3e38])
    return torch.cat([lin, mag, -mag, big, -big, torch.zeros(1)])


_PROBE = _probe_points()
_REFS = (("relu", F.relu), ("gelu", F.gelu), ("silu", F.silu), ("none", lambda t: t))
# callables that ARE the fused epilogues (identity match, no evaluation needed)
_KNOWN_FNS = {F.relu: "relu", torch.relu: "relu", F.gelu: "gelu", F.silu: "silu"}


def classify_activation(fn, dtype=None):
    """Name of the fused GEMM epilogue that computes `fn`, or None (-> ATen path, the function is called).

    Recognised by identity: a string name, an explicit `_tutel_amd_act` tag, F.relu / F.gelu / F.silu
    themselves and default-argument nn.ReLU / nn.GELU / nn.SiLU modules.  Any other callable (typically
    the `lambda x: F.relu(x)` of the reference's examples, helloworld.py:86) is evaluated on a wide probe
    -- 2049 points on [-16, 16] plus every decade from 1e-30 to 3.3e38 in both signs, in fp32 AND in
    the experts' working dtype, twice (stochastic functions are rejected) -- and fused only when it
    matches an epilogue on every point: clipped variants (relu6, hardtanh, clamp) differ there."""
    if isinstance(fn, str):
        return fn if fn in ops.ACT_CODES else None
    tag = getattr(fn, "_tutel_amd_act", None)
    if tag in ops.ACT_CODES:
        return tag
    try:
        if fn in _KNOWN_FNS:
            return _KNOWN_FNS[fn]
    except TypeError:
        pass
    if isinstance(fn, torch.nn.Module):
        if type(fn) is torch.nn.ReLU:
            return "relu"
        if type(fn) is torch.nn.SiLU:
            return "silu"
        if type(fn) is torch.nn.GELU and getattr(fn, "approximate", "none") == "none":
            return "gelu"
        if type(fn) is torch.nn.Identity:
            return "none"
    try:
        with torch.no_grad():
            found = None
            for dt in ([torch.float32] + ([dtype] if dtype not in (None, torch.float32) else [])):
                lim = torch.finfo(dt).max
                probe = _PROBE.clamp(-lim, lim).to(dt)
                a, b = fn(probe.clone()), fn(probe.clone())
                if not (torch.is_tensor(a) and a.shape == probe.shape and a.dtype == dt and torch.equal(a, b)):
                    return None
                name = None
                for cand, ref in _REFS:
                    r = ref(probe)
                    if torch.allclose(a.float(), r.float(), rtol=0, atol=1e-7 if dt == torch.float32 else 0.0, equal_nan=True):
                        name = cand
                        break
                if name is None or (found is not None and name != found):
                    return None
                found = name
            return found
    except Exception:
        pass
    return None


class FusedExpertsNetwork(torch.nn.Module):
    def __init__(self, model_dim, hidden_size_per_expert, num_experts_per_device, sharded_count,
                 activation_fn=None, activation_fn_with_self=None, output_dim=None,
                 has_fc1_bias=True, has_fc2_bias=True):
        super().__init__()
        self.skip_expert = int(os.environ.get("SKIP_EXPERT", "0")) != 0
        assert hidden_size_per_expert % sharded_count == 0, \
            f"Can't evenly divide hidden_size_per_expert ({hidden_size_per_expert}) to {sharded_count} slices."
        self.model_dim = model_dim
        self.hidden_size_per_expert = hidden_size_per_expert
        self.local_experts = num_experts_per_device
        self.sharded_count = sharded_count
        self.hidden_size = hidden_size_per_expert // sharded_count
        self.output_dim = output_dim or model_dim

        self._act_opaque = False
        if activation_fn_with_self is not None:
            assert activation_fn is None, "Option `activation_fn_with_self` has been specified, please keep exactly one of them."
            tag = getattr(activation_fn_with_self, "_tutel_amd_act", None)
            activation_fn = lambda x: activation_fn_with_self(x, self)  # noqa: E731
            if tag is not None:
                activation_fn._tutel_amd_act = tag
            else:
                # a function of (x, module) may read module state or have side effects (fairseq passes
                # dropout + layernorm here): never evaluated on a probe, never replaced by an epilogue
                self._act_opaque = True
        if activation_fn is None:
            activation_fn = F.relu
        self.activation_fn = activation_fn
        self._act_cache = {}
        self._kmajor = KMajorCache()

        E, H = num_experts_per_device, self.hidden_size
        self.batched_fc1_w = torch.nn.Parameter(torch.empty(E, H, model_dim))
        self.batched_fc2_w = torch.nn.Parameter(torch.empty(E, H, self.output_dim))
        if has_fc1_bias:
            self.batched_fc1_bias = torch.nn.Parameter(torch.empty(E, H))
        else:
            self.register_parameter("batched_fc1_bias", None)
        if has_fc2_bias:
            self.batched_fc2_bias = torch.nn.Parameter(torch.empty(E, (self.output_dim + sharded_count - 1) // sharded_count))
        else:
            self.register_parameter("batched_fc2_bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        # per-expert nn.Linear initialisation, in the same RNG order as the reference
        # (ffn.py:39-49) so that seeded runs produce the same weights.
        with torch.no_grad():
            for i in range(self.batched_fc1_w.size(0)):
                fc1 = torch.nn.Linear(self.model_dim, self.hidden_size, bias=self.batched_fc1_bias is not None)
                fc2 = torch.nn.Linear(self.hidden_size, self.output_dim, bias=self.batched_fc2_bias is not None)
                self.batched_fc1_w[i] = fc1.weight
                self.batched_fc2_w[i] = fc2.weight.t()
                if self.batched_fc1_bias is not None:
                    self.batched_fc1_bias[i] = fc1.bias
                if self.batched_fc2_bias is not None:
                    self.batched_fc2_bias[i] = fc2.bias[:self.batched_fc2_bias.size(-1)]

    def extra_repr(self):
        return "model_dim=%d, hidden_size=%d, output_dim=%d, num_experts_per_device=%d. has_fc1_bias=%s, has_fc2_bias=%s." % (
            self.batched_fc1_w.size(2), self.batched_fc1_w.size(1), self.batched_fc2_w.size(2), self.batched_fc1_w.size(0),
            self.batched_fc1_bias is not None, self.batched_fc2_bias is not None)

    # -- fused path -------------------------------------------------------------------------
    def fused_activation(self):
        if self._act_opaque:
            return None
        key = (self.training, self.batched_fc1_w.dtype)
        if key not in self._act_cache:
            self._act_cache[key] = classify_activation(self.activation_fn, key[1])
        return self._act_cache[key]

    def _no_autograd(self, x):
        return not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())))

    def compute_dtype(self, x):
        """dtype the MFMA GEMMs would run in for input x, or None.  Same dtype as the weights; or, under autocast,
        low-precision x against fp32 master weights -- ATen's matmul autocasts those (the reference relies on it,
        moe_layer.py:26-39,265-266, examples/helloworld_amp.py:76-79), the grouped GEMM uses cached casts instead."""
        w = self.batched_fc1_w
        if x.dtype not in (torch.bfloat16, torch.float16):
            return None
        if w.dtype == x.dtype:
            return x.dtype
        if torch.is_autocast_enabled() and w.dtype == torch.float32 and x.dtype == torch.get_autocast_dtype(x.device.type):
            return x.dtype
        return None

    def can_fuse(self, x, ctx):
        """the plain fused route: this rank's own local experts (no weight gathering), no autograd, MFMA-able shapes.  (False does NOT
        mean ATen: adaptive_r = 0 / sharded experts gather their weights in forward() and run the same MFMA kernel on the result.)"""
        if self.skip_expert or not x.is_cuda or not self._no_autograd(x):
            return False
        if getattr(ctx, "adaptive_degree", 1) == 0 or getattr(ctx, "sharded_count", 1) > 1:
            return False
        w1, w2 = self.batched_fc1_w, self.batched_fc2_w
        dt = self.compute_dtype(x)
        return (dt is not None and ops.gemm_supported(dt, w1.size(1), w1.size(2)) and
                ops.gemm_supported(dt, w2.size(2), w2.size(1)) and self.fused_activation() is not None)

    def can_fuse_training(self, x, ctx):
        """the same local experts WITH autograd: bf16 / fp16 (or autocast over fp32 master weights), ReLU, all four GEMM
        products MFMA-able (forward K = M, H; backward K = M_out, H)"""
        if (not _TRAIN_FUSED or self.skip_expert or not x.is_cuda or x.dim() != 3 or self._no_autograd(x)
                or getattr(ctx, "adaptive_degree", 1) == 0 or getattr(ctx, "sharded_count", 1) > 1):
            return False
        w1, w2 = self.batched_fc1_w, self.batched_fc2_w
        dt = self.compute_dtype(x)
        b2 = self.batched_fc2_bias
        return (dt is not None and self.fused_activation() == "relu" and x.size(0) == w1.size(0) and (b2 is None or b2.size(-1) == self.output_dim)
                and ops.gemm_supported(dt, w1.size(1), w1.size(2)) and ops.gemm_supported(dt, w2.size(2), w2.size(1))
                and ops.gemm_supported(dt, w2.size(1), w2.size(2)) and ops.gemm_supported(dt, w1.size(2), w1.size(1)))

    def fused_params(self, dtype):
        """(w1, b1, w2, b2, w2_kmajor) of this rank's experts in the GEMM's compute dtype: the parameters themselves, or
        cached casts under autocast; fc2 as a k-major copy in eval mode."""
        w1, b1, w2, b2 = self.batched_fc1_w, self.batched_fc1_bias, self.batched_fc2_w, self.batched_fc2_bias
        cast = None if w1.dtype == dtype else dtype
        kmajor = self.w2_kmajor_now()
        if cast is not None:
            w1 = self._kmajor.get("fc1.cast", w1, transpose=False, dtype=cast)
            b1 = self._kmajor.get("b1.cast", b1, transpose=False, dtype=cast) if b1 is not None else None
        if kmajor or cast is not None:
            w2 = self._kmajor.get("fc2", w2, transpose=kmajor, dtype=cast)
        if b2 is not None:
            if b2.size(-1) != self.output_dim or cast is not None:
                b2 = self._kmajor.get("b2.cast", b2[:, :self.output_dim], transpose=False, dtype=cast)
        return w1, b1, w2, b2, kmajor

    def forward_fused(self, x, ctx, a_layout=None, R=None, out=None, d_layout=None, slot_map=None, expert_range=None):
        """x [E_loc,R,M] (or the raw all-to-all buffer described by a_layout) -> [E_loc,R,M_out]
        (or written into `out` in d_layout).  Two MFMA grouped-GEMM launches.
        slot_map given: x is the TOKEN array [T,M] and fc1 gathers its rows through the slot map
        (fast_encode fused into the GEMM; R = capacity).
        expert_range = (lo, hi): only local experts lo..hi-1 (x / out then hold hi-lo experts)."""
        counts, align = None, 1
        if getattr(ctx, "megablocks_size", 0) > 0:
            counts, align = ctx.dispatch_count, int(ctx.megablocks_size)
        w1, b1, w2, b2, w2_kmajor = self.fused_params(self.compute_dtype(x))
        if expert_range is not None:
            lo, hi = expert_range
            w1, w2 = w1[lo:hi], w2[lo:hi]
            b1 = b1[lo:hi] if b1 is not None else None
            b2 = b2[lo:hi] if b2 is not None else None
            assert counts is None, "megablocks row counts are not sliced"
        # TUTEL_OPT_FFN_FUSED = 1 (opt-in; the two launches below measured faster): one persistent launch for fc1 -> activation -> fc2
        # where the library takes the shape (<= 128 rows per expert, k-major fc2, plain [E_loc, R, *] operands) -- same tiles, same bits
        if _FFN_FUSED_ON() and w2_kmajor and counts is None and out is None and a_layout is None and d_layout is None and (slot_map is not None or x.dim() == 3):
            y = ops.expert_ffn(x, w1, b1, w2, b2, self.fused_activation(), R=R, smap=slot_map)
            if y is not None:
                return y
        if slot_map is not None:
            h = ops.expert_gemm_gather(x, slot_map, w1, b1, True, self.fused_activation(), R,
                                       row_counts=counts, row_align=align)
        else:
            h = ops.expert_gemm(x, w1, b1, True, act=self.fused_activation(), E_loc=w1.size(0), R=R,
                                a_layout=a_layout, row_counts=counts, row_align=align)
        return ops.expert_gemm(h, w2, b2, w2_kmajor, out=out, d_layout=d_layout, row_counts=counts, row_align=align)

    def w2_kmajor_now(self):
        """eval-mode modules run fc2 on a k-major copy of its weights (KMajorCache)"""
        return _PREPACK and not self.training

    def invalidate_prepacked(self):
        """Drop the eval-mode k-major weight copies (needed only after writes through `param.data`)."""
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

    # -- gathered-weight modes + the reference-equivalent ATen path ------------------------------
    def forward(self, x, ctx):
        if self.skip_expert:
            return x
        if self.can_fuse(x, ctx):
            return self.forward_fused(x.contiguous(), ctx)

        if self.can_fuse_training(x, ctx):
            dt = self.compute_dtype(x)
            b1, b2 = self.batched_fc1_bias, self.batched_fc2_bias
            return _FFNTrain.apply(x.contiguous(), self.batched_fc1_w.to(dt), b1.to(dt) if b1 is not None else None,
                                   self.batched_fc2_w.to(dt), b2.to(dt) if b2 is not None else None)

        w1, w2 = self.batched_fc1_w, self.batched_fc2_w
        b1 = self.batched_fc1_bias.unsqueeze(1) if self.batched_fc1_bias is not None else None
        b2 = self.batched_fc2_bias.unsqueeze(1) if self.batched_fc2_bias is not None else None

        if ctx.adaptive_degree == 0:  # data-parallel experts: gather every rank's weights (ffn.py:83-89)
            E = ctx.num_global_experts
            w1 = net.zero_gather(w1, group=ctx.group).view(E, -1, w1.size(2))
            w2 = net.zero_gather(w2, group=ctx.group).view(E, -1, w2.size(2))
            if b1 is not None:
                b1 = net.zero_gather(b1, group=ctx.group).view(E, 1, -1)
            if b2 is not None:
                b2 = net.zero_gather(b2, group=ctx.group).view(E, 1, -1)
        elif ctx.sharded_count > 1:  # one expert's hidden dim sliced over ranks (ffn.py:91-109)
            mesh = net.get_world_size(ctx.group)
            if 1 < mesh < net.get_world_size():
                ctx.adaptive_degree = ctx.sharded_count
            group_size = ctx.sharded_count // ctx.adaptive_degree
            if group_size > 1:
                zg = net.create_groups_from_world(group_count=-group_size, parent_group=ctx.group).model_group
                w1 = net.zero_gather(w1, group=zg).view(1, -1, ctx.model_dim)
                w2 = net.zero_gather(w2, group=zg).view(1, -1, self.output_dim)
                if b1 is not None:
                    b1 = net.zero_gather(b1, group=zg).view(1, 1, -1)
            if b2 is not None:
                bg = net.create_groups_from_world(group_count=ctx.num_global_experts, parent_group=ctx.group).model_group
                b2 = net.zero_gather(b2, group=bg).view(1, 1, -1)
                if ctx.adaptive_degree > 1:
                    b2 = b2 * (1.0 / ctx.adaptive_degree)
        if b2 is not None and b2.size(-1) != self.output_dim:
            b2 = b2[:, :, :self.output_dim]

        # the gathered / sliced weights on the MFMA grouped GEMM as well (adaptive_r = 0 and sharded experts used to leave
        # the hand-written path): weights as they come out of the gather -- fc2 in its [E, H', M_out] checkpoint layout
        dt = self.compute_dtype(x)
        if (dt is not None and x.is_cuda and self._no_autograd(x) and self.fused_activation() is not None and x.dim() == 3
                and x.size(0) == w1.size(0) and ops.gemm_supported(dt, w1.size(1), w1.size(2))
                and ops.gemm_supported(dt, w2.size(2), w2.size(1))):
            def prep(t):
                return None if t is None else t.detach().to(dt).contiguous()
            b1f = prep(b1.squeeze(1)) if b1 is not None else None
            b2f = prep(b2.squeeze(1)) if b2 is not None else None
            h = ops.expert_gemm(x.contiguous(), prep(w1), b1f, True, act=self.fused_activation())
            return ops.expert_gemm(h, prep(w2), b2f, False)

        y = torch.matmul(x, w1.permute(0, 2, 1))
        if b1 is not None:
            y = y + b1
        y = self.activation_fn(y)
        y = torch.matmul(y, w2)
        if b2 is not None:
            y = y + b2
        return y


ExpertModule = FusedExpertsNetwork
