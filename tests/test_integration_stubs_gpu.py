"""INTEGRATION.md section B documents the ctypes stubs a maintainer of UPSTREAM Tutel would add to bind its native seams to the C ABI
(jit_compiler.py / sparse.py dispatch kernels, gating.py cumsum + extract_critical, ffn.py expert GEMMs).  This test extracts those
code blocks VERBATIM from the markdown, executes them, and drives encode / decode / gate-grad / cumsum / routing / expert GEMM
through them against the oracle -- so the documented binding cannot rot.  Reference seams:
tutel/impls/fast_dispatch.py:16-134, tutel/impls/jit_compiler.py:26-52, tutel/jit_kernels/gating.py:19-24, tutel/experts/ffn.py:114-120."""
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blocks(section):
    """the ```python blocks under the heading that starts with '### <section>' (up to the next '### ')"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"^### " + re.escape(section) + r".*?$(.*?)(?=^### |\Z)", text, re.S | re.M)
    assert m, section
    return re.findall(r"```python\n(.*?)```", m.group(1), re.S)


@pytest.fixture(scope="module")
def ns():
    from tutel_amd import _lib
    _lib.lib()   # builds nothing; makes sure the library is there (and that torch's HIP runtime is the one mapped)
    env = {}
    src = _blocks("B.1")[0].replace("/path/to/libtutel_amd.so", _lib.LIB_PATH)
    assert "/path/to" not in src
    exec(compile(src, "INTEGRATION.md:B.1", "exec"), env)
    return env


def test_b1_dispatch_stub(ns, oracle):
    T, M, E, k = 700, 96, 12, 2
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        g = torch.Generator().manual_seed(3)
        scores = torch.softmax(torch.randn([T, E], generator=g), dim=1)
        crit, _ = oracle.extract_critical(scores, k, 0.5)                 # capacity factor 0.5: some tokens are dropped
        C = crit[4]
        x = torch.randn([T, M], generator=g).to(dtype)
        buf = torch.randn([E * C, M], generator=g).to(dtype)
        gates = [gt.to(dtype) for gt in crit[3]]
        d = ns["AmdDispatch"]([i.cuda() for i in crit[1]], [l.cuda() for l in crit[2]], [gt.cuda() for gt in gates], E, C)
        crit_t = (crit[0], crit[1], crit[2], gates, C, crit[5])
        for use_gates in (False, True):
            enc = d.forward(x.cuda(), use_gates)
            assert torch.equal(enc.cpu().view(E, C, M), oracle.fast_encode(x, crit_t, is_postscore=not use_gates).view(E, C, M))
            dec = d.backward_data(buf.cuda(), use_gates)
            assert torch.equal(dec.cpu(), oracle.fast_decode(buf.view(E, C, M), crit_t, is_postscore=use_gates))
        gg = d.backward_gate(x.cuda(), buf.cuda()).cpu()
        want = torch.stack([oracle.gate_grad(x, buf, crit[1][j], crit[2][j], C) for j in range(k)])
        assert torch.allclose(gg, want, rtol=1e-4, atol=1e-3 * float(want.abs().max()))
        torch.cuda.synchronize()


def test_b2_routing_stubs(ns, oracle):
    blocks = _blocks("B.2")
    env = dict(ns)
    exec(compile(blocks[0], "INTEGRATION.md:B.2[0]", "exec"), env)
    g = torch.Generator().manual_seed(5)
    mask = (torch.rand([1000, 24], generator=g) < 0.2).to(torch.int64)
    got = env["fast_cumsum_sub_one"](mask.cuda())
    assert torch.equal(got.cpu().to(torch.int64), torch.cumsum(mask, dim=0) - 1)
    for T, E, k, cf, dtype in ((4096, 64, 2, 1.0, torch.float32), (513, 16, 1, 0.5, torch.float32), (2000, 128, 4, 1.0, torch.float64)):
        scores = torch.softmax(torch.randn([T, E], generator=g, dtype=torch.float64), dim=1).to(dtype)
        crit, l_ref = oracle.extract_critical(scores, k, cf)
        if dtype == torch.float64:
            continue   # (the stub's _DT table lists the three storage dtypes of the dispatch kernels)
        env.update(T=T, E=E, k=k, dev=torch.device("cuda"), scores=scores.cuda(), capacity=crit[4], normalize_gate=True)
        exec(compile(blocks[1], "INTEGRATION.md:B.2[1]", "exec"), env)
        torch.cuda.synchronize()
        assert torch.equal(env["idx"].cpu(), torch.stack(crit[1]).to(torch.int32)) and torch.equal(env["loc"].cpu(), torch.stack(crit[2]).to(torch.int32))
        assert torch.equal(env["cnt"].cpu(), crit[5].to(torch.int32))
        assert torch.equal(env["gates"].cpu(), torch.stack(crit[3]))
        assert abs(float(env["l_aux"]) - float(l_ref)) < 1e-5
        sm = env["slot_map"].cpu().view(E, crit[4])
        filled = sm >= 0
        assert int(filled.sum()) == int(sum(int(((crit[1][j] >= 0) & (crit[2][j] < crit[4])).sum()) for j in range(k)))


def test_b3_expert_gemm_stub(ns, oracle):
    env = dict(ns)
    exec(compile(_blocks("B.3")[0], "INTEGRATION.md:B.3[0]", "exec"), env)
    gemm = env["_gemm"]
    g = torch.Generator().manual_seed(9)
    for dtype in (torch.bfloat16, torch.float16):
        E_loc, R, K, H = 3, 160, 128, 192
        x = torch.randn([E_loc, R, K], generator=g).to(dtype)
        w1 = (torch.randn([E_loc, H, K], generator=g) / K ** 0.5).to(dtype)      # batched_fc1_w [E, H, M]: k-major
        b1 = torch.randn([E_loc, H], generator=g).to(dtype)
        w2 = (torch.randn([E_loc, H, K], generator=g) / H ** 0.5).to(dtype)      # batched_fc2_w [E, H, M_out]: n-major
        b2 = torch.randn([E_loc, K], generator=g).to(dtype)
        h = gemm(x.cuda(), w1.cuda(), b1.cuda(), True, 1)
        y = gemm(h, w2.cuda(), b2.cuda(), False, 0)
        want = oracle.expert_ffn(x, w1, b1, w2, b2, accum_fp32=True)
        err = (y.cpu().double() - want.double()).abs()
        assert bool((err <= 2 ** -6 * want.double().abs() + 2 ** -7 * float(want.double().abs().max())).all()), float(err.max())
        # megablocks row counts (ffn.py:70-81): rows past ceil(count / align) * align are left untouched
        counts = torch.tensor([160, 37, 0], dtype=torch.int32)
        y2 = torch.full_like(y, 7.0)
        h2 = gemm(x.cuda(), w1.cuda(), b1.cuda(), True, 1, counts.cuda(), 4)
        for e, c in enumerate([160, 40, 0]):
            assert torch.equal(h2[e, :c], h[e, :c])


def test_b3_expert_ffn_stub(ns):
    """B.3's opt-in one-launch FFN stub (tutel_amd_expert_ffn), executed verbatim from the markdown: 1001 (ENOTSUP, nothing launched) while
    TUTEL_OPT_FFN_FUSED is at its default, and with the option on the same bits as the two `_gemm` calls it stands for."""
    from tutel_amd import _lib, ops
    blocks = _blocks("B.3")
    env = dict(ns)
    exec(compile(blocks[0], "INTEGRATION.md:B.3[0]", "exec"), env)
    stub = [b for b in blocks if "tutel_amd_expert_ffn(" in b]
    assert len(stub) == 1
    g = torch.Generator().manual_seed(21)
    dtype = torch.bfloat16
    E_loc, R, M, H, M_out = 64, 96, 512, 1024, 1024
    x = torch.randn([E_loc, R, M], generator=g).to(dtype).cuda()
    w1 = (torch.randn([E_loc, H, M], generator=g) / M ** 0.5).to(dtype).cuda()
    w2t = (torch.randn([E_loc, M_out, H], generator=g) / H ** 0.5).to(dtype).cuda()
    b1, b2 = torch.randn([E_loc, H], generator=g).to(dtype).cuda(), torch.randn([E_loc, M_out], generator=g).to(dtype).cuda()
    env.update(x=x, w1=w1, b1=b1, w2t=w2t, b2=b2, E_loc=E_loc, R=R, M=M, H=H, M_out=M_out)
    want = env["_gemm"](env["_gemm"](x, w1, b1, True, 1), w2t, b2, True, 0)
    exec(compile(stub[0], "INTEGRATION.md:B.3[ffn]", "exec"), env)
    assert env["rc"] == _lib.ENOTSUP, "opt-in: the default answers ENOTSUP"
    try:
        ops.set_option(_lib.OPT_FFN_FUSED, 1)
        exec(compile(stub[0], "INTEGRATION.md:B.3[ffn]", "exec"), env)
    finally:
        ops.set_option(_lib.OPT_FFN_FUSED, -1)
    torch.cuda.synchronize()
    assert env["rc"] == 0 and torch.equal(env["y"], want)


def test_b6_gate_projection_stub(ns, oracle):
    """B.6: the gate projection of a 16-bit gate through the split-K kernel + top-k on its partial sums, executed verbatim from the
    markdown: the logits it returns equal x @ wg^T to an ulp of the dtype, and idx / gates equal the oracle's routing on those logits."""
    from tutel_amd import ops
    blocks = _blocks("B.6")
    T, M, E, k = 1500, 512, 32, 2
    for dtype in (torch.bfloat16, torch.float16):
        g = torch.Generator().manual_seed(13)
        x = torch.randn([T, M], generator=g).to(dtype).cuda()
        gate = torch.nn.Module()
        gate.wg = torch.nn.Linear(M, E, bias=False).to(dtype).cuda()
        env = dict(ns)
        env.update(T=T, M=M, E=E, k=k, dev=torch.device("cuda"), x=x, gate=gate, capacity=k * ((T + E - 1) // E), normalize_gate=True)
        exec(compile(blocks[0], "INTEGRATION.md:B.6[0]", "exec"), env)
        torch.cuda.synchronize()
        assert env["S"] >= 1
        with torch.no_grad():
            ref = x.float() @ gate.wg.weight.float().t()
        ulp = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
        assert float((env["logits"].float() - ref).abs().max()) <= ulp * max(1.0, float(ref.abs().max()))
        scores = ops.gate_topk(env["logits"], k, apply_softmax=True, want_scores=True)[3].cpu()
        crit, _ = oracle.extract_critical(scores, k, 1.0)
        assert torch.equal(env["idx"].cpu(), torch.stack(crit[1]).to(torch.int32))
        assert torch.equal(env["gates"].cpu(), torch.stack(crit[3]).to(dtype))
