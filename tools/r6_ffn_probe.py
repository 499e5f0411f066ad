"""round 6: where the time of the persistent fc1 -> fc2 launch goes (per-workgroup tick sums, csrc/expert_ffn.hip FfnArgs::dbg)"""
import ctypes, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tutel_amd import _lib, ops
L = _lib.lib()
L.tutel_amd_expert_ffn_debug.argtypes = [ctypes.c_void_p]
E, R, M, H = 64, 128, 2048, 2048
g = torch.Generator().manual_seed(1)
x = torch.randn([E, R, M], generator=g).bfloat16().cuda()
w1 = ((torch.rand([E, H, M], generator=g) * 2 - 1) / 16).bfloat16().cuda()
w2 = ((torch.rand([E, M, H], generator=g) * 2 - 1) / 16).bfloat16().cuda()
b1, b2 = torch.randn([E, H], generator=g).bfloat16().cuda(), torch.randn([E, M], generator=g).bfloat16().cuda()
mode = int(os.environ.get("MODE", "-1"))
ops.set_option(_lib.OPT_FFN_FUSED, mode)
dbg = torch.zeros([256 * 24], dtype=torch.int64, device="cuda")
for _ in range(5):
    ops.expert_ffn(x, w1, b1, w2, b2, "relu")
torch.cuda.synchronize()
L.tutel_amd_expert_ffn_debug(ctypes.c_void_p(dbg.data_ptr()))
ops.expert_ffn(x, w1, b1, w2, b2, "relu")
torch.cuda.synchronize()
L.tutel_amd_expert_ffn_debug(None)
tl = dbg[256 * 8:].view(256, 8, 2).cpu()
d = dbg[:256 * 8].view(256, 8).cpu().double()
t0 = int(tl[:, 0, 0].min())
out = {}
out['timeline_us(item start, end) of WG 0, 1, 8, 100, 255'] = {b: [[round((int(tl[b, i, 0]) - t0) / 100, 1), round((int(tl[b, i, 1]) - t0) / 100, 1)] for i in range(4)] for b in (0, 1, 8, 100, 255)}
out['item_durations_mean_us'] = [round(float((tl[:, i, 1] - tl[:, i, 0]).double().mean()) / 100, 2) for i in range(4)]
out['item_start_spread_us'] = [round(float((tl[:, i, 0].max() - tl[:, i, 0].min())) / 100, 2) for i in range(4)]
names = ["ticket", "poll", "fc1_tile", "publish", "fc2_tile", "total", "items", "queue"]
for i, n in enumerate(names[:6]):
    out[n] = dict(mean_us=float(d[:, i].mean()) / 100, max_us=float(d[:, i].max()) / 100, min_us=float(d[:, i].min()) / 100)
out['total_by_queue_mean_us'] = [round(float(d[d[:, 7] == q][:, 5].mean()) / 100, 1) for q in range(8)]
out['last_item_end_by_queue_us(max)'] = [round((int(tl[(d[:, 7] == q)][:, 3, 1].max()) - t0) / 100, 1) for q in range(8)]
out['last_item_end_us'] = dict(min=round((int(tl[:, 3, 1].min()) - t0) / 100, 1), max=round((int(tl[:, 3, 1].max()) - t0) / 100, 1))
out["items_per_wg"] = dict(mean=float(d[:, 6].mean()), max=float(d[:, 6].max()), min=float(d[:, 6].min()))
out["queues"] = [int((d[:, 7] == q).sum()) for q in range(8)]
# timing: fused vs two launches, alternating
def t(fn, n=30):
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
two = lambda: ops.expert_gemm(ops.expert_gemm(x, w1, b1, True, act="relu"), w2, b2, True)
ops.set_option(_lib.OPT_FFN_FUSED, 0); two(); ops.set_option(_lib.OPT_FFN_FUSED, mode)
res = []
for _ in range(3):
    res.append((t(lambda: ops.expert_ffn(x, w1, b1, w2, b2, "relu")), t(two)))
out["us_fused_vs_two_launches"] = res
print(json.dumps(out, indent=1))

# ---- the same through the layer's one-call path (fused location inside the fc1 items), as bench.py runs it
if os.environ.get("LAYER", "1") == "1":
    import bench
    layer = bench.build_layer(2048, 2048, 64, 2, 0, 1, torch.bfloat16, False, 1.0).cuda().eval()
    torch.manual_seed(0)
    xx = torch.randn([16, 256, 2048]).bfloat16().cuda()
    with torch.no_grad():
        for _ in range(5):
            layer(xx)
        torch.cuda.synchronize()
        dbg.zero_()
        L.tutel_amd_expert_ffn_debug(ctypes.c_void_p(dbg.data_ptr()))
        layer(xx)
        torch.cuda.synchronize()
        L.tutel_amd_expert_ffn_debug(None)
    tl = dbg[256 * 8:].view(256, 8, 2).cpu()
    d = dbg[:256 * 8].view(256, 8).cpu().double()
    t0 = int(tl[:, 0, 0].min())
    o2 = {n: round(float(d[:, i].mean()) / 100, 2) for i, n in enumerate(names[:6])}
    o2["total_max"] = float(d[:, 5].max()) / 100
    o2["item_durations_mean_us"] = [round(float((tl[:, i, 1] - tl[:, i, 0]).double().mean()) / 100, 2) for i in range(4)]
    o2["last_item_end_us"] = dict(min=round((int(tl[:, 3, 1].min()) - t0) / 100, 1), max=round((int(tl[:, 3, 1].max()) - t0) / 100, 1))
    print("LAYER", json.dumps(o2))
