"""Dev tool: the expert-parallel test of tests/test_ep_ranks_one_gpu.py with EIGHT ranks sharing one GPU
(too slow for the suite: eight processes import torch at once).  `python tools/ep8_one_gpu_check.py ipc` runs the IPC transport
(round 4: eight rank processes mapping each other's segments, peer stores, flag kernels with eight peers; each case also checks
bit-equality with the hosted exchange) -- the world size of the metric's 8-GPU point."""
import os
import sys

import torch.multiprocessing as mp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_ep_ranks_one_gpu as T  # noqa: E402


def main():
    cases = ((2, 2, True), (1, 1, True), (2, 1, True), (2, 8, True), (2, 2, False))
    if len(sys.argv) > 1 and sys.argv[1] == "ipc":
        cases = ((2, 8, "ipc"), (1, 8, "ipc"), (2, 1, "ipc")) if len(sys.argv) < 3 else ((int(sys.argv[2]), int(sys.argv[3]), "ipc"),)   # (2, 8): 64 global experts, degree 2 = the bench's N = 8 configuration
    for degree, E_loc, native in cases:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = T._free_port()
        procs = [ctx.Process(target=T._worker, args=(r, 8, port, degree, E_loc, q, None, native)) for r in range(8)]
        with T._rank_env(8):   # two hardware queues per process: eight default-sized processes oversubscribe the device's queue slots
            for p in procs:
                p.start()
        res = [q.get(timeout=600) for _ in procs]
        for p in procs:
            p.join(timeout=60)
        for r in sorted(res):
            if not r[1] and "Connection closed by peer" not in str(r[2]) and "hosted exchange callback failed" not in str(r[2]):
                print("rank", r[0], "FAILED:\n" + str(r[2]), flush=True)
        print("world 8, degree", degree, "E_loc", E_loc, ("IPC transport" if native == "ipc" else "native one-call pipeline") if native else "python-orchestrated", "->", all(r[1] for r in res), sorted(set(r[2][:60] for r in res))[:2], flush=True)


if __name__ == "__main__":
    main()
