"""the minimal two-process reproducer for the stale-row observation of round 4 (write-through peer stores
returned rows of the batch in between with two rank processes on one GPU; a one-process probe was clean).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
        tools/r5_two_process_visibility.py [out.txt]

Two (or W) rank processes share cuda:0, rendezvous over gloo and attach the IPC transport exactly as the product does (flag
segment in uncached memory, data segment from hipMalloc mapped with hipIpcOpenMemHandle, signal / wait kernels, epochs).  Then
tutel_amd_ep_ipc_selfcheck (csrc/ep.hip) runs writer process -> flag kernel -> wait kernel -> reader kernel on `MB` MiB per peer,
PASSES passes back to back with NO host synchronisation, the payload changing every pass, for every store flavour
(plain | sc1 | sc0 sc1 | nt) x both queue layouts of the pipeline (reader on the writer's stream | on a side stream of another
priority), REPEATS times.  A vector that still carries an earlier pass' pattern is a stale read; the reader counts them on the
device.  The epoch canaries (one guard of the product path) are checked by the wait kernels on the way: a flag that overtook
its own kernel's stores shows up there as an error naming the rank.

Prints (rank 0) one line per (flavour, layout) and a verdict; the same text goes to the file given as argv[1]."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FLAVOURS = ["plain", "sc1", "sc0 sc1 (write-through)", "nt"]


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    share = os.environ.get("VIS_ONE_GPU", "1") == "1" or torch.cuda.device_count() < world
    dev = torch.device("cuda", 0 if share else rank)
    torch.cuda.set_device(dev)
    from tutel_amd import _lib
    from tutel_amd.impls import ep_native as EN
    mb = int(os.environ.get("MB", 32))
    passes = int(os.environ.get("PASSES", 16))
    repeats = int(os.environ.get("REPEATS", 3))
    lines = []

    def say(s):
        if rank == 0:
            print(s, flush=True)
            lines.append(s)

    def fresh_comm():
        EN.set_transport("ipc")
        c = EN.communicator(None, dev)
        assert c is not None and c.ipc, "the IPC transport did not attach (self-check failed?)"
        return c, EN._open_segment(c, world * (mb << 20), False)

    t0 = time.time()
    comm, seg = fresh_comm()
    say(f"# two-process visibility probe: {world} rank processes on {'ONE device (cuda:0)' if share else 'one device each'}, "
        f"{mb} MiB per peer ({world * mb} MiB per segment), {passes} passes back to back, {repeats} repeats per cell")
    say(f"# attach self-check of the product path (plain stores, {comm.selfcheck}) passed in {time.time() - t0:.2f} s")
    say(f"# {'flavour':28s} {'reader stream':14s} {'stale vectors':>14s} {'of':>12s} {'first (rank, vector)':>22s} {'ms / pass':>10s}  note")
    total_bad, cells = 0, 0
    for fl, name in enumerate(FLAVOURS):
        for side in (False, True):
            bad_sum, first_any, note, dt = 0, None, "", 0.0
            for _ in range(repeats):
                dist.barrier()
                t1 = time.time()
                try:
                    bad, first = EN.ipc_selfcheck(comm, seg, mb << 20, passes, fl, side)
                except _lib.TutelAmdError as ex:      # the canary guard (or a time-out) tripped: say so, start over with a new communicator
                    bad, first = -1, 0
                    note = str(ex)[:160]
                dt += time.time() - t1
                ok = torch.tensor([0 if bad < 0 else 1])
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok) == 0:
                    note = note or "a peer reported an error"
                    comm, seg = fresh_comm()
                    bad = max(bad, 0)
                    bad_sum = -1
                    break
                bad_sum += bad
                if bad and first_any is None:
                    first_any = (first >> 40, first & ((1 << 40) - 1))
            tb = torch.tensor([bad_sum])
            dist.all_reduce(tb, op=dist.ReduceOp.SUM if bad_sum >= 0 else dist.ReduceOp.MIN)
            n_vec = repeats * passes * world * (mb << 20) // 16 * world
            say(f"  {name:28s} {'side' if side else 'same':14s} {int(tb):14d} {n_vec:12d} {str(first_any):>22s} {dt / (repeats * passes) * 1e3:10.3f}  {note}")
            cells += 1
            total_bad += abs(int(tb))
    say(f"# verdict: {'no stale vector and no canary / time-out error in any cell' if total_bad == 0 else 'STALE READS OR ERRORS SEEN -- see the cells above'} "
        f"({cells} cells, {time.time() - t0:.1f} s)")
    EN.destroy_all()
    if rank == 0 and len(sys.argv) > 1:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
        open(sys.argv[1], "w").write("\n".join(lines) + "\n")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
