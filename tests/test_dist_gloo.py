"""Multi-rank sharding protocol on CPU (gloo, world_size 2 and 3): the block-range cut (dhts_shard_cut, the same host
arithmetic dhts_bam_set_shard uses) and the 8-byte hand-off check between adjacent shards.  The per-shard "decode" is
stood in by the oracle here (no GPU in this container); on the GPU the same protocol is exercised by
tests/test_gpu_bam.py::test_read_bam_sharded_concat."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_records, break_it, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import duckhts_amd
        import orc
        from duckhts_amd import synth
        arr, _ = synth.bam_segment(n_records, seed=11, threads=1)
        data = arr.tobytes()
        z = orc.bgzf_inflate_all(data)
        r = orc.bam_read(data)
        uoff = np.concatenate([[0], np.cumsum(z["ulen"].astype(np.int64))])
        b0, b1 = duckhts_amd.shard_cut(z["coff"].astype(np.uint64), len(data), rank, world)
        lo, hi = uoff[b0], uoff[b1]
        starts = r["rec_off"]
        sel = np.nonzero((starts >= lo) & (starts < hi))[0]
        # a shard owns the records that START inside its block range; its last record may end in the next shard's blocks
        first = int(starts[sel[0]]) if len(sel) else int(hi)
        last_end = int(starts[sel[-1] + 1]) if len(sel) and sel[-1] + 1 < len(starts) else int(len(z["data"]))
        if break_it and rank == 0:
            last_end += 1
        span = torch.tensor([first, last_end, len(sel)], dtype=torch.int64)
        got = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(got, span)
        spans = [tuple(int(x) for x in g) for g in got]
        try:
            total = duckhts_amd.check_handoff(spans)
            ok = (total == r["n_rows"]) and spans[0][0] == r["first_rec_off"]
        except RuntimeError:
            ok = False
        q.put((rank, ok, b0, b1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_handoff_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 30000, False, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _, _ in res)
    # ranges tile the block list exactly
    assert res[0][2] == 0 and all(res[i][3] == res[i + 1][2] for i in range(world - 1))


def test_shard_handoff_detects_break():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 20000, True, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert not any(ok for _, ok, _, _ in res)


def test_shard_cut_properties():
    import duckhts_amd
    coff = np.cumsum(np.r_[0, np.random.RandomState(0).randint(100, 60000, 999)]).astype(np.uint64)
    total = int(coff[-1]) + 5000
    for world in (1, 2, 3, 8, 64):
        cuts = [duckhts_amd.shard_cut(coff, total, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == len(coff)
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        sizes = [int(coff[b1 - 1]) - int(coff[b0]) if b1 > b0 else 0 for b0, b1 in cuts]
        if world <= 8:
            assert max(sizes) - min(sizes) < 2 * 60000 + total // world // 10
