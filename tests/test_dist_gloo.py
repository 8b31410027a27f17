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


def _file_worker(rank, world, port, path, q):
    """one file, `world` ranks: every rank derives its byte window on its own (dhts_shard_window probes the file for a BGZF block
    start), then the ranks exchange (first record, end of last record) as BGZF VIRTUAL offsets -- their inflated-stream numbering
    differs because each one holds only header + window.  The decode of a window is stood in by the oracle."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import duckhts_amd
        import orc
        data = open(path, "rb").read()
        z = orc.bgzf_inflate_all(data)
        r = orc.bam_read(data)
        coff = z["coff"].astype(np.int64)
        uoff = np.concatenate([[0], np.cumsum(z["ulen"].astype(np.int64))])
        k = int(np.searchsorted(uoff, r["first_rec_off"], side="right")) - 1      # block holding the first record
        header_bytes = int(coff[k])                                               # blocks entirely in front of the first record's block
        wb, we, own_end = duckhts_amd.shard_window(path, rank, world, header_bytes)
        assert rank == 0 or wb in set(coff.tolist()) or wb == len(data), "a window starts on a block start"
        # blocks this rank owns: those that START in [wb, own_end); the window with its halo holds the end of its last record
        owned = np.nonzero((coff >= wb) & (coff < (own_end if rank < world - 1 else len(data))))[0]
        lo = uoff[owned[0]] if len(owned) else 0
        hi = uoff[owned[-1] + 1] if len(owned) else 0
        starts = r["rec_off"]
        sel = np.nonzero((starts >= lo) & (starts < hi))[0]

        def voff(u):                                                              # bgzf_tell of inflated position u
            b = min(int(np.searchsorted(uoff, u, side="right")) - 1, len(coff) - 1)
            return (int(coff[b]) << 16) | int(u - uoff[b])
        ends = np.append(starts[1:], len(z["data"]))                              # end of record i = start of record i + 1
        first = voff(int(starts[sel[0]])) if len(sel) else -1
        last_end = voff(int(ends[sel[-1]])) if len(sel) else -1
        halo_ok = (not len(sel)) or int(coff[min(int(np.searchsorted(uoff, int(ends[sel[-1]]) - 1, side="right")) - 1, len(coff) - 1)]) < we
        span = torch.tensor([first, last_end, len(sel), wb, we, own_end, int(halo_ok)], dtype=torch.int64)
        got = [torch.zeros(7, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(got, span)
        rows = [tuple(int(x) for x in g) for g in got]
        spans = [(a, b, n) for a, b, n, *_ in rows if n > 0]
        ok = duckhts_amd.check_handoff(spans) == r["n_rows"] and all(x[6] for x in rows)
        ok = ok and rows[0][3] == 0 and all(rows[i][5] <= rows[i + 1][3] or rows[i + 1][2] == 0 for i in range(world - 1))   # ownership ranges are disjoint and ordered
        q.put((rank, bool(ok), rows[rank][2]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_one_file_byte_range_cut_gloo(world, tmp_path):
    """one file, N ranks -- the byte-range cut of dhts_open_path_shard and the virtual-offset hand-off"""
    from duckhts_amd import synth
    arr, _ = synth.bam_segment(40000, seed=17, threads=1)
    path = os.path.join(str(tmp_path), "one.bam")
    arr.tofile(path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_file_worker, args=(r, world, port, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert all(n > 0 for _, _, n in res), res                                    # every rank got work


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
