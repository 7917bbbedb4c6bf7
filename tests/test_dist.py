"""Multi-GPU host path on CPU: world_size 2 over gloo.  The per-rank device MSM and the device
combine are stubbed with the CPU oracle (this is a test, the oracle is the checker); what is under
test is blaze_amd.multi_gpu: the shard partition, the single all-gather of fixed-size partials in
rank order, and that every rank ends with identical bytes equal to the unsharded result."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, curve, n, pf, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import oracle
    from blaze_amd.multi_gpu import shard_range, sharded_msm

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pts, sc, expected = oracle.input_generator(curve, n, pf, 4242)   # same on every rank
        pb = oracle.point_bytes(curve) * pf
        lo, hi = shard_range(n, rank, world)
        partial = oracle.msm_pippenger(curve, bytes(pts[lo * pb: hi * pb]), bytes(sc[lo * 32: hi * 32]), hi - lo, pf, threads=2)

        def combine(partials, count):  # what MSMClient.combine_partials does on the GPU
            rb = oracle.result_bytes(curve)
            acc = None
            for i in range(count):
                xy, on = oracle.decode_result(curve, partials[i * rb: (i + 1) * rb])
                assert on
                acc = oracle.point_add(curve, acc, xy)
            return oracle.result_from_affine(curve, acc)

        full = sharded_msm(partial, combine, dist)
        q.put((rank, full == expected, full.hex()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("curve,n,pf", [("BLS381", 777, 1), ("BN254", 300, 8)])
def test_sharded_msm_world2_gloo(curve, n, pf):
    import torch.multiprocessing as mp

    import oracle

    oracle.build()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, curve, n, pf, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert len({h for _, _, h in res}) == 1          # every rank holds the same normalised bytes


def test_shard_range_partitions_exactly():
    from blaze_amd.multi_gpu import shard_range

    for n in (0, 1, 7, 8, 1 << 26, (1 << 26) + 5):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
