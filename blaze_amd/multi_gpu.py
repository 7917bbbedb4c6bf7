"""Multi-GPU MSM: shard by element chunk and / or scalar chunk, one process per GPU, one tiny exchange.

MSM is a sum, so rank g of G takes elements [g n/G, (g+1) n/G) - its slice of the scalars and of
the points (`shard_range`) - or, since round 3, a slice of the scalars' BITS of a larger
element chunk (`shard_layout`: the library picks the mix by the window planner's cost; a rank's partial
result then carries the weight 2^bit_lo of its range, and the partials still simply add up).  Each rank runs
the full single-GPU pipeline through its own MSMClient; the G partial results (144 / 96 bytes each) are exchanged with ONE all-gather (RCCL over
xGMI when the process group backend is "nccl"; gloo in the CPU tests) and every rank adds them in
rank order on its device (`MSMClient.combine_partials`), so every rank returns identical, normalised
bytes.  RCCL's reduce ops are arithmetic, not a group law, hence all-gather + local add instead of
all-reduce.  The reference has no multi-device layer (README.md:20-22 leaves it to a "management
layer"); this module is that layer for the MSM path only."""
from __future__ import annotations

from typing import Tuple


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition of [0, n): first n % world ranks get one extra element."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_layout(curve, n: int, rank: int, world: int) -> dict:
    """The split the library picks for `world` equal devices (include/blaze_hip.h blz_msm_shard_layout): rank's element chunk
    [first, first + count) and scalar range [bit_lo, bit_hi) - hand the latter to MSMClient.set_scalar_range."""
    import ctypes as C

    from ._lib import check, lib

    out = (C.c_uint32 * 4)()
    check(lib().blz_msm_shard_layout(int(curve), n, world, rank, out))
    return {"first": int(out[0]), "count": int(out[1]), "bit_lo": int(out[2]), "bit_hi": int(out[3])}


SHARD_SCALARS_FROM_HOST = 1   # include/blaze_hip.h BLZ_SHARD_*: the scalars cross the rank's PCIe link with every task
SHARD_BASES_FROM_HOST = 2     # ... and the bases too (DMA flow)


def _layout_dict(out) -> dict:
    return {"first": int(out[0]), "count": int(out[1]), "bit_lo": int(out[2]), "bit_hi": int(out[3]), "ranges": int(out[4]),
            "est_compute_ms": out[5] / 1e3, "est_link_ms": out[6] / 1e3, "device_mib": int(out[7])}


def shard_layout_ex(curve, n: int, rank: int, world: int, flags: int = 0, ranges: int | None = None) -> dict:
    """blz_msm_shard_layout_ex: the split with the flow's transfers priced in (flags: SHARD_*); `ranges` asks for the
    estimates of one candidate (R scalar ranges) instead of the library's pick."""
    import ctypes as C

    from ._lib import check, lib

    out = (C.c_uint32 * 8)()
    if ranges is None:
        check(lib().blz_msm_shard_layout_ex(int(curve), n, world, rank, flags, out))
    else:
        check(lib().blz_msm_shard_layout_candidate(int(curve), n, world, rank, flags, ranges, out))
    return _layout_dict(out)


def all_gather_partials(partial: bytes, dist, device=None) -> bytes:
    """One all-gather of the fixed-size partial results, returned concatenated in rank order."""
    import torch

    world = dist.get_world_size()
    t = torch.frombuffer(bytearray(partial), dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    out = torch.empty(world * t.numel(), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy().tobytes()


def sharded_msm(local_partial: bytes, combine, dist, device=None) -> bytes:
    """`local_partial`: this rank's MSM result bytes over its shard; `combine(partials, count)`:
    rank-ordered group sum (MSMClient.combine_partials on the GPU).  Returns the full result."""
    world = dist.get_world_size()
    gathered = all_gather_partials(local_partial, dist, device)
    return combine(gathered, world)
