"""Image partition across ranks and the final gather (torch.distributed plumbing only).

The path shards by pixels: rank r of P renders the 8-row bands b with b % P == r (scene replicated, RNG a pure
function of (seed, pixel, sample) so any partition reproduces the single-GPU image bit for bit).  The only
exchange is one gather of the HDR band buffers to rank 0 per render — RCCL over xGMI on GPUs (backend "nccl"),
gloo in the CPU tests.  The reference has no multi-device code; its closest relative is the Embree backend's
16x16 tile pool (src/headless/EmbreeHeadlessRenderer.mm:2538-2542, 3166-3178).
"""
from __future__ import annotations

from typing import List, Optional

BAND_ROWS = 8   # PTR_BAND_ROWS of include/ptr_abi.h


def band_count(height: int, part: int, parts: int) -> int:
    """Number of BAND_ROWS-row bands owned by `part` (same rule as ptr_part_band_count in the C-ABI)."""
    bands = (height + BAND_ROWS - 1) // BAND_ROWS
    if parts <= 0 or part >= parts or bands <= part:
        return 0
    return (bands - part + parts - 1) // parts


def max_band_count(height: int, parts: int) -> int:
    return band_count(height, 0, parts)


def gather_bands(local, height: int, rank: int, world: int, group=None):
    """Gather every rank's [bands*BAND_ROWS, W, 3] buffer on rank 0 and interleave the bands into the image.

    `local` must be padded to max_band_count(height, world)*BAND_ROWS rows so all ranks send equal blocks.
    Returns the [height, W, 3] image on rank 0, None elsewhere.
    """
    import torch
    import torch.distributed as dist

    if world == 1:
        return local[:height]
    rows = max_band_count(height, world) * BAND_ROWS
    assert local.shape[0] == rows, (local.shape, rows)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()   # CPU rehearsal backend: stage through host memory
    if rank == 0:
        recv: Optional[List] = [torch.empty_like(local) for _ in range(world)]
    else:
        recv = None
    dist.gather(local, gather_list=recv, dst=0, group=group)
    if rank != 0:
        return None
    return assemble(recv, height)


def assemble(parts_out, height: int):
    """Interleave per-rank band buffers: band b of rank p is image band p + b*P."""
    import torch

    world = len(parts_out)
    width = parts_out[0].shape[1]
    total_bands = (height + BAND_ROWS - 1) // BAND_ROWS
    stacked = torch.stack([p.reshape(-1, BAND_ROWS, width, 3) for p in parts_out], dim=1)  # [b, P, BAND_ROWS, W, 3]
    img = stacked.reshape(-1, BAND_ROWS, width, 3)[:total_bands]                            # band index = b*P + p
    return img.reshape(-1, width, 3)[:height]
