"""Gallery sharding over the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards naturally: pairs are independent, so rank r owns the contiguous
gallery slice ``shard_bounds(G, world, r)``, scores every query against it with no data-path
communication, and ONE all-gather of the float32 score blocks gives every rank the [Q, G]
matrix the ranker needs.  The payload is tiny (Q*G/world*4 bytes per rank) and latency-bound.
The reference has no counterpart: it fans query chunks out to CPU processes over a shared-memory
gallery (similarity.py:146-197).
"""

from __future__ import annotations

import os


def shard_bounds(n_items: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous slice [start, end) of rank ``rank``; the first ``n_items % world`` ranks hold one
    more item (the same rule the reference uses for its query chunks, similarity.py:146-157)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def init_from_env(backend: str | None = None):
    """Initialise the default process group from torchrun's environment; returns (rank, world, local_rank).
    A single process without the variables runs as world 1 without a process group."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gather_score_blocks(local_scores, n_gallery: int, group=None):
    """All-gather the per-rank [Q, G_r] float32 blocks into the full [Q, G] matrix on every rank.

    Shards may differ by one item; blocks are padded to the widest shard for the collective
    (all_gather_into_tensor needs equal sizes) and the padding is dropped when the blocks are laid
    side by side.  Works on CUDA tensors (RCCL) and CPU tensors (gloo)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_scores
    world = dist.get_world_size(group)
    nq = local_scores.shape[0]
    widths = [shard_bounds(n_gallery, world, r) for r in range(world)]
    wmax = max(e - s for s, e in widths)
    send = local_scores
    if send.shape[1] != wmax:
        send = torch.zeros((nq, wmax), dtype=local_scores.dtype, device=local_scores.device)
        send[:, : local_scores.shape[1]] = local_scores
    send = send.contiguous()
    flat = torch.empty((world * nq, wmax), dtype=send.dtype, device=send.device)  # concatenated form: gloo and RCCL
    dist.all_gather_into_tensor(flat, send, group=group)
    recv = flat.view(world, nq, wmax)
    if all(e - s == wmax for s, e in widths):
        return recv.permute(1, 0, 2).reshape(nq, world * wmax).contiguous()
    return torch.cat([recv[r, :, : e - s] for r, (s, e) in enumerate(widths)], dim=1).contiguous()


def matrix_digest(matrix) -> int:
    """64-bit position-weighted checksum of a float32 matrix's bit patterns (computed where the matrix lives)."""
    import torch

    bits = matrix.contiguous().view(torch.int32).reshape(-1).to(torch.int64)
    weight = torch.arange(bits.numel(), dtype=torch.int64, device=bits.device) % 1000003 + 1
    return int((bits * weight).sum().item())


def gather_digests(digest: int, device=None, group=None) -> list[int]:
    """Every rank's digest on every rank (all_gather of one int64 each)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        return [digest]
    mine = torch.tensor([digest], dtype=torch.int64, device=device if device is not None else "cpu")
    out = torch.empty((dist.get_world_size(group),), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return [int(v) for v in out.tolist()]


def barrier(group=None):
    import torch.distributed as dist

    if dist.is_initialized():
        dist.barrier(group=group)


def max_over_ranks(value: float, device=None, group=None) -> float:
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
