"""S-scores and retrieval metrics from the ranks of the true matches.

``cmp`` / ``cmp_all`` mirror the reference's ``parse_results.py:4-35`` (same arguments, same
printed line).  ``rank1`` and ``mean_average_precision`` are the two figures BASELINE.json names;
with one relevant gallery item per query (dataloader.py:94-99) they are functions of the same
rank vector: rank-1 = mean(rank == 1), mAP = mean(1 / rank).
"""

from __future__ import annotations

from typing import Sequence


def cmp(rankings: Sequence[int], p: int, total_shoeprints: int, total_shoemarks: int) -> float:
    """Share of all queries whose true match lies within the top ``p`` percent of the gallery."""
    cutoff = (p * total_shoeprints) / 100
    hits = 0
    for rank in rankings:
        if rank <= cutoff:
            hits += 1
    return hits / total_shoemarks


def s_scores(rankings: Sequence[int], total_shoeprints: int, total_shoemarks: int) -> dict[str, float]:
    return {f"S{p}": cmp(rankings, p, total_shoeprints, total_shoemarks) * 100 for p in (1, 5, 10, 15, 20)}


def format_s_scores(rankings: Sequence[int], total_shoeprints: int, total_shoemarks: int) -> str:
    return " ".join(f"{k}:{v:.2f}" for k, v in s_scores(rankings, total_shoeprints, total_shoemarks).items())


def cmp_all(rankings: Sequence[int], total_shoeprints: int, total_shoemarks: int) -> None:
    """Print ``S1:.. S5:.. S10:.. S15:.. S20:..`` with two decimals (parse_results.py:27-35)."""
    print(format_s_scores(rankings, total_shoeprints, total_shoemarks))


def rank1(rankings: Sequence[int]) -> float:
    return sum(1 for r in rankings if r == 1) / max(1, len(rankings))


def mean_average_precision(rankings: Sequence[int]) -> float:
    return sum(1.0 / r for r in rankings) / max(1, len(rankings))
