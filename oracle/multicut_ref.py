"""Exact multicut for SMALL graphs by enumeration.  TEST INFRASTRUCTURE ONLY (tests import it to bound the product's approximate solver).

The reference merges slice objects with ``elf.segmentation.multicut.multicut_decomposition`` (``micro_sam/multi_dimensional_segmentation.py:364-373``;
python-elf / nifty are not vendored and absent here - parity against the library itself is UNPINNED).  What can be pinned without the
library is the OBJECTIVE it minimises (nifty's multicut objective: the sum of the costs of the cut edges, positive = attractive): this
module enumerates every partition of up to ~10 nodes and returns an optimal one, so that tests can state how far the product's solver
(GAEC + Kernighan-Lin style refinement) is from the optimum and that it is exact where the structure guarantees it (forests)."""
from __future__ import annotations

from typing import Tuple

import numpy as np


def _partitions(n: int):
    """Restricted growth strings: every partition of n items exactly once."""
    lab = [0] * n

    def rec(i, m):
        if i == n:
            yield tuple(lab)
            return
        for c in range(m + 1):
            lab[i] = c
            yield from rec(i + 1, max(m, c + 1))
    if n == 0:
        yield ()
        return
    yield from rec(1, 1)


def energy(uv_ids: np.ndarray, costs: np.ndarray, labels) -> float:
    labels = np.asarray(labels)
    uv = np.asarray(uv_ids, dtype=np.int64)
    return float(np.asarray(costs, dtype=np.float64)[labels[uv[:, 0]] != labels[uv[:, 1]]].sum())


def optimal_multicut(n_nodes: int, uv_ids: np.ndarray, costs: np.ndarray) -> Tuple[np.ndarray, float]:
    """(labels, energy) of a minimum-energy partition; partitions that split a cluster into disconnected parts are covered as well (they
    are never better than the partition with the parts separated, which is enumerated too)."""
    assert n_nodes <= 11, "enumeration is for small graphs"
    best, best_e = None, float("inf")
    for lab in _partitions(n_nodes):
        e = energy(uv_ids, costs, lab)
        if e < best_e - 1e-12:
            best, best_e = np.array(lab, dtype=np.int64), e
    return best, best_e
