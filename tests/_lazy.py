"""Comparisons that hold on every collision path of the library (include/rp_amd.h: rp_last_path)."""
import numpy as np


def lazy_relaxed(status, cost, orun, ctx, out, lo=0):
    """Plans that ran the cost-ordered collision stage (rp_last_path() == 1, list rounds, or 3, bounded sweep; the reference's own order of work,
    reactive_planner.py:1031-1062) label the colliding candidates they had to look at: at least every one that sorts before the
    winner.  Returns the device labels with the others marked as the eager query would, and their number -- after checking
    that each of them really sorts behind the winner (and that there is a winner: without one every candidate was looked at)."""
    if ctx is None or out is None or ctx.last_path() not in (1, 3):
        return status, 0
    unl = ((orun.status & 3) == 3) & ((status & 3) == 1)
    if unl.any():
        assert out.best_index >= 0
        idx = lo + np.flatnonzero(unl)
        assert np.all((cost[unl] > out.best_cost) | ((cost[unl] == out.best_cost) & (idx > out.best_index)))
    fixed = status.copy()
    fixed[unl] = (fixed[unl] & ~np.uint32(3)) | np.uint32(3)
    return fixed, int(unl.sum())
