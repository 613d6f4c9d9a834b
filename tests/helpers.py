"""Shared comparison helpers for the parity tests."""
import torch


def topk_equivalent(idx, scores, ref_idx, ref_scores, full_ref_scores=None, tol=2e-6):
    """Tolerance-aware top-k comparison.

    Returns (n_exact_rows, n_rows, ok).  A position may differ from the oracle only if the
    oracle's own scores of the two rows are within ``tol`` (a near-tie that fp32 summation order
    can flip); scores must agree to 1e-5 everywhere.
    """
    idx = idx.cpu().long(); ref_idx = ref_idx.cpu().long()
    scores = scores.cpu(); ref_scores = ref_scores.cpu()
    ok = bool(torch.allclose(scores, ref_scores, atol=1e-5, rtol=0))
    exact = (idx == ref_idx).all(dim=1)
    for q in torch.nonzero(~exact).flatten().tolist():
        for p in torch.nonzero(idx[q] != ref_idx[q]).flatten().tolist():
            if full_ref_scores is not None:
                a = full_ref_scores[q, idx[q, p]]
            else:
                a = scores[q, p]
            if abs(float(a) - float(ref_scores[q, p])) > tol:
                ok = False
    return int(exact.sum()), idx.shape[0], ok


# ---- parity counts: a tracked number, not a `-s` print --------------------------------
# Tests record how many queries of a shape are index-exact against the oracle; the counts are written to
# gpurun_out/parity_counts.json (merged back by gpurun; copied to profiles/ for the record) and printed in
# pytest's terminal summary (tests/conftest.py), so they also appear in the driver's log tail.
PARITY = {}


def record_parity(shape: str, exact: int, n: int, score_near_ties: int = 0, probe_near_ties: int = 0, **extra):
    import json
    import os
    PARITY[shape] = dict(exact=int(exact), n=int(n), score_near_ties=int(score_near_ties),
                         probe_near_ties_excluded=int(probe_near_ties), **extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.environ.get("AURA_PARITY_JSON", os.path.join(root, "gpurun_out", "parity_counts.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        old = {}
        if os.path.exists(path):
            with open(path) as f:
                old = json.load(f)
        old.update(PARITY)
        with open(path, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass
