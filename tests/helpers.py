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
