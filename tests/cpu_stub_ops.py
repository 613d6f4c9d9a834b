"""CPU stand-in for ``aura_snn_rag_amd.ops`` built from the oracle -- TEST INFRASTRUCTURE ONLY.

Lets the ``-m "not gpu"`` suite exercise the product's HOST logic (slot planning, batched-write
splitting at rebuild boundaries, id maps, candidate fallback, sharded merge plumbing) without a
GPU.  It is injected by monkeypatching inside tests; the product never imports it."""
import torch
import torch.nn.functional as F

from oracle import aura_oracle as O


KNN_FLAG_NO_CANDIDATES, KNN_FLAG_LISTS_STALE = 64, 128


class AuraDeviceError(RuntimeError):
    pass


def bank_row_norms(bank, inv_norm, row0, n):
    inv_norm[row0:row0 + n] = 1.0 / bank[row0:row0 + n].norm(dim=1).clamp_min(1e-12)


def bank_write(bank, loc, meta, inv_norm, feats, slots, cur_loc, now, centroids=None,
               centroid_counts=None, eff_k=0, distinct_slots=False, serial=False):
    for i, slot in enumerate(slots.tolist()):
        f = feats[i]
        bank[slot] = f
        loc[slot] = cur_loc
        meta[slot] = torch.tensor([1.0, now, -1.0, 0.0])
        inv_norm[slot] = 1.0 / f.norm().clamp_min(1e-12)
        if centroids is not None:
            d = torch.norm(centroids[:eff_k] - f, dim=1)
            c = int(torch.argmin(d))
            centroid_counts[c] += 1
            eta = 1.0 / centroid_counts[c].clamp(min=1.0)
            centroids[c] = (1 - eta) * centroids[c] + eta * f
            meta[slot, 2] = c


def bank_decay(meta, rate, count):
    meta[:count, 0] *= (1.0 - rate)


def knn_search(bank, inv_norm, meta, queries, k, now, count=None, loc=None, q_loc=None, idx_base=0,
               force_dense=False, centroids=None, nprobe=0, check_overflow=True, fp32_scan=False, shadow=None,
               rho=None, return_flag=False):
    N = bank.shape[0] if count is None else count
    nq = queries.shape[0]
    scores = torch.full((nq, k), float("-inf"))
    idx = torch.full((nq, k), -1, dtype=torch.int32)
    m_norm = F.normalize(bank[:N], dim=1)
    temporal = torch.exp(-(now - meta[:N, 1]) / 3600.0)
    for i in range(nq):
        sim = torch.mm(F.normalize(queries[i:i + 1], dim=1), m_norm.t()).squeeze(0)
        sp = torch.zeros_like(sim)
        if q_loc is not None:
            sp = 1.0 / (1.0 + torch.norm(loc[:N] - q_loc[i], dim=1))
        comb = (0.5 * sim + 0.3 * sp + 0.2 * temporal) * meta[:N, 0]
        if centroids is not None:
            cd = torch.norm(centroids - queries[i], dim=1)
            top = torch.topk(-cd, k=nprobe).indices
            mask = torch.zeros(N, dtype=torch.bool)
            for c in top:
                mask |= meta[:N, 2] == c
            comb = torch.where(mask, comb, torch.full_like(comb, float("-inf")))
        kk = min(k, int((comb > float("-inf")).sum()))
        if kk:
            s, p = torch.topk(comb, kk)
            scores[i, :kk], idx[i, :kk] = s, (p + idx_base).to(torch.int32)
    return (scores, idx, torch.zeros(1, dtype=torch.int32)) if return_flag else (scores, idx)


def topk_merge(scores, idx, k):
    S, nq, _ = scores.shape
    fs = scores.permute(1, 0, 2).reshape(nq, S * k)
    fi = idx.permute(1, 0, 2).reshape(nq, S * k).long()
    s, i = O.merge_topk(fs, fi, k)
    return s, i.to(torch.int32)


def bank_gather(bank, idx):
    out = torch.zeros(*idx.shape, bank.shape[1])
    ok = idx >= 0
    out[ok] = bank[idx[ok].long()]
    return out


def kmeans_assign(bank, centroids, count, k):
    return torch.argmin(torch.cdist(bank[:count], centroids[:k]), dim=1).to(torch.int32)


def group_by_cluster(assign, k=256):
    a = assign.to(torch.int32)
    order = torch.sort(a, stable=True).indices.to(torch.int32)
    valid = a >= 0
    lens = torch.bincount(a.clamp(min=0).long(), weights=valid.float(), minlength=k)[:k].long()
    n_neg = (a.numel() - valid.sum()).reshape(1)
    return order, torch.cat([n_neg, n_neg + torch.cumsum(lens, 0)]).to(torch.int32)


def kmeans_update(bank, assign, centroids, k, counts=None, meta=None, update_means=True):
    n = assign.numel()
    for c in range(k):
        m = assign == c
        if update_means and m.any():
            centroids[c] = bank[:n][m].mean(dim=0)
        if counts is not None:
            counts[c] = m.sum()
    if meta is not None:
        meta[:n, 2] = assign.float()
    return group_by_cluster(assign, k)


def kmeans_segment_means(bank, order, seg_off, centroids, k, sums_only=False):
    for c in range(k):
        rows = order[int(seg_off[c]):int(seg_off[c + 1])].long()
        if rows.numel():
            centroids[c] = bank[rows].sum(dim=0) if sums_only else bank[rows].mean(dim=0)
        elif sums_only:
            centroids[c] = 0


def ivf2_slack(update_interval):
    return (min(max(int(update_interval), 64), 512) + 15) // 16 * 16


def ivf_capacity(longest_lists_total, k):
    cap = max(2048, (int(longest_lists_total) + 2047) // 2048 * 2048)
    return cap if (cap // 2048) * k <= 16384 else None


def knn_search_ivf(bank, inv_norm, meta, queries, k, now, count, centroids, nprobe, list_rows, list_off,
                   list_len, cap, idx_base=0):
    # host-logic stand-in: the lists must describe the same candidate sets as the mask
    n = count
    assert int(list_off[256]) == n and int(list_len.sum()) == int((meta[:n, 2] >= 0).sum())
    for c in range(256):
        rows = list_rows[int(list_off[c]):int(list_off[c]) + int(list_len[c])].long()
        assert bool((meta[rows, 2] == c).all())
    s, i = knn_search(bank, inv_norm, meta, queries, k, now, count=count, idx_base=idx_base,
                      centroids=centroids, nprobe=nprobe)
    return s, i, torch.zeros(1, dtype=torch.int32)
