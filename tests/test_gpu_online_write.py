"""the online nearest-centroid / running-mean update (hippocampal.py:218-230) with the
distances taken out of the serial chain (aura_bank_write_online) must equal the one-workgroup serial kernel
(aura_bank_write with centroids, the checker; itself pinned to the oracle in test_gpu_knn.py) BIT FOR BIT:
centroid ids, counts, centroid values, bank rows, norms."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _state(M, D, k_rows, seed, dev, clustered=False, counts_zero=False):
    g = torch.Generator().manual_seed(seed)
    cent = torch.zeros(256, D)
    if clustered:
        cent[:k_rows] = 4.0 * torch.randn(k_rows, D, generator=g)
    else:
        cent[:k_rows] = 0.3 * torch.randn(k_rows, D, generator=g)
    counts = torch.zeros(256)
    if not counts_zero:
        counts[:k_rows] = torch.randint(1, 5000, (k_rows,), generator=g).float()
    st = dict(bank=torch.zeros(M, D), loc=torch.zeros(M, 2), meta=torch.zeros(M, 4), inv=torch.zeros(M),
              cent=cent, counts=counts)
    return {k: v.to(dev) for k, v in st.items()}


def _write(ops, st, feats, slots, eff_k, serial):
    cur = torch.tensor([0.25, -1.0], device=feats.device)
    ops.bank_write(st["bank"], st["loc"], st["meta"], st["inv"], feats, slots, cur, 1234.0,
                   centroids=st["cent"], centroid_counts=st["counts"], eff_k=eff_k, distinct_slots=True,
                   serial=serial)


CASES = [
    # (rows, D, eff_k, clustered, counts_zero)
    (1, 64, 256, False, False),
    (7, 64, 256, False, False),
    (256, 768, 256, False, False),      # Gaussian rows: best and second best centroid are near-ties all the time
    (512, 768, 256, False, False),
    (300, 768, 256, True, False),       # clustered rows: the bound decides almost every row
    (513, 96, 256, False, True),        # counts start at 0: eta = 1, the centroid jumps onto the row
    (200, 50, 200, False, False),       # D % 4 != 0 (scalar loads), fewer than 256 centroids
    (5000, 128, 256, False, False),     # more than one phase A / phase B chunk
    (400, 1024, 17, True, False),
    (300, 960, 256, False, False),      # the widest row whose elements all live on waves 1-15 (wave 0 only decides)
    (120, 964, 64, False, False),       # one step wider: wave 0 owns elements again
]


@pytest.mark.parametrize("n,D,eff_k,clustered,counts_zero", CASES)
def test_online_write_equals_serial_kernel(n, D, eff_k, clustered, counts_zero):
    from aura_snn_rag_amd import ops
    dev = torch.device("cuda")
    M = n + 10
    a = _state(M, D, eff_k, 11, dev, clustered, counts_zero)
    b = {k: v.clone() for k, v in a.items()}
    g = torch.Generator().manual_seed(5)
    if clustered:
        pick = torch.randint(0, eff_k, (n,), generator=g)
        feats = a["cent"][:eff_k].cpu()[pick] + 0.05 * torch.randn(n, D, generator=g)
    else:
        feats = torch.randn(n, D, generator=g)
    feats = feats.to(dev).contiguous()
    slots = torch.randperm(M, generator=g)[:n].to(dev)
    # two batches in a row: the second starts from the table the first left
    h = n // 2 if n > 1 else n
    for lo, hi in ((0, h), (h, n)):
        if hi > lo:
            _write(ops, a, feats[lo:hi].contiguous(), slots[lo:hi].contiguous(), eff_k, serial=True)
            _write(ops, b, feats[lo:hi].contiguous(), slots[lo:hi].contiguous(), eff_k, serial=False)
    torch.cuda.synchronize()
    assert torch.equal(a["meta"], b["meta"]), "centroid ids differ"
    assert torch.equal(a["counts"], b["counts"])
    assert torch.equal(a["cent"], b["cent"]), "centroid values differ"
    assert torch.equal(a["bank"], b["bank"]) and torch.equal(a["inv"], b["inv"]) and torch.equal(a["loc"], b["loc"])


def test_online_write_exact_ties_go_to_the_lower_centroid():
    """Identical centroids (exact distance ties, moved and unmoved) and rows identical to earlier rows."""
    from aura_snn_rag_amd import ops
    dev = torch.device("cuda")
    D, n = 64, 300
    a = _state(n + 4, D, 256, 3, dev)
    a["cent"][10] = a["cent"][200]; a["cent"][11] = a["cent"][200]; a["cent"][250] = a["cent"][3]
    a["counts"][10] = a["counts"][200] = a["counts"][11] = 7.0
    b = {k: v.clone() for k, v in a.items()}
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(n, D, generator=g)
    feats[40:80] = a["cent"][200].cpu() + 0.01 * torch.randn(40, D, generator=g)     # fight over the tied trio
    feats[100] = feats[41]; feats[101] = feats[41]
    feats = feats.to(dev).contiguous()
    slots = torch.arange(n, device=dev)
    _write(ops, a, feats, slots, 256, serial=True)
    _write(ops, b, feats, slots, 256, serial=False)
    torch.cuda.synchronize()
    assert torch.equal(a["meta"], b["meta"]) and torch.equal(a["counts"], b["counts"]) and torch.equal(a["cent"], b["cent"])


def test_product_write_path_uses_the_online_kernel_and_matches_the_oracle_bank():
    """create_episodic_memories with the index on (distinct slots -> aura_bank_write_online) against a second
    bank forced through the serial kernel: same metadata, centroids, counts after interleaved rebuilds."""
    from aura_snn_rag_amd import ops
    from aura_snn_rag_amd.core.hippocampal import HippocampalFormation
    kw = dict(feature_dim=96, max_memories=4096, n_place_cells=8, n_time_cells=4, n_grid_cells=4, device="cuda",
              use_centroid_index=True)
    a, b = HippocampalFormation(**kw), HippocampalFormation(**kw)
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(1500, 96, generator=g)
    ids = [f"m{i}" for i in range(1500)]
    real = ops.bank_write

    def serial_only(*args, **kwargs):
        kwargs["serial"] = True
        return real(*args, **kwargs)

    torch.manual_seed(1)
    a.create_episodic_memories(ids, feats)
    ops.bank_write = serial_only
    try:
        torch.manual_seed(1)
        b.create_episodic_memories(ids, feats)
    finally:
        ops.bank_write = real
    assert a._index_ready and b._index_ready
    ma, mb = a.memory_metadata.clone(), b.memory_metadata.clone()
    ma[:, 1] = 0; mb[:, 1] = 0                                       # wall-clock timestamps differ
    assert torch.equal(ma, mb)
    assert torch.equal(a.centroids, b.centroids) and torch.equal(a.centroid_counts, b.centroid_counts)
