"""Host-side logic of the product on CPU (HIP ops replaced by the oracle stub, see
tests/cpu_stub_ops.py): write planning, batched == sequential semantics, rebuild triggers, id
maps, candidate fallback, ingest record handling (ports of the reference's
tests/test_ingestion_and_gating.py:30-79)."""
import json

import pytest
import torch

from oracle import aura_oracle as O
from tests import cpu_stub_ops as stub

NOW = 1.7e9 + 5.0


@pytest.fixture()
def hmod(monkeypatch):
    from aura_snn_rag_amd.core import hippocampal as H
    monkeypatch.setattr(H, "ops", stub)
    monkeypatch.setattr(H.time, "time", lambda: NOW)
    return H


def _hf(H, D=16, M=400, **kw):
    return H.HippocampalFormation(n_place_cells=4, n_time_cells=3, n_grid_cells=3, max_memories=M,
                                  feature_dim=D, device="cpu", **kw)


def test_state_dict_names_match_reference(hmod):
    names = set(_hf(hmod).state_dict().keys())
    assert names == {"place_centers", "place_radii", "grid_spacings", "grid_orientations", "grid_phases",
                     "time_intervals", "time_widths", "memory_features", "memory_locations", "memory_metadata",
                     "k_const", "centroids", "centroid_counts"}


def test_batched_write_equals_sequential_and_oracle(hmod):
    feats = torch.randn(300, 16) * (2 * torch.rand(300, 1))
    banks = []
    for mode in ("seq", "batch", "oracle"):
        if mode == "oracle":
            b = O.OracleBank(400, 16, centroids_k=8, centroids_update_interval=32)
            torch.manual_seed(3)
            for i in range(300):
                b.write(f"m{i}", feats[i], NOW)
            banks.append((b.features, b.metadata, b.centroids, b.centroid_counts, b.id_to_idx, b.count))
            continue
        hf = _hf(hmod)
        hf.centroids_k, hf.centroids_update_interval = 8, 32
        torch.manual_seed(3)      # after construction: the ctor draws place/grid cells from the RNG
        if mode == "seq":
            for i in range(300):
                hf.create_episodic_memory(f"m{i}", "e", feats[i])
        else:
            for i in range(0, 300, 75):
                hf.create_episodic_memories([f"m{j}" for j in range(i, i + 75)], feats[i:i + 75])
        banks.append((hf.memory_features, hf.memory_metadata, hf.centroids, hf.centroid_counts, hf.id_to_idx,
                      hf.memory_count))
    for other in banks[1:]:
        assert torch.equal(banks[0][0], other[0]) and torch.equal(banks[0][1], other[1])
        assert torch.allclose(banks[0][2], other[2], atol=1e-6) and torch.equal(banks[0][3], other[3])
        assert banks[0][4] == other[4] and banks[0][5] == other[5]


def test_recall_paths_and_fallback(hmod):
    hf = _hf(hmod)
    hf.centroids_k, hf.centroids_update_interval = 8, 32
    ob = O.OracleBank(400, 16, centroids_k=8, centroids_update_interval=32)
    feats = torch.randn(200, 16)
    torch.manual_seed(1)
    for i in range(200):
        ob.write(f"m{i}", feats[i], NOW)
    torch.manual_seed(1)
    hf.create_episodic_memories([f"m{i}" for i in range(200)], feats)
    assert hf._index_ready
    for j in (0, 50, 199):
        q = feats[j] + 0.05 * torch.randn(16)
        assert [r[0] for r in hf.retrieve_similar_memories(q, k=5)] == [r[0] for r in ob.recall_ids(q, 5, NOW)]
    hf.use_centroid_index = ob.use_centroid_index = False
    q = feats[3]
    got, ref = hf.retrieve_similar_memories(q, k=7), ob.recall_ids(q, 7, NOW)
    assert [g[0] for g in got] == [r[0] for r in ref]
    assert hf.retrieve_similar_memories(q, k=1000)[0][0] == "m3"     # k clamps to the count
    s, r = hf.recall_batch(feats[:4], k=3)
    assert r[:, 0].tolist() == [0, 1, 2, 3]


def test_overflow_policies_and_failed_write_is_transactional(hmod, monkeypatch):
    feats = torch.randn(7, 16)
    hf = _hf(hmod, M=4, use_centroid_index=False)
    for i in range(7):
        hf.create_episodic_memory(f"m{i}", "e", feats[i])
    assert hf.memory_count == 4 and hf.id_to_idx == {"m0": 0, "m1": 1, "m2": 2, "m3": 3, "m4": 0, "m5": 0, "m6": 0}
    assert hf._idx_to_id[0] == "m6"
    fifo = _hf(hmod, M=4, use_centroid_index=False, overflow="fifo")
    for i in range(7):
        fifo.create_episodic_memory(f"m{i}", "e", feats[i])
    assert [fifo.id_to_idx[f"m{i}"] for i in (4, 5, 6)] == [0, 1, 2]

    def boom(*a, **k):
        raise RuntimeError("launch failed")
    hf2 = _hf(hmod, M=8, use_centroid_index=False)
    hf2.create_episodic_memory("ok", "e", feats[0])
    monkeypatch.setattr(stub, "bank_write", boom)
    with pytest.raises(RuntimeError):
        hf2.create_episodic_memory("bad", "e", feats[1])
    assert hf2.memory_count == 1 and "bad" not in hf2.id_to_idx


class DummyTokenizer:
    def encode(self, text, **kwargs):
        return [1, 2, 3]


def _stub_one_shot(prefix):
    def fn(text, tokenizer, model, hippocampus, device, memory_id=None):
        mem_id = memory_id or f"{prefix}-{hippocampus.memory_count}"
        hippocampus.create_episodic_memory(mem_id, mem_id, torch.ones(hippocampus.memory_features.shape[1]))
        fn.texts.append(text)
        return mem_id
    fn.texts = []
    return fn


def test_ingest_helpers(hmod, monkeypatch, tmp_path):
    from aura_snn_rag_amd import ingest
    hf = _hf(hmod, M=50)
    one = _stub_one_shot("jsonl")
    monkeypatch.setattr(ingest, "one_shot_memorize_text", one)
    p = tmp_path / "a.jsonl"
    p.write_text("\n".join([json.dumps({"text": "hello world"}), "", "not json",
                            json.dumps({"instruction": "do X", "output": "done"}),
                            json.dumps({"prompt": "p", "completion": "c"}),
                            json.dumps({"input": "i", "output": "o"}), json.dumps({"other": 1}),
                            json.dumps("bare string")]) + "\n")
    assert ingest.ingest_jsonl_to_memory(str(p), DummyTokenizer(), object(), hf, device="cpu", max_items=10) == 5
    assert one.texts == ["hello world", "Instruction: do X\nResponse: done", "Prompt: p\nCompletion: c",
                         "Input: i\nOutput: o", "bare string"]
    assert hf.memory_count == 5 and "jsonl-4" in hf.id_to_idx
    assert ingest.ingest_jsonl_to_memory(str(p), DummyTokenizer(), object(), hf, device="cpu", max_items=2) == 2
    one2 = _stub_one_shot("csv")
    monkeypatch.setattr(ingest, "one_shot_memorize_text", one2)
    c = tmp_path / "a.csv"
    c.write_text("Q1,A1\nonly_one_column\n , \nQ2,A2,extra\n")
    assert ingest.ingest_csv_pairs_to_memory(str(c), DummyTokenizer(), object(), hf, device="cpu") == 2
    assert one2.texts == ["Question: Q1\nAnswer: A1", "Question: Q2\nAnswer: A2"]
    assert ingest.ingest_jsonl_to_memory(str(p), None, object(), hf, device="cpu") == 0
    # custom store / retrieve / bulk
    mid = ingest.store_custom_memory(hf, torch.randn(3, 16), memory_id="ext")
    assert mid == "ext" and "ext" in hf.id_to_idx
    assert ingest.retrieve_custom_memories(hf, hf.memory_features[hf.id_to_idx["ext"]].unsqueeze(0), k=1)[0][0] == "ext"
    n = ingest.ingest_feature_batches(hf, [([f"b{i}" for i in range(5)], torch.randn(5, 16).to(torch.bfloat16)),
                                           ([f"c{i}" for i in range(5)], torch.randn(5, 16))], max_items=7)
    assert n == 7 and "c1" in hf.id_to_idx and "c2" not in hf.id_to_idx


def test_processor_combines_zones_like_reference():
    from aura_snn_rag_amd.base.snn_processor import NeuromorphicProcessor, _softmax64
    import numpy as np

    class Z(torch.nn.Module):
        def __init__(self, s):
            super().__init__(); self.s = s
        def forward(self, x, context=None):
            return x * self.s, {"avg_firing_rate": self.s}
    p = NeuromorphicProcessor(d_model=4)
    p.add_zone("a", Z(1.0)); p.add_zone("b", Z(3.0)); p.add_zone("c", Z(100.0))
    x = torch.randn(2, 4)
    y = p.process(x, zone_weights={"a": 0.7, "b": 0.3, "c": 0.001})
    w = _softmax64(np.array([0.7, 0.3]))
    assert torch.allclose(y, float(w[0]) * x + float(w[1]) * 3 * x, atol=1e-6)
    assert set(p._current_zone_activities) == {"a", "b"}


def test_bank_state_roundtrip(hmod):
    """memory_count / id maps survive a save-load cycle (the reference drops them)."""
    hf = _hf(hmod, M=6, use_centroid_index=False)
    feats = torch.randn(8, 16)
    for i in range(8):                       # overflows: slot 0 is overwritten twice
        hf.create_episodic_memory(f"m{i}", "e", feats[i])
    sd, bs = hf.state_dict(), hf.bank_state()
    hf2 = _hf(hmod, M=6, use_centroid_index=False)
    hf2.load_state_dict(sd)
    assert hf2.memory_count == 0             # what the reference gives you
    hf2.load_bank_state(bs)
    assert hf2.memory_count == 6 and hf2.id_to_idx == hf.id_to_idx
    assert hf2.retrieve_similar_memories(feats[3], k=1)[0][0] == "m3"
    assert hf2.retrieve_similar_memories(feats[7], k=1)[0][0] == "m7"
    import pickle
    assert pickle.loads(pickle.dumps(bs)) == bs


def test_bulk_write_implicit_ids(hmod):
    hf = _hf(hmod, M=100)
    hf.centroids_k = 4
    feats = torch.randn(60, 16)
    hf.create_episodic_memory("first", "e", feats[0])
    assert hf.bulk_write(feats[1:41], id_prefix="doc-", first_index=1000) == 40
    assert hf.memory_count == 41 and hf._index_ready          # one rebuild at the end
    assert hf.id_of_row(0) == "first" and hf.id_of_row(1) == "doc-1000" and hf.id_of_row(40) == "doc-1039"
    hf.use_centroid_index = False
    assert hf.retrieve_similar_memories(feats[17], k=1)[0][0] == "doc-1016"
    assert hf.bulk_write(torch.randn(100, 16)) == 59           # clipped at max_memories
    assert len(hf.id_to_idx) == 1                              # no per-row Python objects
