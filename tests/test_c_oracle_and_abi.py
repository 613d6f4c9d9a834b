"""(1) The C oracle (oracle/aura_oracle.c: the scalar arithmetic the HIP kernels implement) agrees
bit for bit with the PyTorch-op oracle; (2) libaura_hip.so loads without a GPU and exports every
symbol that include/aura_hip.h declares; (3) the product refuses to run without HIP tensors."""
import os
import re

import pytest
import torch

from oracle import aura_oracle as O
from oracle import c_oracle as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_oracle_matches_torch_oracle():
    g = torch.Generator().manual_seed(0)
    I = 20 * torch.rand(300, 100, generator=g)
    v, u = O.izh_initial_state(300, 0.2)
    for a, b in zip(O.izh_run(I, v, u, 0.02, 0.2, -65, 8, 0.2), C.izh_run_nt(I, v, u, 0.02, 0.2, -65, 8, 0.2)):
        assert torch.equal(a, b)
    x = torch.randn(4, 9, 32, generator=g)
    beta, thr = torch.full((32,), 0.95), torch.full((32,), 0.5)
    for a, b in zip(O.lif_run(x, torch.zeros(4, 32), beta, thr), C.lif_run(x, torch.zeros(4, 32), beta, thr)):
        assert torch.equal(a, b)
    for dt in (torch.float32, torch.bfloat16):
        h = (torch.randn(50, 16, 96, generator=g) * 3).to(dt)
        v0, t0 = O.gif_initial_state(50, 96, 1.0, dt)
        for a, b in zip(O.gif_run(h, v0, t0, O.gif_decay(), 8, 0.01, 1.0), C.gif_run(h, v0, t0, O.gif_decay(), 8, 0.01, 1.0)):
            assert torch.equal(a, b)
    p = O.adex_params(a=2.0, b=60.0)
    Ia = 600 * torch.rand(64, 100, generator=g)
    sa, _, _ = O.adex_run(Ia, torch.full((64,), -70.0), torch.zeros(64), p)
    sb, _, _ = C.adex_run_nt(Ia, torch.full((64,), -70.0), torch.zeros(64), p)
    assert (sa != sb).float().mean().item() <= 1e-4     # expf vs SLEEF exp: 1-ulp differences only


def test_c_oracle_knn_matches_torch_oracle():
    g = torch.Generator().manual_seed(1)
    bank = torch.randn(3000, 48, generator=g)
    meta = torch.zeros(3000, 4); meta[:, 0] = 0.5 + 0.5 * torch.rand(3000, generator=g); meta[:, 1] = 1.7e9
    q = bank[5] + 0.05 * torch.randn(48, generator=g)
    s, i = C.knn_query(bank, meta, q, 1.7e9, 8)
    ri, rs = O.knn_exact_batch(bank, meta[:, 0], meta[:, 1], q.unsqueeze(0), 8, 1.7e9)
    assert i.long().tolist() == ri[0].tolist() and torch.allclose(s, rs[0], atol=1e-5)


def test_library_exports_every_declared_symbol():
    from aura_snn_rag_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "aura_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(aura_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in aura_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes prototypes and header disagree"
    assert lib.aura_version().decode().endswith("gfx950")


def test_product_has_no_cpu_fallback():
    from aura_snn_rag_amd import ops
    from aura_snn_rag_amd.base.neuron import IzhikevichNeuron, VectorizedLIFNeuron
    from aura_snn_rag_amd.core.hippocampal import HippocampalFormation
    from aura_snn_rag_amd.core.language_zone.gif_neuron import GIFNeuron
    with pytest.raises(ops.AuraDeviceError):
        IzhikevichNeuron()(torch.zeros(4, 8))
    with pytest.raises(ops.AuraDeviceError):
        VectorizedLIFNeuron(8)(torch.zeros(2, 8))
    with pytest.raises(ops.AuraDeviceError):
        GIFNeuron(4, 4)(torch.zeros(1, 2, 4))
    hf = HippocampalFormation(feature_dim=8, max_memories=16, n_place_cells=4, n_time_cells=3, n_grid_cells=3, device="cpu")
    with pytest.raises(ops.AuraDeviceError):
        hf.create_episodic_memory("a", "a", torch.ones(8))
    assert hf.retrieve_similar_memories(torch.ones(8)) == []      # empty bank: [] as the reference


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "aura_snn_rag_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"
