"""ctypes access to the C oracle (``oracle/aura_oracle.c``).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_float, c_int, c_int64, c_void_p

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libaura_oracle.so")
_lib = None


def build() -> str:
    subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
    return SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        _lib = ctypes.CDLL(SO)
    return _lib


def _p(t: torch.Tensor):
    assert t.device.type == "cpu" and t.is_contiguous()
    return c_void_p(t.data_ptr())


def izh_run_nt(I, v, u, a, b, c, d, dt):
    I = I.contiguous(); S = torch.empty_like(I); v = v.clone(); u = u.clone()
    lib().oracle_izh_run_nt(_p(I), _p(S), _p(v), _p(u), c_float(a), c_float(b), c_float(c), c_float(d),
                            c_float(dt), c_int64(I.shape[0]), c_int64(I.shape[1]))
    return S, v, u


def adex_run_nt(I, V, w, params):
    I = I.contiguous(); S = torch.empty_like(I); V = V.clone(); w = w.clone()
    p = params.float().contiguous()
    lib().oracle_adex_run_nt(_p(I), _p(S), _p(V), _p(w), _p(p), c_int64(I.shape[0]), c_int64(I.shape[1]))
    return S, V, w


def lif_run(x, mem, beta, thr):
    x = x.contiguous(); S = torch.empty_like(x); mem = mem.clone()
    B, T, size = x.shape
    lib().oracle_lif_run(_p(x), _p(S), _p(mem), _p(beta.contiguous()), _p(thr.contiguous()),
                         c_int64(B), c_int64(T), c_int64(size))
    return S, mem


def gif_run(h, v, theta, decay, L, alpha, thr0):
    h = h.contiguous(); S = torch.empty_like(h); v = v.clone(); theta = theta.clone()
    rows, T, H = h.shape
    fn = lib().oracle_gif_run_f32 if h.dtype == torch.float32 else lib().oracle_gif_run_bf16
    fn(_p(h), _p(S), _p(v), _p(theta), c_float(decay), c_int(L), c_float(alpha), c_float(thr0),
       c_int64(rows), c_int64(T), c_int64(H))
    return S, v, theta


def knn_query(bank, meta, q, now, k):
    """Reference-cost single query (bank re-normalised per query) -> (scores[k], idx[k])."""
    N, D = bank.shape
    scores = torch.empty(N)
    lib().oracle_knn_scores(_p(bank), _p(meta), _p(q.contiguous()), c_float(now), c_int64(N), c_int64(D),
                            _p(scores))
    out_s = torch.empty(k); out_i = torch.empty(k, dtype=torch.int32)
    lib().oracle_topk(_p(scores), c_int64(N), c_int(k), _p(out_s), _p(out_i))
    return out_s, out_i


def sleef_expf_u10(x: torch.Tensor) -> torch.Tensor:
    """SLEEF's vector expf_u10 restated (see aura_oracle.c)."""
    x = x.float().contiguous(); out = torch.empty_like(x)
    lib().oracle_sleef_expf_u10(_p(x), _p(out), c_int64(x.numel()))
    return out
