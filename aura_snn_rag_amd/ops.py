"""Tensor-level wrappers over the C ABI (``include/aura_hip.h``).

Every function takes HIP-resident, contiguous torch tensors, checks shapes/dtypes on the host
(a bad shape reaching a hand-written kernel can fault the GPU) and launches on torch's current
stream.  There is no fallback: CPU tensors raise ``AuraDeviceError``.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import check, KNN_FLAG_LISTS_STALE, KNN_FLAG_NO_CANDIDATES


class AuraDeviceError(RuntimeError):
    """Raised when a hot-path op is asked to run on something that is not a HIP device."""


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need(t: torch.Tensor, name: str, dtype=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise AuraDeviceError(
            f"{name} is on {t.device}: aura_snn_rag_amd runs the hot path only as HIP kernels on "
            f"an AMD GPU (no CPU/PyTorch fallback).")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def lib():
    return _lib.load()


# ---------------------------------------------------------------------------------------
# neurons
# ---------------------------------------------------------------------------------------

def izh_run_nt(I, spikes, v, u, a, b, c, d, dt) -> None:
    """I, spikes: [N, T] fp32; v, u: [N] fp32 (updated in place)."""
    _need(I, "I", torch.float32); _need(spikes, "spikes", torch.float32)
    _need(v, "v", torch.float32); _need(u, "u", torch.float32)
    N, T = I.shape
    if spikes.shape != I.shape or v.numel() != N or u.numel() != N:
        raise ValueError("izh_run_nt: shape mismatch")
    check(lib().aura_izh_run_nt(_p(I), _p(spikes), _p(v), _p(u), a, b, c, d, dt, N, T, _stream()),
          "aura_izh_run_nt")


def izh_run_btd(I, spikes, v, u, a, b, c, d, dt) -> None:
    """I, spikes: [B, T, D] fp32; v, u: [B*D] fp32 indexed b*D+d."""
    _need(I, "I", torch.float32); _need(spikes, "spikes", torch.float32)
    _need(v, "v", torch.float32); _need(u, "u", torch.float32)
    B, T, D = I.shape
    if spikes.shape != I.shape or v.numel() != B * D or u.numel() != B * D:
        raise ValueError("izh_run_btd: shape mismatch")
    check(lib().aura_izh_run_btd(_p(I), _p(spikes), _p(v), _p(u), a, b, c, d, dt, B, T, D,
                                 _stream()), "aura_izh_run_btd")


def _adex_params(params: Sequence[float]):
    if len(params) != 11:
        raise ValueError("AdEx needs 11 parameters")
    return (ctypes.c_float * 11)(*[float(x) for x in params])


def adex_run_nt(I, spikes, V, w, params: Sequence[float]) -> None:
    _need(I, "I", torch.float32); _need(spikes, "spikes", torch.float32)
    _need(V, "V", torch.float32); _need(w, "w", torch.float32)
    N, T = I.shape
    if spikes.shape != I.shape or V.numel() != N or w.numel() != N:
        raise ValueError("adex_run_nt: shape mismatch")
    arr = _adex_params(params)
    check(lib().aura_adex_run_nt(_p(I), _p(spikes), _p(V), _p(w), ctypes.addressof(arr), N, T,
                                 _stream()), "aura_adex_run_nt")


def adex_run_btd(I, spikes, V, w, params: Sequence[float]) -> None:
    _need(I, "I", torch.float32); _need(spikes, "spikes", torch.float32)
    _need(V, "V", torch.float32); _need(w, "w", torch.float32)
    B, T, D = I.shape
    if spikes.shape != I.shape or V.numel() != B * D or w.numel() != B * D:
        raise ValueError("adex_run_btd: shape mismatch")
    arr = _adex_params(params)
    check(lib().aura_adex_run_btd(_p(I), _p(spikes), _p(V), _p(w), ctypes.addressof(arr), B, T, D,
                                  _stream()), "aura_adex_run_btd")


def lif_run(x, spikes, mem, beta, threshold) -> None:
    """x, spikes: [B, T, size]; mem: [B, size] in/out; beta, threshold: [size].  fp32, or all five bf16 (a
    module moved to bf16: every op rounds to bf16)."""
    dt = x.dtype
    if dt not in (torch.float32, torch.bfloat16):
        raise TypeError(f"lif_run supports fp32 and bf16, got {dt}")
    for t, n in ((x, "x"), (spikes, "spikes"), (mem, "mem"), (beta, "beta"), (threshold, "threshold")):
        _need(t, n, dt)
    B, T, size = x.shape
    if spikes.shape != x.shape or mem.shape != (B, size) or beta.numel() != size or \
            threshold.numel() != size:
        raise ValueError("lif_run: shape mismatch")
    fn = lib().aura_lif_run if dt == torch.float32 else lib().aura_lif_run_bf16
    check(fn(_p(x), _p(spikes), _p(mem), _p(beta), _p(threshold), B, T, size, _stream()), "aura_lif_run")


def gif_run(h, out, v, theta, decay: float, L: int, alpha: float, threshold: float, T: int,
            time_invariant: bool = False, mean_out: bool = False) -> None:
    """h: [rows, T, H] (or [rows, H] if time_invariant); out: [rows, T, H] (or [rows, H] if
    mean_out); v, theta: [rows, H] in/out.  fp32 or bf16 (all four the same dtype)."""
    dt = h.dtype
    if dt not in (torch.float32, torch.bfloat16):
        raise TypeError(f"gif_run supports fp32 and bf16, got {dt}")
    for t, n in ((h, "h"), (out, "out"), (v, "v"), (theta, "theta")):
        _need(t, n, dt)
    rows, H = v.shape
    if theta.shape != v.shape:
        raise ValueError("gif_run: v/theta shape mismatch")
    if tuple(h.shape) != ((rows, H) if time_invariant else (rows, T, H)):
        raise ValueError(f"gif_run: h has shape {tuple(h.shape)}")
    if tuple(out.shape) != ((rows, H) if mean_out else (rows, T, H)):
        raise ValueError(f"gif_run: out has shape {tuple(out.shape)}")
    flags = (_lib.GIF_TIME_INVARIANT if time_invariant else 0) | (_lib.GIF_MEAN_OUT if mean_out else 0)
    check(lib().aura_gif_run(_p(h), _p(out), _p(v), _p(theta), decay, int(L), alpha, threshold,
                             rows, T, H, _lib.DTYPE_F32 if dt == torch.float32 else _lib.DTYPE_BF16,
                             flags, _stream()), "aura_gif_run")


# ---------------------------------------------------------------------------------------
# surrogate-gradient training path + prosody-modulated GIF (fp32 and bf16)
# ---------------------------------------------------------------------------------------

def _same_f32(shape, *named) -> None:
    for t, n in named:
        _need(t, n, torch.float32)
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{n}: expected shape {tuple(shape)}, got {tuple(t.shape)}")


def _same_dtype(dtype, shape, *named) -> None:
    for t, n in named:
        _need(t, n, dtype)
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{n}: expected shape {tuple(shape)}, got {tuple(t.shape)}")


def gif_train_forward(h, spikes, v, theta, save_a, save_theta, decay: float, L: int, alpha: float,
                      threshold: float) -> None:
    """h [rows, T, H] -> spikes, save_a, save_theta [rows, T, H]; v, theta [rows, H] in/out (fp32 or bf16)."""
    rows, T, H = h.shape
    if h.dtype == torch.bfloat16:
        _same_dtype(torch.bfloat16, (rows, T, H), (h, "h"), (spikes, "spikes"), (save_a, "save_a"), (save_theta, "save_theta"))
        _same_dtype(torch.bfloat16, (rows, H), (v, "v"), (theta, "theta"))
        check(lib().aura_gif_train_forward_bf16(_p(h), _p(spikes), _p(v), _p(theta), _p(save_a), _p(save_theta),
                                                decay, int(L), alpha, threshold, rows, T, H, _stream()),
              "aura_gif_train_forward_bf16")
        return
    _same_f32((rows, T, H), (h, "h"), (spikes, "spikes"), (save_a, "save_a"), (save_theta, "save_theta"))
    _same_f32((rows, H), (v, "v"), (theta, "theta"))
    check(lib().aura_gif_train_forward(_p(h), _p(spikes), _p(v), _p(theta), _p(save_a), _p(save_theta),
                                       decay, int(L), alpha, threshold, rows, T, H, _stream()),
          "aura_gif_train_forward")


def gif_backward(save_a, save_theta, g_spikes, g_h, g_v, g_theta, decay: float, L: int, alpha: float,
                 threshold: float) -> None:
    """g_v, g_theta [rows, H]: gradients of the final state in, of the initial state out (fp32 or bf16)."""
    rows, T, H = save_a.shape
    if save_a.dtype == torch.bfloat16:
        _same_dtype(torch.bfloat16, (rows, T, H), (save_a, "save_a"), (save_theta, "save_theta"),
                    (g_spikes, "g_spikes"), (g_h, "g_h"))
        _same_dtype(torch.bfloat16, (rows, H), (g_v, "g_v"), (g_theta, "g_theta"))
        check(lib().aura_gif_backward_bf16(_p(save_a), _p(save_theta), _p(g_spikes), _p(g_h), _p(g_v), _p(g_theta),
                                           decay, int(L), alpha, threshold, rows, T, H, _stream()),
              "aura_gif_backward_bf16")
        return
    _same_f32((rows, T, H), (save_a, "save_a"), (save_theta, "save_theta"), (g_spikes, "g_spikes"),
              (g_h, "g_h"))
    _same_f32((rows, H), (g_v, "g_v"), (g_theta, "g_theta"))
    check(lib().aura_gif_backward(_p(save_a), _p(save_theta), _p(g_spikes), _p(g_h), _p(g_v), _p(g_theta),
                                  decay, int(L), alpha, threshold, rows, T, H, _stream()),
          "aura_gif_backward")


def lif_train_forward(x, mem_in, beta, threshold, spikes, mem_out, pre) -> None:
    """One recording LIF step, [B, size]; fp32 or all bf16."""
    B, size = x.shape
    dt = x.dtype
    if dt not in (torch.float32, torch.bfloat16):
        raise TypeError(f"lif_train_forward supports fp32 and bf16, got {dt}")
    _same_dtype(dt, (B, size), (x, "x"), (mem_in, "mem_in"), (spikes, "spikes"), (mem_out, "mem_out"), (pre, "pre"))
    _same_dtype(dt, (size,), (beta, "beta"), (threshold, "threshold"))
    fn = lib().aura_lif_train_forward if dt == torch.float32 else lib().aura_lif_train_forward_bf16
    check(fn(_p(x), _p(mem_in), _p(beta), _p(threshold), _p(spikes), _p(mem_out), _p(pre), B, size, _stream()),
          "aura_lif_train_forward")


def lif_backward(pre, g_spikes, g_mem, beta, threshold, slope, g_x, g_mem_prev, raw_slope) -> None:
    """raw_slope [B, size] is fp32 in both forms (sum it over the batch for d/d slope)."""
    B, size = pre.shape
    dt = pre.dtype
    if dt not in (torch.float32, torch.bfloat16):
        raise TypeError(f"lif_backward supports fp32 and bf16, got {dt}")
    _same_dtype(dt, (B, size), (pre, "pre"), (g_spikes, "g_spikes"), (g_mem, "g_mem"), (g_x, "g_x"),
                (g_mem_prev, "g_mem_prev"))
    _same_f32((B, size), (raw_slope, "raw_slope"))
    _same_dtype(dt, (size,), (beta, "beta"), (threshold, "threshold"), (slope, "slope"))
    fn = lib().aura_lif_backward if dt == torch.float32 else lib().aura_lif_backward_bf16
    check(fn(_p(pre), _p(g_spikes), _p(g_mem), _p(beta), _p(threshold), _p(slope), _p(g_x), _p(g_mem_prev),
             _p(raw_slope), B, size, _stream()), "aura_lif_backward")


def _prosody_dtype(h, who: str):
    if h.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"{who} supports fp32 and bf16, got {h.dtype}")
    return h.dtype


def gif_prosody_run(h, gains, spikes, v, theta, decay: float, L: int, alpha: float, threshold: float,
                    strength: float) -> None:
    """h, spikes [rows, T, H]; gains [rows, T] or None; v, theta [rows, H] in/out.  fp32, or everything
    (gains included) bf16."""
    rows, T, H = h.shape
    dt = _prosody_dtype(h, "gif_prosody_run")
    _same_dtype(dt, (rows, T, H), (h, "h"), (spikes, "spikes"))
    _same_dtype(dt, (rows, H), (v, "v"), (theta, "theta"))
    if gains is not None:
        _same_dtype(dt, (rows, T), (gains, "gains"))
    fn = lib().aura_gif_prosody_run if dt == torch.float32 else lib().aura_gif_prosody_run_bf16
    check(fn(_p(h), _p(gains), _p(spikes), _p(v), _p(theta), decay, int(L), alpha, threshold, strength, rows, T, H,
             _stream()), "aura_gif_prosody_run")


def gif_prosody_train_forward(h, gains, spikes, v, theta, save_a, save_theta, decay: float, L: int, alpha: float,
                              threshold: float, strength: float) -> None:
    rows, T, H = h.shape
    dt = _prosody_dtype(h, "gif_prosody_train_forward")
    _same_dtype(dt, (rows, T, H), (h, "h"), (spikes, "spikes"), (save_a, "save_a"), (save_theta, "save_theta"))
    _same_dtype(dt, (rows, H), (v, "v"), (theta, "theta"))
    if gains is not None:
        _same_dtype(dt, (rows, T), (gains, "gains"))
    fn = lib().aura_gif_prosody_train_forward if dt == torch.float32 else lib().aura_gif_prosody_train_forward_bf16
    check(fn(_p(h), _p(gains), _p(spikes), _p(v), _p(theta), _p(save_a), _p(save_theta), decay, int(L), alpha,
             threshold, strength, rows, T, H, _stream()), "aura_gif_prosody_train_forward")


def gif_prosody_backward(save_a, save_theta, h, gains, g_spikes, g_h, g_gains, g_v, g_theta, decay: float, L: int,
                         alpha: float, threshold: float, strength: float) -> None:
    """g_gains [rows, T], fp32 in both forms, must be zero on entry (channels are accumulated into it)."""
    rows, T, H = save_a.shape
    dt = _prosody_dtype(save_a, "gif_prosody_backward")
    _same_dtype(dt, (rows, T, H), (save_a, "save_a"), (save_theta, "save_theta"), (h, "h"), (g_spikes, "g_spikes"),
                (g_h, "g_h"))
    _same_dtype(dt, (rows, H), (g_v, "g_v"), (g_theta, "g_theta"))
    if gains is not None:
        _same_dtype(dt, (rows, T), (gains, "gains"))
        _same_f32((rows, T), (g_gains, "g_gains"))
    fn = lib().aura_gif_prosody_backward if dt == torch.float32 else lib().aura_gif_prosody_backward_bf16
    check(fn(_p(save_a), _p(save_theta), _p(h), _p(gains), _p(g_spikes), _p(g_h), _p(g_gains), _p(g_v), _p(g_theta),
             decay, int(L), alpha, threshold, strength, rows, T, H, _stream()), "aura_gif_prosody_backward")


# ---------------------------------------------------------------------------------------
# episodic bank
# ---------------------------------------------------------------------------------------

def bank_row_norms(bank, inv_norm, row0: int, n: int) -> None:
    _need(bank, "bank", torch.float32); _need(inv_norm, "inv_norm", torch.float32)
    M, D = bank.shape
    if row0 < 0 or n < 0 or row0 + n > M or inv_norm.numel() != M:
        raise ValueError("bank_row_norms: range out of bounds")
    check(lib().aura_bank_row_norms(_p(bank), _p(inv_norm), row0, n, D, _stream()),
          "aura_bank_row_norms")


def bank_shadow_update(bank, inv_norm, shadow, rho, row0: int = 0, n: Optional[int] = None, slots=None) -> None:
    """shadow[r] = bf16(bank[r] * inv_norm[r]) (the normalised row) and rho[r] = its L2 rounding
    residual, for the rows ``slots`` (int64 [n], device) or [row0, row0 + n)."""
    _need(bank, "bank", torch.float32); _need(shadow, "shadow", torch.bfloat16)
    _need(inv_norm, "inv_norm", torch.float32); _need(rho, "rho", torch.float32)
    M, D = bank.shape
    if shadow.shape != bank.shape or D % 8 or inv_norm.numel() != M or rho.numel() != M:
        raise ValueError("bank_shadow_update: shadow must match the bank, D % 8 == 0, inv_norm / rho [rows]")
    if slots is not None:
        _need(slots, "slots", torch.int64)
        n = slots.numel()
    elif n is None:
        n = M - row0
    if row0 < 0 or n < 0 or (slots is None and row0 + n > M):
        raise ValueError("bank_shadow_update: range out of bounds")
    check(lib().aura_bank_shadow_update(_p(bank), _p(inv_norm), _p(shadow), _p(rho), _p(slots), row0, n, D,
                                        _stream()), "aura_bank_shadow_update")


def make_shadow(bank, inv_norm, count: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(shadow bf16 [M, D], rho fp32 [M]) of rows [0, count) of a bank whose 1/||row|| are current."""
    M, D = bank.shape
    shadow = torch.empty(M, D, dtype=torch.bfloat16, device=bank.device)
    rho = torch.zeros(M, dtype=torch.float32, device=bank.device)
    bank_shadow_update(bank, inv_norm, shadow, rho, 0, M if count is None else count)
    return shadow, rho


def bank_write(bank, loc, meta, inv_norm, feats, slots, cur_loc, now: float,
               centroids=None, centroid_counts=None, eff_k: int = 0, distinct_slots: bool = False,
               serial: bool = False) -> None:
    """Write feats[n, D] into rows ``slots`` (int64 [n], device).  With ``centroids`` the reference's online
    nearest-centroid / running-mean update runs in row order (``hippocampal.py:218-230``): through
    ``aura_bank_write_online`` (distances out of the serial chain, same bits) when the caller vouches that the
    slots are distinct, else -- or with ``serial`` (tests: the checker) -- through the one-workgroup kernel."""
    for t, n_ in ((bank, "bank"), (loc, "loc"), (meta, "meta"), (inv_norm, "inv_norm"),
                  (feats, "feats"), (cur_loc, "cur_loc")):
        _need(t, n_, torch.float32)
    _need(slots, "slots", torch.int64)
    M, D = bank.shape
    n = feats.shape[0]
    sd = loc.shape[1]
    if feats.shape != (n, D) or slots.numel() != n or meta.shape != (M, 4) or \
            loc.shape[0] != M or cur_loc.numel() != sd or inv_norm.numel() != M:
        raise ValueError("bank_write: shape mismatch")
    # slots are planned on the host by HippocampalFormation._plan_slots (always within [0, M))
    if centroids is not None:
        _need(centroids, "centroids", torch.float32)
        _need(centroid_counts, "centroid_counts", torch.float32)
        if centroids.shape[1] != D or not (0 < eff_k <= centroids.shape[0] <= 256) or \
                centroid_counts.numel() < eff_k:
            raise ValueError("bank_write: centroid shape mismatch")
    if centroids is not None and distinct_slots and not serial and n > 0:
        L = lib()
        nbytes = L.aura_bank_write_online_workspace_bytes(n)
        ws = _workspace(bank.device, nbytes)
        base = (ws.data_ptr() + 255) // 256 * 256
        check(L.aura_bank_write_online(_p(bank), _p(loc), _p(meta), _p(inv_norm), _p(centroids),
                                       _p(centroid_counts), eff_k, _p(feats), _p(slots), _p(cur_loc), sd,
                                       now, n, D, base, nbytes, _stream()), "aura_bank_write_online")
        return
    check(lib().aura_bank_write(_p(bank), _p(loc), _p(meta), _p(inv_norm), _p(centroids),
                                _p(centroid_counts), eff_k, _p(feats), _p(slots), _p(cur_loc), sd,
                                now, n, D, _stream()), "aura_bank_write")


def bank_decay(meta, rate: float, count: int) -> None:
    _need(meta, "meta", torch.float32)
    if count < 0 or count > meta.shape[0] or meta.shape[1] != 4:
        raise ValueError("bank_decay: bad count")
    check(lib().aura_bank_decay(_p(meta), rate, count, _stream()), "aura_bank_decay")


# Scratch memory and the overflow flag are per (device, stream): two recalls in flight on different
# HIP streams never share candidate lists, thresholds or flags.  A buffer that has to grow is replaced
# (the old one is returned to torch's caching allocator, which keeps it alive for work already queued
# on its stream).
_workspaces = {}
_ovf_flags = {}


def _wkey(device):
    return (device, torch.cuda.current_stream(device).cuda_stream)


def _workspace(device, nbytes: int) -> torch.Tensor:
    key = _wkey(device)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _overflow_flag(device) -> torch.Tensor:
    """int32 [1], reset by the library's prep kernel on every call."""
    key = _wkey(device)
    f = _ovf_flags.get(key)
    if f is None:
        f = _ovf_flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return f


def knn_search(bank, inv_norm, meta, queries, k: int, now: float, count: Optional[int] = None,
               loc=None, q_loc=None, idx_base: int = 0, force_dense: bool = False,
               centroids=None, nprobe: int = 0, check_overflow: bool = True,
               fp32_scan: bool = False, shadow=None, rho=None, return_flag: bool = False):
    """Exact batched recall over rows [0, count) -> (scores [nq, k] fp32, idx [nq, k] int32).

    ``centroids`` (256 x D) + ``nprobe`` switches on the reference's centroid-candidate
    selection.  ``check_overflow`` reads one int back (a host sync) and transparently re-runs the
    dense path if a candidate list overflowed; pass False inside latency-critical loops whose
    data is known to be well behaved.  ``return_flag`` additionally returns the overflow flag
    (int32 [1] on the device) so that a caller can fold the check into a read of its own.
    """
    _need(bank, "bank", torch.float32); _need(inv_norm, "inv_norm", torch.float32)
    _need(meta, "meta", torch.float32); _need(queries, "queries", torch.float32)
    M, D = bank.shape
    N = M if count is None else int(count)
    nq = queries.shape[0]
    if queries.dim() != 2 or queries.shape[1] != D:
        raise ValueError(f"knn_search: queries must be [nq, {D}]")
    if not (0 < N <= M) or meta.shape != (M, 4) or inv_norm.numel() != M:
        raise ValueError("knn_search: bad bank/count")
    if not (0 < k <= min(N, 1024)):
        raise ValueError(f"knn_search: k={k} must be in [1, min(N, 1024)]")
    sd = 0
    if q_loc is not None:
        _need(loc, "loc", torch.float32); _need(q_loc, "q_loc", torch.float32)
        sd = loc.shape[1]
        if q_loc.shape != (nq, sd) or loc.shape[0] != M or sd > 4:
            raise ValueError("knn_search: location shape mismatch")
    if centroids is not None:
        _need(centroids, "centroids", torch.float32)
        if centroids.shape != (256, D) or not (0 < nprobe <= 256):
            raise ValueError("knn_search: centroids must be [256, D]")
    dev = bank.device
    out_s = torch.empty(nq, k, dtype=torch.float32, device=dev)
    out_i = torch.empty(nq, k, dtype=torch.int32, device=dev)
    if nq == 0:
        return (out_s, out_i, torch.zeros(1, dtype=torch.int32, device=dev)) if return_flag else (out_s, out_i)
    L = lib()
    nbytes = L.aura_knn_workspace_bytes(N, nq, k)
    if nbytes < 0:
        raise _lib.AuraHipError("aura_knn_workspace_bytes failed")
    ws = _workspace(dev, nbytes)
    base = (ws.data_ptr() + 255) // 256 * 256
    ovf = _overflow_flag(dev)

    use_shadow = shadow is not None and q_loc is None
    if use_shadow:
        _need(shadow, "shadow", torch.bfloat16); _need(rho, "rho", torch.float32)
        if shadow.shape != bank.shape or rho.numel() != M:
            raise ValueError("knn_search: shadow must have the bank's shape, rho one entry per row")

    def run(flags):
        if use_shadow:
            check(L.aura_knn_search_shadow(_p(bank), _p(shadow), _p(rho), _p(inv_norm), _p(meta), _p(queries), now,
                                           N, D, nq, k, idx_base, _p(out_s), _p(out_i), base, nbytes,
                                           flags, _p(ovf), _p(centroids), nprobe, _stream()),
                  "aura_knn_search_shadow")
            return
        check(L.aura_knn_search_ex(_p(bank), _p(inv_norm), _p(meta), _p(loc), sd, _p(queries),
                                   _p(q_loc), now, N, D, nq, k, idx_base, _p(out_s), _p(out_i),
                                   base, nbytes, flags, _p(ovf), _p(centroids), nprobe, _stream()),
              "aura_knn_search_ex")

    run(_lib.KNN_FORCE_DENSE if force_dense else (_lib.KNN_FP32_SCAN if fp32_scan else 0))
    if check_overflow and not force_dense and (int(ovf.item()) & ~_lib.KNN_FLAG_NO_CANDIDATES) != 0:
        run(_lib.KNN_FORCE_DENSE)
    return (out_s, out_i, ovf) if return_flag else (out_s, out_i)


def ivf_capacity(longest_lists_total: int, k: int) -> Optional[int]:
    """Candidate slots per query for the IVF recall: the sum of the nprobe longest lists rounded
    up to 2048; None if the two-level select cannot hold it ((cap/2048)*k must be <= 16384)."""
    cap = max(2048, (int(longest_lists_total) + 2047) // 2048 * 2048)
    return cap if (cap // 2048) * k <= 16384 else None


def knn_search_ivf(bank, inv_norm, meta, queries, k: int, now: float, count: int, centroids,
                   nprobe: int, list_rows, list_off, list_len, cap: int, idx_base: int = 0
                   ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Inverted-list recall over rows [0, count): (scores [nq, k], idx [nq, k], overflow flag [1]).
    The flag is a device tensor (non-zero: some query's probed lists exceed the slot capacity and
    the caller must use the masked full scan); reading it is the caller's decision."""
    _need(bank, "bank", torch.float32); _need(inv_norm, "inv_norm", torch.float32)
    _need(meta, "meta", torch.float32); _need(queries, "queries", torch.float32)
    _need(centroids, "centroids", torch.float32)
    for t, n in ((list_rows, "list_rows"), (list_off, "list_off"), (list_len, "list_len")):
        _need(t, n, torch.int32)
    M, D = bank.shape
    nq = queries.shape[0]
    if queries.dim() != 2 or queries.shape[1] != D or not (0 < count <= M) or meta.shape != (M, 4):
        raise ValueError("knn_search_ivf: shape mismatch")
    if centroids.shape != (256, D) or not (0 < nprobe <= 8):
        raise ValueError("knn_search_ivf: centroids must be [256, D], nprobe in [1, 8]")
    if list_rows.numel() != count or list_off.numel() != 257 or list_len.numel() != 256:
        raise ValueError("knn_search_ivf: list arrays do not match count")
    if not (0 < k <= 1024):
        raise ValueError("knn_search_ivf: k must be in [1, 1024]")
    dev = bank.device
    out_s = torch.empty(nq, k, dtype=torch.float32, device=dev)
    out_i = torch.empty(nq, k, dtype=torch.int32, device=dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev) if nq == 0 else _overflow_flag(dev)
    if nq == 0:
        return out_s, out_i, ovf
    L = lib()
    nbytes = L.aura_knn_ivf_workspace_bytes(nq, k, cap)
    if nbytes < 0:
        raise ValueError(f"knn_search_ivf: capacity {cap} is not usable with k={k}")
    ws = _workspace(dev, nbytes)
    base = (ws.data_ptr() + 255) // 256 * 256
    check(L.aura_knn_search_ivf(_p(bank), _p(inv_norm), _p(meta), _p(queries), now, count, D, nq, k,
                                _p(centroids), nprobe, _p(list_rows), _p(list_off), _p(list_len), cap,
                                idx_base, _p(out_s), _p(out_i), base, nbytes, _p(ovf), _stream()),
          "aura_knn_search_ivf")
    return out_s, out_i, ovf


def ivf2_slack(update_interval: int) -> int:
    """Free entries kept behind every inverted list (a multiple of 16): writes between two centroid
    rebuilds (every ``update_interval`` inserts in the reference) are appended in place."""
    return (min(max(int(update_interval), 64), 512) + 15) // 16 * 16


def ivf2_alloc_rows(max_rows: int, slack: int) -> int:
    """Sorted rows to allocate for a bank of ``max_rows``: every list padded to 16 plus its slack."""
    return (int(max_rows) + 256 * (slack + 16) + 15) // 16 * 16


def ivf2_layout(order, seg_off, slack: int, sorted_rows, pad_off, list_len) -> None:
    """Fill the list-sorted layout (in place, no host sync) from rows grouped by centroid id:
    ``order`` int32 [n] (rows of list c = order[seg_off[c] : seg_off[c+1]]), ``seg_off`` int32 [257].
    ``sorted_rows`` int32 [n_alloc] <- row ids, -1 elsewhere; ``pad_off`` int32 [257] <- list starts
    (multiples of 16, ``slack`` free entries behind every list); ``list_len`` int32 [256]."""
    dev = order.device
    off = seg_off.to(torch.int64)
    lens = off[1:] - off[:-1]
    cap = (lens + slack + 15) // 16 * 16
    po = torch.zeros(257, dtype=torch.int64, device=dev)
    po[1:] = torch.cumsum(cap, 0)
    pad_off.copy_(po.to(torch.int32))
    list_len.copy_(lens.to(torch.int32))
    sorted_rows.fill_(-1)
    n = order.numel()                                          # rows before seg_off[0] have no list
    if n:
        pos = torch.arange(n, device=dev, dtype=torch.int64)
        lid = torch.searchsorted(off[1:].contiguous(), pos, right=True).clamp_(max=255)
        dst = po[lid] + (pos - off[lid])
        listed = pos >= off[0]
        sorted_rows[dst[listed]] = order[listed]


def bank_shadow_sorted(bank, inv_norm, sorted_rows, out, rho, pos_of_row=None, n_sorted: Optional[int] = None) -> None:
    """out[i] = bf16 shadow row of bank row sorted_rows[i] (zeros where it is -1) for i < n_sorted;
    refreshes rho[row] and the reverse map pos_of_row[row] = i."""
    _need(bank, "bank", torch.float32); _need(sorted_rows, "sorted_rows", torch.int32)
    _need(inv_norm, "inv_norm", torch.float32); _need(rho, "rho", torch.float32)
    _need(out, "out", torch.bfloat16)
    M, D = bank.shape
    n = sorted_rows.numel() if n_sorted is None else int(n_sorted)
    if D % 8 or out.shape[1] != D or not (0 <= n <= min(sorted_rows.numel(), out.shape[0])) or \
            inv_norm.numel() != M or rho.numel() != M:
        raise ValueError("bank_shadow_sorted: shape mismatch")
    if pos_of_row is not None:
        _need(pos_of_row, "pos_of_row", torch.int32)
        if pos_of_row.numel() != M:
            raise ValueError("bank_shadow_sorted: pos_of_row must have one entry per bank row")
    check(lib().aura_bank_shadow_sorted(_p(bank), _p(inv_norm), _p(sorted_rows), _p(out), _p(rho), _p(pos_of_row),
                                        n, D, _stream()), "aura_bank_shadow_sorted")


def build_ivf2(bank, inv_norm, cids, slack: int = 0, max_rows: Optional[int] = None, rho=None):
    """Inverted lists in the layout of ``knn_search_ivf2`` for rows [0, n) with centroid ids ``cids``
    (any numeric dtype, < 0: no list): returns a dict with sorted_bf16, rho, sorted_rows, pad_off,
    list_len, pos_of_row, flag, n_sorted (sorted rows in use)."""
    dev = bank.device
    M, D = bank.shape
    n = cids.numel()
    order, seg_off = group_by_cluster(cids, 256)
    n_alloc = ivf2_alloc_rows(M if max_rows is None else max_rows, slack)
    st = dict(sorted_bf16=torch.empty(n_alloc, D, dtype=torch.bfloat16, device=dev),
              sorted_rows=torch.empty(n_alloc, dtype=torch.int32, device=dev),
              pad_off=torch.zeros(257, dtype=torch.int32, device=dev),
              list_len=torch.zeros(256, dtype=torch.int32, device=dev),
              pos_of_row=torch.full((M,), -1, dtype=torch.int32, device=dev),
              flag=torch.zeros(1, dtype=torch.int32, device=dev),
              rho=torch.zeros(M, dtype=torch.float32, device=dev) if rho is None else rho,
              n_sorted=min(n_alloc, ivf2_alloc_rows(n, slack)), slack=slack)
    ivf2_layout(order, seg_off, slack, st["sorted_rows"], st["pad_off"], st["list_len"])
    bank_shadow_sorted(bank, inv_norm, st["sorted_rows"], st["sorted_bf16"], st["rho"], st["pos_of_row"],
                       st["n_sorted"])
    return st


def ivf2_append(bank, inv_norm, meta, slots, sorted_shadow, sorted_rows, pad_off, list_len, pos_of_row, rho,
                flag, row_constants=None, row_constants_now: float = 0.0) -> None:
    """Keep the inverted lists current after a write of the DISTINCT rows ``slots`` (int64 [n]): the
    row's old entry becomes a hole, the row is appended to the list of meta[slot][2].  ``row_constants``
    (``ivf2_row_constants`` for ``row_constants_now``): the cached table follows the touched entries."""
    _need(bank, "bank", torch.float32); _need(inv_norm, "inv_norm", torch.float32)
    _need(meta, "meta", torch.float32); _need(slots, "slots", torch.int64)
    _need(sorted_shadow, "sorted_shadow", torch.bfloat16); _need(rho, "rho", torch.float32)
    for t, n_ in ((sorted_rows, "sorted_rows"), (pad_off, "pad_off"), (list_len, "list_len"),
                  (pos_of_row, "pos_of_row"), (flag, "flag")):
        _need(t, n_, torch.int32)
    M, D = bank.shape
    if sorted_shadow.shape[1] != D or sorted_shadow.shape[0] < sorted_rows.numel() or pad_off.numel() != 257 or \
            list_len.numel() != 256 or pos_of_row.numel() != M or rho.numel() != M or meta.shape != (M, 4) or D % 8:
        raise ValueError("ivf2_append: shape mismatch")
    if row_constants is not None:
        _need(row_constants, "row_constants", torch.float32)
        if row_constants.dim() != 2 or row_constants.shape[1] != 4 or row_constants.shape[0] < sorted_rows.numel():
            raise ValueError("ivf2_append: row_constants must be [>= sorted rows, 4]")
    check(lib().aura_ivf2_append(_p(bank), _p(inv_norm), _p(meta), _p(slots), slots.numel(), D, _p(sorted_shadow),
                                 _p(sorted_rows), _p(pad_off), _p(list_len), _p(pos_of_row), _p(rho), _p(flag),
                                 _p(row_constants), float(row_constants_now), _stream()), "aura_ivf2_append")


def ivf2_row_constants(meta, rho, sorted_rows, n_sorted: int, D: int, now: float, out) -> None:
    """out[i] (fp32 [>= n_sorted, 4]) = the score constants of sorted row i at time ``now`` (see
    ``aura_ivf2_row_constants``): pass ``out`` to ``knn_search_ivf2(row_constants=...)`` for calls with the
    same ``now`` while metadata, rho and the list layout are unchanged."""
    _need(meta, "meta", torch.float32); _need(rho, "rho", torch.float32)
    _need(sorted_rows, "sorted_rows", torch.int32); _need(out, "out", torch.float32)
    n = int(n_sorted)
    if out.dim() != 2 or out.shape[1] != 4 or not (0 <= n <= min(out.shape[0], sorted_rows.numel())) or \
            meta.dim() != 2 or meta.shape[1] != 4 or rho.numel() != meta.shape[0]:
        raise ValueError("ivf2_row_constants: shape mismatch")
    check(lib().aura_ivf2_row_constants(_p(meta), _p(rho), _p(sorted_rows), n, int(D), float(now), _p(out), _stream()),
          "aura_ivf2_row_constants")


def centroid_probe(queries, centroids, nprobe: int = 8) -> torch.Tensor:
    """ids [nq, 8] int32: column p < nprobe = the p-th nearest of the 256 centroid rows to each query (L2 on the
    unnormalised query, ties to the lower row, ``hippocampal.py:261-262``) -- the probes ``knn_search_ivf2``
    computes for itself, for callers that want them once per query (``sharded.ShardedHippocampus``)."""
    _need(queries, "queries", torch.float32); _need(centroids, "centroids", torch.float32)
    nq, D = queries.shape
    if centroids.shape != (256, D) or not (0 < nprobe <= 8):
        raise ValueError("centroid_probe: centroids must be [256, D], nprobe in [1, 8]")
    ids = torch.full((nq, 8), -1, dtype=torch.int32, device=queries.device)
    if nq == 0:
        return ids
    L = lib()
    nbytes = L.aura_centroid_probe_workspace_bytes(nq)
    ws = _workspace(queries.device, nbytes)
    base = (ws.data_ptr() + 255) // 256 * 256
    check(L.aura_centroid_probe(_p(centroids), _p(queries), D, nq, nprobe, _p(ids), base, nbytes, _stream()),
          "aura_centroid_probe")
    return ids


def knn_search_ivf2(bank, inv_norm, meta, queries, k: int, now: float, centroids, nprobe: int,
                    sorted_shadow, rho, sorted_rows, pad_off, list_len, idx_base: int = 0,
                    n_sorted: Optional[int] = None, lists_flag=None, probe_ids=None, row_constants=None
                    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Inverted-list recall through the two-stage scan: (scores [nq, k], idx [nq, k], overflow flag [1]).
    Same results as ``knn_search_ivf``; layout arrays from ``ivf2_layout`` + ``bank_shadow_sorted``
    (kept current by ``ivf2_append``).  ``n_sorted``: sorted rows in use (a multiple of 16 that covers
    pad_off[256]; default: all of ``sorted_rows``).  ``lists_flag``: ``ivf2_append``'s flag; if it is
    set the returned overflow flag carries ``KNN_FLAG_LISTS_STALE``.  ``probe_ids``: ``centroid_probe``'s
    output for these queries and this centroid table (the probes are then not recomputed).
    ``row_constants``: ``ivf2_row_constants`` of the current lists for exactly this ``now`` (as fp32)."""
    _need(bank, "bank", torch.float32); _need(inv_norm, "inv_norm", torch.float32)
    _need(meta, "meta", torch.float32); _need(queries, "queries", torch.float32)
    _need(centroids, "centroids", torch.float32); _need(sorted_shadow, "sorted_shadow", torch.bfloat16)
    _need(rho, "rho", torch.float32)
    for t, n in ((sorted_rows, "sorted_rows"), (pad_off, "pad_off"), (list_len, "list_len")):
        _need(t, n, torch.int32)
    M, D = bank.shape
    nq = queries.shape[0]
    ns = sorted_rows.numel() if n_sorted is None else int(n_sorted)
    if queries.dim() != 2 or queries.shape[1] != D or meta.shape != (M, 4) or rho.numel() != M:
        raise ValueError("knn_search_ivf2: shape mismatch")
    if centroids.shape != (256, D) or not (0 < nprobe <= 8):
        raise ValueError("knn_search_ivf2: centroids must be [256, D], nprobe in [1, 8]")
    if sorted_shadow.shape[1] != D or not (0 < ns <= min(sorted_rows.numel(), sorted_shadow.shape[0])) or \
            pad_off.numel() != 257 or list_len.numel() != 256 or ns % 16:
        raise ValueError("knn_search_ivf2: layout arrays do not match")
    if not (0 < k <= 256) or D % 8 or D > 768:
        raise ValueError("knn_search_ivf2: k <= 256, D % 8 == 0, D <= 768")
    dev = bank.device
    out_s = torch.empty(nq, k, dtype=torch.float32, device=dev)
    out_i = torch.empty(nq, k, dtype=torch.int32, device=dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev) if nq == 0 else _overflow_flag(dev)
    if nq == 0:
        return out_s, out_i, ovf
    L = lib()
    nbytes = L.aura_knn_ivf2_workspace_bytes(ns, nq, k)
    ws = _workspace(dev, nbytes)
    base = (ws.data_ptr() + 255) // 256 * 256
    if lists_flag is not None:
        _need(lists_flag, "lists_flag", torch.int32)
    if row_constants is not None:
        _need(row_constants, "row_constants", torch.float32)
        if row_constants.dim() != 2 or row_constants.shape[1] != 4 or row_constants.shape[0] < ns:
            raise ValueError("knn_search_ivf2: row_constants must be [>= n_sorted, 4]")
    if probe_ids is not None:
        _need(probe_ids, "probe_ids", torch.int32)
        if tuple(probe_ids.shape) != (nq, 8):
            raise ValueError("knn_search_ivf2: probe_ids must be [nq, 8] (centroid_probe)")
        check(L.aura_knn_search_ivf2_probed(_p(bank), _p(inv_norm), _p(meta), _p(sorted_shadow), _p(rho),
                                            _p(sorted_rows), _p(pad_off), _p(list_len), _p(lists_flag),
                                            _p(row_constants), ns, M,
                                            _p(queries), now, D, nq, k, _p(centroids), nprobe, _p(probe_ids),
                                            idx_base, _p(out_s), _p(out_i), base, nbytes, _p(ovf), _stream()),
              "aura_knn_search_ivf2_probed")
        return out_s, out_i, ovf
    check(L.aura_knn_search_ivf2(_p(bank), _p(inv_norm), _p(meta), _p(sorted_shadow), _p(rho), _p(sorted_rows),
                                 _p(pad_off), _p(list_len), _p(lists_flag), _p(row_constants), ns, M, _p(queries), now, D,
                                 nq, k,
                                 _p(centroids), nprobe, idx_base, _p(out_s), _p(out_i), base, nbytes,
                                 _p(ovf), _stream()), "aura_knn_search_ivf2")
    return out_s, out_i, ovf


class Ivf2Plan:
    """``knn_search_ivf2`` for ONE list layout, validated once: the per-call path is a shape check of the queries,
    two output allocations and the C call.  (Between a recall's flag read and the next recall's first launch the
    GPU idles; the generic wrapper spends ~30 us of Python there -- a dozen tensor checks, ~25 ``data_ptr()`` calls,
    a workspace-size query -- against a 0.7 ms step.)  Built by ``HippocampalFormation`` after every re-pack of its
    lists; any tensor of the layout being replaced invalidates the plan (the owner drops it)."""

    def __init__(self, bank, inv_norm, meta, centroids, nprobe: int, sorted_shadow, rho, sorted_rows, pad_off, list_len,
                 n_sorted: int, lists_flag, row_constants):
        _need(bank, "bank", torch.float32); _need(inv_norm, "inv_norm", torch.float32)
        _need(meta, "meta", torch.float32); _need(centroids, "centroids", torch.float32)
        _need(sorted_shadow, "sorted_shadow", torch.bfloat16); _need(rho, "rho", torch.float32)
        for t, n in ((sorted_rows, "sorted_rows"), (pad_off, "pad_off"), (list_len, "list_len"), (lists_flag, "lists_flag")):
            _need(t, n, torch.int32)
        _need(row_constants, "row_constants", torch.float32)
        M, D = bank.shape
        ns = int(n_sorted)
        if meta.shape != (M, 4) or rho.numel() != M or centroids.shape != (256, D) or not (0 < nprobe <= 8):
            raise ValueError("Ivf2Plan: shape mismatch")
        if sorted_shadow.shape[1] != D or not (0 < ns <= min(sorted_rows.numel(), sorted_shadow.shape[0])) or \
                pad_off.numel() != 257 or list_len.numel() != 256 or ns % 16 or D % 8 or D > 768:
            raise ValueError("Ivf2Plan: layout arrays do not match")
        if row_constants.dim() != 2 or row_constants.shape[1] != 4 or row_constants.shape[0] < ns:
            raise ValueError("Ivf2Plan: row_constants must be [>= n_sorted, 4]")
        self._keep = (bank, inv_norm, meta, centroids, sorted_shadow, rho, sorted_rows, pad_off, list_len, lists_flag,
                      row_constants)
        self.M, self.D, self.ns, self.nprobe, self.device = M, D, ns, int(nprobe), bank.device
        self._head = tuple(t.data_ptr() for t in (bank, inv_norm, meta, sorted_shadow, rho, sorted_rows, pad_off, list_len,
                                                  lists_flag, row_constants))
        self._cent = centroids.data_ptr()
        self._bytes = {}
        self._fn = lib().aura_knn_search_ivf2_signal
        # completion word: the call's last workgroup stores the flag and a sequence number into host-mapped memory;
        # wait_flag polls it -- no device-to-host copy, no stream synchronisation (aura_knn_search_ivf2_signal)
        hw = ctypes.c_void_p()
        check(lib().aura_host_word_alloc(ctypes.byref(hw)), "aura_host_word_alloc")
        self._hw_ptr = hw.value
        self._hw = (ctypes.c_uint32 * 2).from_address(hw.value)
        self._seq = 0
        self._last_ovf = None

    def __del__(self):
        try:
            if getattr(self, "_hw_ptr", None):
                torch.cuda.synchronize(self.device)          # no launch may still hold the word
                lib().aura_host_word_free(self._hw_ptr)
                self._hw_ptr = None
        except Exception:
            pass

    def wait_flag(self, timeout_s: float = 5.0) -> int:
        """The flag of the last ``run`` (blocks until that call's last launch has finished)."""
        import time as _t
        hw, seq = self._hw, self._seq
        if hw[1] != seq:
            t_end = None
            n = 0
            while hw[1] != seq:
                n += 1
                if (n & 0xfff) == 0:                         # every ~4000 polls: give up after timeout_s
                    now = _t.perf_counter()
                    if t_end is None:
                        t_end = now + timeout_s
                    elif now > t_end:
                        return int(self._last_ovf.item())     # (never seen: the ordinary read still works)
        return int(hw[0])

    def matches(self, bank, meta, centroids, sorted_shadow, n_sorted: int, row_constants) -> bool:
        k = self._keep
        return (k[0] is bank and k[2] is meta and k[3] is centroids and k[4] is sorted_shadow and k[10] is row_constants
                and self.ns == int(n_sorted))

    def run(self, queries, k: int, now: float, probe_ids=None, idx_base: int = 0):
        if not (queries.is_cuda and queries.dtype == torch.float32 and queries.dim() == 2 and queries.shape[1] == self.D
                and queries.is_contiguous()):
            raise ValueError("Ivf2Plan.run: queries must be a contiguous fp32 [nq, D] tensor on the bank's device")
        nq = queries.shape[0]
        if not (0 < k <= 256):
            raise ValueError("Ivf2Plan.run: k <= 256")
        dev = self.device
        out_s = torch.empty(nq, k, dtype=torch.float32, device=dev)
        out_i = torch.empty(nq, k, dtype=torch.int32, device=dev)
        if nq == 0:
            return out_s, out_i, torch.zeros(1, dtype=torch.int32, device=dev)
        nbytes = self._bytes.get((nq, k))
        if nbytes is None:
            nbytes = self._bytes[(nq, k)] = lib().aura_knn_ivf2_workspace_bytes(self.ns, nq, k)
        stream = torch.cuda.current_stream(dev).cuda_stream
        key = (dev, stream)
        ws = _workspaces.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = _workspace(dev, nbytes)
        ovf = _ovf_flags.get(key)
        if ovf is None:
            ovf = _overflow_flag(dev)
        base = (ws.data_ptr() + 255) // 256 * 256
        h = self._head
        pid = None
        if probe_ids is not None:
            _need(probe_ids, "probe_ids", torch.int32)
            if tuple(probe_ids.shape) != (nq, 8):
                raise ValueError("Ivf2Plan.run: probe_ids must be [nq, 8] (centroid_probe)")
            pid = probe_ids.data_ptr()
        self._seq = (self._seq + 1) & 0x7fffffff or 1
        self._last_ovf = ovf
        check(self._fn(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], self.ns, self.M, queries.data_ptr(),
                       now, self.D, nq, k, self._cent, self.nprobe, pid, idx_base, out_s.data_ptr(), out_i.data_ptr(), base,
                       nbytes, ovf.data_ptr(), self._hw_ptr, self._seq, stream), "aura_knn_search_ivf2_signal")
        return out_s, out_i, ovf


class HostFlag:
    """A host-mapped completion word for call chains that end in an entry point without one (the staged recall):
    ``signal(flag_tensor)`` enqueues a one-thread launch that stores the flag and a sequence number;
    ``wait()`` polls it and returns the flag -- no device-to-host copy, no stream synchronisation (a blocking wait
    on a ~1 ms stream costs the host 1-2 ms on this runtime)."""

    def __init__(self, device):
        self.device = torch.device(device)
        hw = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().aura_host_word_alloc(ctypes.byref(hw)), "aura_host_word_alloc")
        self._hw_ptr = hw.value
        self._hw = (ctypes.c_uint32 * 2).from_address(hw.value)
        self._seq = 0
        self._flag = None

    def __del__(self):
        try:
            if getattr(self, "_hw_ptr", None):
                torch.cuda.synchronize(self.device)          # no launch may still hold the word
                lib().aura_host_word_free(self._hw_ptr)
                self._hw_ptr = None
        except Exception:
            pass

    def signal(self, flag: torch.Tensor) -> None:
        _need(flag, "flag", torch.int32)
        self._seq = (self._seq + 1) & 0x7fffffff or 1
        self._flag = flag
        check(lib().aura_signal_flag(_p(flag), self._hw_ptr, self._seq, _stream()), "aura_signal_flag")

    def wait(self, timeout_s: float = 5.0) -> int:
        import time as _t
        hw, seq = self._hw, self._seq
        t_end, n = None, 0
        while hw[1] != seq:
            n += 1
            if (n & 0xfff) == 0:                             # every ~4000 polls: give up after timeout_s
                now = _t.perf_counter()
                if t_end is None:
                    t_end = now + timeout_s
                elif now > t_end:
                    return int(self._flag.item())             # (never seen: the ordinary read still works)
        return int(hw[0])


class Ivf2Staged:
    """``knn_search_ivf2`` in stages (``aura_knn_search_ivf2_staged``) for one pass of at most 8192 queries:
    ``stage1(k2)`` -> bounds [nq, 2] (the k-th and the k2-th largest sampled lower bound of every query on this
    bank), ``stage2(bound [nq])`` -> (scores, idx, overflow flag) with every threshold raised to ``bound`` first --
    or ``stage2_bounds(bound, k2)`` -> the filtered candidates' own bounds [nq, 2] and ``stage3(bound2 [nq])`` -> the
    result re-scored only where a candidate can still reach ``bound2``.  Between the calls the caller combines the
    bounds of all shards of a row-sharded bank; nothing else may use this stream's kNN workspace in between."""

    MAX_QUERIES = 8192

    def __init__(self, bank, inv_norm, meta, queries, k: int, now: float, centroids, nprobe: int,
                 sorted_shadow, rho, sorted_rows, pad_off, list_len, idx_base: int = 0,
                 n_sorted: Optional[int] = None, lists_flag=None, probe_ids=None, row_constants=None,
                 out=None, validated: bool = False):
        """``out``: (scores [nq, k] fp32, idx [nq, k] int32) to write into (contiguous; e.g. slices of the caller's
        result); ``validated``: the caller has run this constructor's checks on the same tensors before (a pass of a
        multi-pass recall, or the next call on an unchanged bank) -- only the per-call shapes are checked."""
        if validated:
            self._init_fast(bank, inv_norm, meta, queries, k, now, centroids, nprobe, sorted_shadow, rho, sorted_rows,
                            pad_off, list_len, idx_base, n_sorted, lists_flag, probe_ids, row_constants, out)
            return
        _need(bank, "bank", torch.float32); _need(inv_norm, "inv_norm", torch.float32)
        _need(meta, "meta", torch.float32); _need(queries, "queries", torch.float32)
        _need(centroids, "centroids", torch.float32); _need(sorted_shadow, "sorted_shadow", torch.bfloat16)
        _need(rho, "rho", torch.float32)
        for t, n in ((sorted_rows, "sorted_rows"), (pad_off, "pad_off"), (list_len, "list_len")):
            _need(t, n, torch.int32)
        M, D = bank.shape
        nq = queries.shape[0]
        ns = sorted_rows.numel() if n_sorted is None else int(n_sorted)
        if queries.dim() != 2 or queries.shape[1] != D or meta.shape != (M, 4) or rho.numel() != M or \
                not (0 < nq <= self.MAX_QUERIES):
            raise ValueError("Ivf2Staged: shape mismatch (1..8192 queries per staged pass)")
        if centroids.shape != (256, D) or not (0 < nprobe <= 8) or not (0 < k <= 256) or D % 8 or D > 768:
            raise ValueError("Ivf2Staged: centroids [256, D], nprobe <= 8, k <= 256, D % 8 == 0, D <= 768")
        if sorted_shadow.shape[1] != D or not (0 < ns <= min(sorted_rows.numel(), sorted_shadow.shape[0])) or \
                pad_off.numel() != 257 or list_len.numel() != 256 or ns % 16:
            raise ValueError("Ivf2Staged: layout arrays do not match")
        if lists_flag is not None:
            _need(lists_flag, "lists_flag", torch.int32)
        if row_constants is not None:
            _need(row_constants, "row_constants", torch.float32)
            if row_constants.dim() != 2 or row_constants.shape[1] != 4 or row_constants.shape[0] < ns:
                raise ValueError("Ivf2Staged: row_constants must be [>= n_sorted, 4]")
        if probe_ids is not None:
            _need(probe_ids, "probe_ids", torch.int32)
            if tuple(probe_ids.shape) != (nq, 8):
                raise ValueError("Ivf2Staged: probe_ids must be [nq, 8]")
        self._init_fast(bank, inv_norm, meta, queries, k, now, centroids, nprobe, sorted_shadow, rho, sorted_rows, pad_off,
                        list_len, idx_base, ns, lists_flag, probe_ids, row_constants, out)

    def _init_fast(self, bank, inv_norm, meta, queries, k, now, centroids, nprobe, sorted_shadow, rho, sorted_rows,
                   pad_off, list_len, idx_base, n_sorted, lists_flag, probe_ids, row_constants, out):
        M, D = bank.shape
        nq = queries.shape[0]
        ns = sorted_rows.numel() if n_sorted is None else int(n_sorted)
        if not (queries.is_cuda and queries.dtype == torch.float32 and queries.dim() == 2 and queries.shape[1] == D
                and queries.is_contiguous() and 0 < nq <= self.MAX_QUERIES):
            raise ValueError("Ivf2Staged: queries must be a contiguous fp32 [1..8192, D] tensor on the bank's device")
        if probe_ids is not None and not (probe_ids.is_cuda and probe_ids.dtype == torch.int32 and
                                          tuple(probe_ids.shape) == (nq, 8) and probe_ids.is_contiguous()):
            raise ValueError("Ivf2Staged: probe_ids must be a contiguous int32 [nq, 8] tensor")
        dev = bank.device
        self._keep = (bank, inv_norm, meta, queries, centroids, sorted_shadow, rho, sorted_rows, pad_off, list_len,
                      lists_flag, probe_ids, row_constants)
        self.nq, self.k = nq, int(k)
        if out is None:
            self.out_s = torch.empty(nq, k, dtype=torch.float32, device=dev)
            self.out_i = torch.empty(nq, k, dtype=torch.int32, device=dev)
        else:
            self.out_s, self.out_i = out
            if not (self.out_s.is_contiguous() and self.out_i.is_contiguous() and tuple(self.out_s.shape) == (nq, k)
                    and tuple(self.out_i.shape) == (nq, k) and self.out_s.dtype == torch.float32
                    and self.out_i.dtype == torch.int32 and self.out_s.device == dev and self.out_i.device == dev):
                raise ValueError("Ivf2Staged: out must be contiguous (fp32 [nq, k], int32 [nq, k]) on the bank's device")
        self.ovf = _overflow_flag(dev)
        L = lib()
        self._nbytes = L.aura_knn_ivf2_workspace_bytes(ns, nq, k)
        self._ws = _workspace(dev, self._nbytes)
        base = (self._ws.data_ptr() + 255) // 256 * 256
        # (a plain tuple, not a closure over self: a self-referencing lambda made every staged pass a reference cycle
        #  that only the cyclic collector freed -- with its 2-MB result views -- and the allocator answered the pile-up
        #  with fresh hipMallocs: recalls of 1.2 ms read 1.6-3.4 ms at random)
        self._args = (_p(bank), _p(inv_norm), _p(meta), _p(sorted_shadow), _p(rho), _p(sorted_rows), _p(pad_off),
                      _p(list_len), _p(lists_flag), _p(row_constants), ns, M, _p(queries), now, D, nq, k, _p(centroids),
                      nprobe, _p(probe_ids), idx_base, _p(self.out_s), _p(self.out_i), base, self._nbytes, _p(self.ovf))

    def _call(self, stage: int, k2: int, bounds) -> None:
        check(lib().aura_knn_search_ivf2_staged(*self._args, stage, k2, _p(bounds), _stream()),
              "aura_knn_search_ivf2_staged")

    def stage1(self, k2: int = 0) -> torch.Tensor:
        b = torch.empty(self.nq, 2, dtype=torch.float32, device=self.out_s.device)
        self._call(1, int(k2), b)
        return b

    def stage2(self, bound: torch.Tensor):
        _need(bound, "bound", torch.float32)
        if bound.numel() != self.nq:
            raise ValueError("Ivf2Staged.stage2: one bound per query")
        self._call(2, 0, bound)
        return self.out_s, self.out_i, self.ovf

    def stage2_bounds(self, bound: torch.Tensor, k2: int) -> torch.Tensor:
        """Stage 2 up to the filter scan, then the CANDIDATES' bounds [nq, 2] (the k-th and the k2-th largest lower
        bound among this bank's candidates, -inf where there are fewer) instead of the refine: for a second, much
        tighter combination over the shards, consumed by ``stage3``."""
        _need(bound, "bound", torch.float32)
        if bound.numel() != self.nq:
            raise ValueError("Ivf2Staged.stage2_bounds: one bound per query")
        b = torch.empty(self.nq, 2, dtype=torch.float32, device=self.out_s.device)
        b.view(-1)[: self.nq].copy_(bound.reshape(-1))     # in: one bound per query; out: [nq, 2]
        self._call(4, int(k2), b)
        return b

    def stage3(self, bound: torch.Tensor):
        """The refine against a lower bound of every query's k-th best score over ALL shards: candidates whose
        upper bound stays below it are not re-scored (rows of this bank's top k that cannot be in the global top k
        come back as -1)."""
        _need(bound, "bound", torch.float32)
        if bound.numel() != self.nq:
            raise ValueError("Ivf2Staged.stage3: one bound per query")
        self._call(3, 0, bound)
        return self.out_s, self.out_i, self.ovf


def topk_merge(scores, idx, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """scores, idx: [S, nq, k] per-shard lists -> merged (scores [nq, k], idx [nq, k])."""
    _need(scores, "scores", torch.float32); _need(idx, "idx", torch.int32)
    S, nq, kk = scores.shape
    if idx.shape != scores.shape or kk != k:
        raise ValueError("topk_merge: shape mismatch")
    out_s = torch.empty(nq, k, dtype=torch.float32, device=scores.device)
    out_i = torch.empty(nq, k, dtype=torch.int32, device=scores.device)
    check(lib().aura_topk_merge(_p(scores), _p(idx), S, nq, k, _p(out_s), _p(out_i), _stream()),
          "aura_topk_merge")
    return out_s, out_i


def bank_gather(bank, idx) -> torch.Tensor:
    """rows of ``bank`` at int32 ``idx`` (any shape); indices outside the bank (-1) -> zeros."""
    _need(bank, "bank", torch.float32); _need(idx, "idx", torch.int32)
    D = bank.shape[1]
    n = idx.numel()
    out = torch.empty(*idx.shape, D, dtype=torch.float32, device=bank.device)
    check(lib().aura_bank_gather(_p(bank), bank.shape[0], _p(idx), _p(out), n, D, _stream()), "aura_bank_gather")
    return out


def kmeans_assign(bank, centroids, count: int, k: int) -> torch.Tensor:
    """Nearest of the first k centroids for rows [0, count) -> int32 [count]."""
    _need(bank, "bank", torch.float32); _need(centroids, "centroids", torch.float32)
    M, D = bank.shape
    if not (0 <= count <= M) or centroids.shape[1] != D or not (0 < k <= min(256, centroids.shape[0])):
        raise ValueError("kmeans_assign: shape mismatch")
    assign = torch.empty(count, dtype=torch.int32, device=bank.device)
    ws = torch.empty(256, dtype=torch.float32, device=bank.device)
    check(lib().aura_kmeans_assign(_p(bank), _p(centroids), _p(ws), _p(assign), count, D, k,
                                   _stream()), "aura_kmeans_assign")
    return assign


def group_by_cluster(assign, k: int = 256) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rows grouped by cluster id (stable): (order int32 [n], seg_off int32 [k+1]) with
    order[seg_off[c] : seg_off[c+1]] = rows of cluster c; ids < 0 sort first (seg_off[0] = their
    count).  Device-side sort + bincount, no host sync."""
    a = assign.to(torch.int32)
    order = torch.sort(a, stable=True).indices.to(torch.int32).contiguous()
    valid = a >= 0
    lens = torch.bincount(a.clamp(min=0).long(), weights=valid.to(torch.float32), minlength=k)[:k].to(torch.int64)
    n_neg = (a.numel() - valid.sum()).reshape(1)
    seg_off = torch.cat([n_neg, n_neg + torch.cumsum(lens, 0)]).to(torch.int32).contiguous()
    return order, seg_off


def kmeans_segment_means(bank, order, seg_off, centroids, k: int, sums_only: bool = False) -> None:
    """centroids[c] = mean of bank[order[seg_off[c]:seg_off[c+1]]] for c < k (empty clusters keep theirs);
    ``sums_only``: the sums instead (zeros for empty clusters) -- a shard's partial result."""
    _need(bank, "bank", torch.float32); _need(centroids, "centroids", torch.float32)
    _need(order, "order", torch.int32); _need(seg_off, "seg_off", torch.int32)
    M, D = bank.shape
    N = order.numel()
    if N > M or centroids.shape[1] != D or not (0 < k <= min(256, centroids.shape[0])) or seg_off.numel() < k + 1 or D % 4:
        raise ValueError("kmeans_segment_means: shape mismatch")
    L = lib()
    nbytes = L.aura_kmeans_means_workspace_bytes(N, D, k)
    ws = _workspace(bank.device, nbytes)
    base = (ws.data_ptr() + 255) // 256 * 256
    check(L.aura_kmeans_segment_means(_p(bank), _p(order), _p(seg_off), _p(centroids), base, nbytes, N, D, k,
                                      1 if sums_only else 0, _stream()), "aura_kmeans_segment_means")


def kmeans_commit(assign, seg_off, meta, counts, k: int) -> None:
    """meta[i][2] = assign[i]; counts[c] = rows of cluster c."""
    _need(assign, "assign", torch.int32); _need(seg_off, "seg_off", torch.int32); _need(meta, "meta", torch.float32)
    N = assign.numel()
    if meta.shape[0] < N or meta.shape[1] != 4 or seg_off.numel() < k + 1 or not (0 < k <= 256):
        raise ValueError("kmeans_commit: shape mismatch")
    if counts is not None:
        _need(counts, "counts", torch.float32)
        if counts.numel() < k:
            raise ValueError("kmeans_commit: counts too small")
    check(lib().aura_kmeans_commit(_p(assign), _p(seg_off), _p(meta), _p(counts), N, k, _stream()),
          "aura_kmeans_commit")


def kmeans_update(bank, assign, centroids, k: int, counts=None, meta=None, update_means: bool = True):
    """One Lloyd update from an assignment: means (``update_means``), counts and metadata ids.
    Returns the grouping (order, seg_off) it used."""
    order, seg_off = group_by_cluster(assign, k)
    if update_means:
        kmeans_segment_means(bank, order, seg_off, centroids, k)
    if meta is not None:
        kmeans_commit(assign, seg_off, meta, counts, k)
    elif counts is not None:
        counts[:k] = (seg_off[1:k + 1] - seg_off[:k]).to(torch.float32)
    return order, seg_off


def addition_linear(x, weight_patterns, bias=None) -> torch.Tensor:
    """-||w_o - x_b||_1 (+ bias): x [B, in], weight_patterns [out, in] -> [B, out]."""
    _need(x, "x", torch.float32); _need(weight_patterns, "weight_patterns", torch.float32)
    if x.dim() != 2 or weight_patterns.dim() != 2 or x.shape[1] != weight_patterns.shape[1]:
        raise ValueError("addition_linear: x must be [B, in] and weights [out, in]")
    if bias is not None:
        _need(bias, "bias", torch.float32)
        if bias.numel() != weight_patterns.shape[0]:
            raise ValueError("addition_linear: bias shape mismatch")
    out = torch.empty(x.shape[0], weight_patterns.shape[0], dtype=torch.float32, device=x.device)
    check(lib().aura_addition_linear(_p(x), _p(weight_patterns), _p(bias), _p(out), x.shape[0],
                                     x.shape[1], weight_patterns.shape[0], _stream()),
          "aura_addition_linear")
    return out


def addition_linear_backward(x, weight_patterns, g_out, need_x: bool = True, need_w: bool = True):
    """Gradients of ``addition_linear``: (g_x [B, in] or None, g_w [out, in] or None); abs -> sign, sign(0) = 0."""
    _need(x, "x", torch.float32); _need(weight_patterns, "weight_patterns", torch.float32)
    _need(g_out, "g_out", torch.float32)
    B, IN = x.shape
    OUT = weight_patterns.shape[0]
    if weight_patterns.shape[1] != IN or tuple(g_out.shape) != (B, OUT):
        raise ValueError("addition_linear_backward: shape mismatch")
    g_x = torch.empty_like(x) if need_x else None
    g_w = torch.empty_like(weight_patterns) if need_w else None
    check(lib().aura_addition_linear_backward(_p(x), _p(weight_patterns), _p(g_out), _p(g_x), _p(g_w), B, IN, OUT,
                                              _stream()), "aura_addition_linear_backward")
    return g_x, g_w
