"""Loader for the upstream reference modules (TEST INFRASTRUCTURE ONLY).

Used by ``oracle/gen_golden.py`` and ``tests/test_oracle_vs_reference.py`` in the build
container, where the reference checkout is mounted read-only at ``/root/reference``.  It never
runs on the GPU box (the reference does not travel) and is never imported by the product package.

The reference's package ``__init__`` files pull in modules that need Python >= 3.12
(``src/base/events.py:6`` is a PEP 695 ``type`` statement), so a plain ``import core.hippocampal``
raises SyntaxError on this image's Python 3.10.  The hot-path files themselves are 3.10-clean.
We therefore register empty namespace packages whose ``__path__`` points at the real directories
and let the import system load the hot-path files unmodified.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("AURA_REFERENCE_ROOT", "/root/reference")

_NAMESPACES = {
    "src": "src",
    "src.core": "src/core",
    "src.core.language_zone": "src/core/language_zone",
    "src.base": "src/base",
    "src.maths": "src/maths",
    "core": "src/core",
    "core.language_zone": "src/core/language_zone",
    "base": "src/base",
    "maths": "src/maths",
}


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "src", "core"))


class _EventBusStandIn:
    """Harness-owned stand-in for ``src/base/events.py:20`` (not loadable on 3.10)."""

    def __init__(self, *a, **k):
        self.events = []

    def subscribe(self, *a, **k):
        return None

    def broadcast_neuron_fired(self, payload):
        self.events.append(payload)


def install() -> None:
    """Register the namespace stubs (idempotent)."""
    if not available():
        raise RuntimeError(f"reference checkout not found at {REF_ROOT}")
    sys.dont_write_bytecode = True  # the reference mount is read-only
    for name, rel in _NAMESPACES.items():
        if name in sys.modules:
            continue
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF_ROOT, rel)]
        mod.__package__ = name
        sys.modules[name] = mod
    for name in ("src.base.events", "base.events"):
        if name not in sys.modules:
            ev = types.ModuleType(name)
            ev.EventBus = _EventBusStandIn
            sys.modules[name] = ev


def load(name: str):
    """Import one reference module by dotted name, e.g. ``core.hippocampal``."""
    install()
    return importlib.import_module(name)
