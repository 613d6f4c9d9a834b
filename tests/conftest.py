import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library must exist before any test: build it if the tree is fresh."""
    lib = os.path.join(ROOT, "aura_snn_rag_amd", "lib", "libaura_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """One line with the index-exact counts the parity tests recorded (tests/helpers.record_parity)."""
    try:
        from tests.helpers import PARITY
    except Exception:
        return
    if PARITY:
        import json
        short = {k: f"{v['exact']}/{v['n']}" + (f" (+{v['score_near_ties']} score near-ties)" if v['score_near_ties'] else "")
                    + (f" ({v['probe_near_ties_excluded']} probe near-ties excluded)" if v['probe_near_ties_excluded'] else "")
                 for k, v in sorted(PARITY.items())}
        terminalreporter.write_line("PARITY_COUNTS index-exact queries vs the CPU oracle: " + json.dumps(short))
