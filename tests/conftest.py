"""pytest configuration: import paths, the `gpu` marker, shared fixtures."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
for p in (str(ROOT), str(ROOT / "kosmos-x_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
