import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("LB_ALLOW_SYNTHETIC", "1")     # tests run on seeded synthetic weights / embeddings by design
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size SDXL models against the CPU fp32 oracle (minutes of host time); "
                                       "`-m \"gpu and not slow\"` is the quick loop, the driver runs everything")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def results_log():
    """Parity metrics collected across GPU tests, dumped to gpurun_out/parity_metrics.json."""
    import json
    log = {}
    yield log
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_metrics.json")
    merged = {}
    if os.path.isfile(path):                 # several pytest invocations of one GPU call add up
        try:
            with open(path) as fh:
                merged = json.load(fh)
        except Exception:
            merged = {}
    merged.update(log)
    with open(path, "w") as fh:
        json.dump(merged, fh, indent=1, sort_keys=True)
