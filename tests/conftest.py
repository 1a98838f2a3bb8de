import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fb_dir(tmp_path_factory):
    """FB15k-237-shaped synthetic graph (generated, never committed)."""
    from openkeonspark_amd.synthetic import make_dataset, FB15K237
    return make_dataset("/tmp/okes_fb15k237_shaped", FB15K237)


@pytest.fixture(scope="session")
def wn_dir(tmp_path_factory):
    """WN18RR-shaped synthetic graph (BASELINE config #3)."""
    from openkeonspark_amd.synthetic import make_dataset, WN18RR
    return make_dataset("/tmp/okes_wn18rr_shaped", WN18RR)


def parity_report(test, **metrics):
    """Print (and, where gpurun_out/ is writable, log) what a tolerance carve-out actually observed, so that a
    regression inside the allowed bound stays visible: every '<= N rows may differ' assert reports its count."""
    import json
    line = json.dumps(dict(test=test, **{k: (float(v) if hasattr(v, "dtype") else v) for k, v in metrics.items()}))
    print("[parity] " + line)
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_counts.jsonl"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
