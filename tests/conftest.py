import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (runs the HIP path through the C-ABI)")
    # the server networks time their tile configurations per (layer, shape) at bind: one file for the whole run, so that the tests'
    # many handles and child processes time a layer once (OCR_SRV_TUNE_FILE; every configuration gives the same bits)
    if "OCR_SRV_TUNE_FILE" not in os.environ:
        import tempfile
        os.environ["OCR_SRV_TUNE_FILE"] = os.path.join(tempfile.gettempdir(), "ocr_srv_tune_%d.txt" % os.getpid())


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def built():
    """The CPU suite needs the oracle (and the C-ABI .so for the export test): build once if missing."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])
    if not os.path.exists(os.path.join(ROOT, "cpp-paddle-ocr_amd", "lib", "libocr_hip.so")):
        from __graft_entry__ import build
        build()
    import synth_weights
    synth_weights.ensure(ROOT)
    return True


@pytest.fixture(scope="session")
def card():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "card_jd_bgr.npy"))
