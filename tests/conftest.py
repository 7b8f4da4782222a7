import os
import sys

import pytest
# torch first: it bundles its own HIP runtime, and whichever copy of libamdhip64 is loaded first serves the whole process.
# With libstorm_hip loaded first (ROCm's copy), a later torch.cuda initialisation reports "No HIP GPUs are available".
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def known_answers():
    import json
    with open(os.path.join(GOLDEN, "reference_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import storm_ref
    storm_ref.build()
    storm_ref.lib()
    return storm_ref


@pytest.fixture(scope="session")
def oracle_models(oracle):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = oracle.Model.from_gsm_npz(os.path.join(GOLDEN, f"gsm_{name}.npz"))
        return cache[name]

    return get
