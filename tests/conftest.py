import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden_known():
    return load_golden("known_answers.json")


@pytest.fixture(scope="session")
def golden_small():
    return load_golden("small_layers.json")


@pytest.fixture(scope="session")
def golden_medium():
    return load_golden("medium_traces.json")


@pytest.fixture(scope="session")
def golden_cli():
    return load_golden("cli_outputs.json")
