import json
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
def golden_functions():
    with open(os.path.join(GOLDEN, "reference_functions.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_samplers():
    with open(os.path.join(GOLDEN, "reference_samplers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_distributions():
    with open(os.path.join(GOLDEN, "reference_distributions.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def small_model():
    from nanosim_amd import model
    return model.load_model(os.path.join(GOLDEN, "model_small", "training"), chimeric=True, homopolymer=True,
                            fastq=True)


@pytest.fixture(scope="session")
def small_ref():
    from nanosim_amd import model
    return model.read_fasta(os.path.join(GOLDEN, "genome_small.fa"), "linear")


@pytest.fixture(scope="session")
def circ_ref():
    from nanosim_amd import model
    return model.read_fasta(os.path.join(GOLDEN, "genome_circ.fa"), "circular")
