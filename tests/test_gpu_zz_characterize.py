"""GPU parity of the training-side histogramming (second half): ns_cs_histograms (k_cs_hist, one alignment per thread)
against the oracle's two-list restatement of src/besthit_to_histogram.py:hist() and against the files the reference itself wrote
(tests/golden/reference_hist.json.gz).  (The file sorts behind the other -m gpu files on purpose: it is the newest kernel of the engine — and for the same reason the file runs
in a CHILD pytest first: a device fault or a hang there fails this file with the child's output, not the whole -m gpu run.)"""
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from nanosim_amd import characterize
from nanosim_amd import engine as E
from tests import oracle_lib as O
from tests.test_characterize import same_counts

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fx():
    with gzip.open(os.path.join(ROOT, "tests", "golden", "reference_hist.json.gz"), "rt") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def child_ok():
    if os.environ.get("NS_CSH_CHILD"):
        return
    try:
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT,
                           env=dict(os.environ, NS_CSH_CHILD="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    except subprocess.TimeoutExpired as ex:
        pytest.fail("the child run of this file did not finish in 900 s:\n" + str(ex.stdout or "")[-3000:])
    if r.returncode != 0:
        pytest.fail("the child run of this file failed (exit %d):\n%s" % (r.returncode, r.stdout[-4000:]))


@pytest.fixture(scope="module")
def eng(child_ok):
    e = E.Engine(0)
    yield e
    e.close()


def test_gpu_counts_equal_the_oracle_and_the_reference_files(fx, eng, tmp_path):
    t = characterize.count(eng, fx["cs"], cap=256)            # (256 < the fixture's longest match: the matrix is sized again)
    same_counts(t, O.cs_hist(fx["cs"]))
    prefix = str(tmp_path / "training_genome")
    characterize.hist(prefix, fx["cs"], eng)                  # B:150-151: "_genome" is cut off the prefix
    for name, text in fx["files"].items():
        assert open(str(tmp_path / "training") + name).read() == text, name


def test_gpu_edge_cases_and_many_alignments(fx, eng):
    for cs in ([":5*ag:3", "*ac:7", ":2"], [":9-a", "+g:4", "-t", "*ac*gt", ":3+a:2"], ["", "*ag", "", "+a-c*gt", ":1"], ["garbage", ":12", "::7*a1*ab:3"]):
        same_counts(characterize.count(eng, cs, cap=64), O.cs_hist(cs, cap=64))
    with pytest.raises(ValueError):
        characterize.count(eng, [":4=ACG:2"])
    assert characterize.count(eng, [])["dic"].sum() == 0
    # 40 x the fixture, shuffled: several workgroups, the carry of prev_match across alignments that start with an error
    rng = np.random.default_rng(3)
    many = [fx["cs"][i] for i in rng.integers(0, len(fx["cs"]), 40 * len(fx["cs"]))] + ["*ag:3", "+c", "-g*ac:9"]
    t = characterize.count(eng, many)
    same_counts(t, O.cs_hist(many))
    assert t["ms_kernel"] > 0


def test_gpu_maf_counts_equal_the_oracle_and_the_reference_files(eng, tmp_path):
    """the MAF branch (B:188-315) through ns_maf_histograms: GPU == oracle == the files the REAL hist(prefix, "maf") wrote"""
    import gzip
    import json
    with gzip.open(os.path.join(ROOT, "tests", "golden", "reference_hist_maf.json.gz"), "rt") as f:
        fxm = json.load(f)
    pairs = [tuple(p) for p in fxm["maf"]]
    same_counts(characterize.count_maf(eng, pairs, cap=256), O.maf_hist(pairs))
    maf = tmp_path / "training_besthit.maf"
    with open(maf, "w") as f:
        for i, (r, q) in enumerate(pairs):
            f.write("s ref 0 %d + 1000000 %s\ns read%d 0 %d + %d %s\n" % (len(r), r, i, len(q), len(q), q))
    characterize.hist(str(tmp_path / "training"), characterize.maf_pairs(str(maf)), eng, alnm_ftype="maf")
    for name, text in fxm["files"].items():
        assert open(str(tmp_path / "training") + name).read() == text, name
    rng = np.random.default_rng(9)
    many = [pairs[i] for i in rng.integers(0, len(pairs), 20 * len(pairs))] + [("", ""), ("A-C", "AT-")]
    same_counts(characterize.count_maf(eng, many), O.maf_hist(many))
