"""Fixture provenance as a check: every ``tests/golden/gen_*.py`` is re-run against ``/root/reference`` into a scratch
copy of the tree and every array it writes must equal the committed ``.npz`` bit for bit.  The generators import the reference's own
code by path (stand-in modules provide third-party INTERFACES only), so "pinned to the reference" is then a property the suite
re-establishes whenever the reference is present -- in the build container; on the GPU box (no /root/reference) the test is skipped."""
import glob
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"
GENERATORS = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, "gen_*.py")))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "competitive_rl")),
                                reason="the reference tree is only present in the build container")


@pytest.fixture(scope="module")
def scratch(tmp_path_factory):
    """<tmp>/tests/golden holds copies of the generators and their stand-ins (they write beside themselves); everything else of the
    repo -- the checker, the package's assets -- is reached through links, so the scratch tree never shadows or edits the real one."""
    top = tmp_path_factory.mktemp("regen")
    for name in os.listdir(ROOT):
        if name in ("tests", ".git", "gpurun_out", ".pytest_cache", "__pycache__"):
            continue
        os.symlink(os.path.join(ROOT, name), os.path.join(top, name))
    os.makedirs(os.path.join(top, "tests", "golden"))
    for p in glob.glob(os.path.join(ROOT, "tests", "*.py")):
        shutil.copy(p, os.path.join(top, "tests"))
    for p in glob.glob(os.path.join(GOLD, "*.py")):
        shutil.copy(p, os.path.join(top, "tests", "golden"))
    # all generators at once (they are independent single-threaded scripts; the longest takes ~75 s): the module costs the longest one
    out_dir = os.path.join(top, "tests", "golden")
    procs = {g: subprocess.Popen([sys.executable, os.path.join(out_dir, g)], cwd=str(top), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                 env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1"))
             for g in GENERATORS}
    yield str(top), procs
    for pr in procs.values():
        if pr.poll() is None:
            pr.kill()


def test_every_committed_fixture_has_a_generator():
    written = set()
    for g in GENERATORS:
        src = open(os.path.join(GOLD, g)).read()
        written |= {os.path.basename(p) for p in glob.glob(os.path.join(GOLD, "*.npz")) if os.path.basename(p) in src}
    assert written == {os.path.basename(p) for p in glob.glob(os.path.join(GOLD, "*.npz"))}, "a fixture no generator names"


@pytest.mark.parametrize("gen", GENERATORS)
def test_generator_reproduces_the_committed_arrays(gen, scratch):
    top, procs = scratch
    out_dir = os.path.join(top, "tests", "golden")
    stdout, stderr = procs[gen].communicate(timeout=1500)
    assert procs[gen].returncode == 0, (gen, stdout[-1500:], stderr[-3000:])
    src = open(os.path.join(GOLD, gen)).read()
    made = sorted(p for p in glob.glob(os.path.join(out_dir, "*.npz")) if os.path.basename(p) in src)  # (the fixtures this generator names)
    assert made, (gen, "wrote no fixture")
    for path in made:
        name = os.path.basename(path)
        committed = os.path.join(GOLD, name)
        assert os.path.exists(committed), (gen, name, "is not a committed fixture")
        new, old = np.load(path), np.load(committed)
        assert sorted(new.files) == sorted(old.files), (name, sorted(set(new.files) ^ set(old.files)))
        for k in old.files:
            a, b = new[k], old[k]
            assert a.dtype == b.dtype and a.shape == b.shape, (name, k, a.dtype, b.dtype, a.shape, b.shape)
            same = np.array_equal(a, b) if a.dtype.kind in "OSU" else a.tobytes() == b.tobytes()
            assert same, (name, k, "regenerated array differs from the committed one")
