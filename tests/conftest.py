import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_group(npz, prefix):
    """Sub-dict of an npz whose keys start with `prefix/` (prefix stripped)."""
    pre = prefix + "/"
    return {k[len(pre):]: npz[k] for k in npz.files if k.startswith(pre)}


def g3_case(tag):
    """(npz, main0, target0) of a G3 learn() golden. The big-batch cases (BASELINE configs[3], [4]: 'xarm1024',
    'panda2048', tests/golden/g3_learn_big.npz) start from the initial weights of the small case with the same
    (S, A, seed), which make_golden.py asserted when it wrote them; the target starts equal to the main net."""
    import numpy as np
    small = np.load(os.path.join(GOLDEN, "g3_learn.npz"))
    if tag in ("kuka", "panda"):
        return small, load_group(small, f"{tag}/main0"), load_group(small, f"{tag}/target0")
    if tag == "h128":       # NAF(10, 5, 128), batch 64: the network of the reference's own agent test (make_golden.py --only g3h128)
        g = np.load(os.path.join(GOLDEN, "g3_learn_h128.npz"))
        return g, load_group(g, "h128/main0"), load_group(g, "h128/target0")
    big = np.load(os.path.join(GOLDEN, "g3_learn_default.npz" if tag in G3_DEFAULT_TAGS else "g3_learn_big.npz"))
    init = str(big[f"{tag}/init_of"])
    return big, load_group(small, f"{init}/main0"), load_group(small, f"{init}/target0")


# 'kuka64' / 'kuka128' / 'kuka192': the reference's learn() at configs[0]'s batch, at its default batch (rl_framework.py:68-74)
# and at three 64-row blocks (tests/golden/g3_learn_default.npz, make_golden.py --only g3def)
G3_DEFAULT_TAGS = ["kuka64", "kuka128", "kuka192"]
G3_TAGS = ["kuka", "panda", "xarm1024", "panda2048"] + G3_DEFAULT_TAGS
G3_ORACLE_TAGS = G3_TAGS + ["h128"]      # (+ layer_size 128: the oracle is pinned there too; the GPU test of it is its own)
