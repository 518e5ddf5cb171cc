"""The randomised parity campaign of tools/fuzz_parity.py under pytest: fixed seeds, a fixed number
of cases per kernel family (random shapes, strides, layouts, parameters, thresholds, step changes,
zeros / NaNs / infinities / negative determinants), the HIP path through the C ABI against the CPU
oracle.  Bit-exact change maps, 1e-5 relative for the float outputs; the ill-posed n_eff corner of
non-local means is recognised from the oracle alone and not compared, as in the tool.
About forty seconds in all."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 20261004
# (sized for ~40 s in all: the suite runs under a 900 s limit.  The long campaign is tools/fuzz_parity.py,
#  its result per round under profiles/ -- r06: 5 248 omnibus / c3 cases in 240 s, no difference)
CASES_PER_FAMILY = {'omnibus': 250, 'omnibus_ml': 300, 'c3': 400, 'nlmeans': 2500, 'correlate': 5000, 'gaussian': 6000}


@pytest.fixture(scope='module')
def fuzz(device, oracle):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import fuzz_parity
    return fuzz_parity


@pytest.mark.parametrize('family', sorted(CASES_PER_FAMILY))
def test_fuzz_family(fuzz, family):
    fails = []
    for i in range(CASES_PER_FAMILY[family]):
        rng = np.random.default_rng([SEED, i])
        try:
            ok, desc = fuzz.CASES[family](rng)
        except Exception as e:              # noqa: BLE001  an error is a failure too
            ok, desc = False, {'exception': repr(e)}
        if not ok:
            fails.append((i, desc))
    assert not fails, '%d of %d %s cases differ from the oracle; replay with tools/fuzz_replay.py: %s' % (
        len(fails), CASES_PER_FAMILY[family], family, fails[:5])
