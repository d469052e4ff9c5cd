"""Long-run whole-model parity at real frame sizes (tools/state_check.py), last in the GPU suite: the pipelined path --
two frames a launch -- for a hundred frames and more, then further frames through the traffic-audit instantiations of
the per-pixel kernel; every position and the WHOLE model against the oracle, bit for bit."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import oat_amd
    return oat_amd


def test_long_run_model_parity_with_audited_steps(A):
    """The pipelined path at a real frame size, full occupancy: 120 frames of one 1080p SURVEY-8d stream through
    enqueue / collect (ring 4), then six more through the traffic-audit instantiation of the per-pixel kernel; every
    position and the WHOLE model (counters, weights, variances, means) must be the oracle's, bit for bit.  (The
    short model-parity sequences of test_gpu_parity.py run small frames; a fault of round 2 -- the audit instantiation updating the
    model wrongly after a change of a load's type -- showed only here, tools/state_check.py.)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import state_check
    msgs = []
    assert state_check.run(1080, 1920, 120, 24, audited=6, log=msgs.append) == 0, msgs
    assert state_check.run(480, 640, 90, 16, streams=3, audited=4, log=msgs.append) == 0, msgs
    # a dense model (five live modes everywhere): from its second density probe on the library runs the instantiation
    # whose slot-1..4 loads use the streaming cache policy -- same numbers
    assert state_check.run(480, 640, 100, 10, streams=2, audited=4, dense=True, log=msgs.append) == 0, msgs
