import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session', autouse=True)
def _debug_options_from_env():
    """FDSR_TEST_DEBUG_OPTION="name=value[,name=value]": a TEST-harness variable (the library itself never reads the
    environment) that child pytest sessions use to run a subset of tests under one launcher option (fdsr_debug_option)."""
    spec = os.environ.get('FDSR_TEST_DEBUG_OPTION')
    if spec:
        from fastdiffsr_amd import _lib
        for item in spec.split(','):
            k, v = item.split('=')
            _lib.debug_option(k.strip(), int(v))
    yield


# ---------------------------------------------------------------------------------------------------------------------
# One CPU oracle image of a 256x256 20-step loop costs 10-25 s; several GPU test modules compare an engine image with it.
# They share this session-wide cache (keyed by the weights and the exact cond / noise bytes), and wherever a test is free to
# choose its inputs it plants `standard_pair()` at the batch index it checks, so that ONE oracle image serves them all.
# ---------------------------------------------------------------------------------------------------------------------
_ORACLE_IMAGES = {}


def standard_pair():
    """The B=1 256x256 inputs of the parity tests: (cond [1,3,256,256], noise [20,1,3,256,256])."""
    from fastdiffsr_amd.synth import synth_inputs
    return synth_inputs(1, 256, 256, 20)


def plant_standard_pair(cond, noise, index):
    """Overwrite image `index` of a batch's inputs with the standard pair (in place); returns (cond, noise)."""
    c1, n1 = standard_pair()
    cond[index] = c1[0]
    noise[:, index] = n1[:, 0]
    return cond, noise


def oracle_loop_image(sd, cfg, cond, noise):
    """oracle p_sample_loop(cond, noise) for ONE image under the val schedule, cached for the session."""
    import hashlib
    import numpy as np
    from fastdiffsr_amd.arch import FASTDIFFSR_SCHEDULE_VAL
    from fastdiffsr_amd.synth import state_dict_sha256
    from oracle import fdsr_oracle as O
    h = hashlib.sha256()
    h.update(state_dict_sha256(sd).encode())
    h.update(repr(cfg).encode())
    h.update(np.ascontiguousarray(cond.numpy()).tobytes())
    h.update(np.ascontiguousarray(noise.numpy()).tobytes())
    key = h.hexdigest()
    if key not in _ORACLE_IMAGES:
        _ORACLE_IMAGES[key] = O.p_sample_loop(O.to_torch_sd(sd), cfg, O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL), cond, noise)
    return _ORACLE_IMAGES[key].clone()          # tensor2img clamps its argument in place, like the reference's
