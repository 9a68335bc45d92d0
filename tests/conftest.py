import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the tests force layouts (LWHIP_LANE_SPLIT, LWHIP_LWAVES, LWHIP_TILE_GENERIC ...): the library reads those experiment knobs only
# in a process that sets LWHIP_DEBUG
os.environ.setdefault('LWHIP_DEBUG', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (gfx950) device')


def pytest_addoption(parser):
    parser.addoption('--lw-shuffle', type=int, default=None, metavar='SEED',
                     help='run the collected tests in a seeded random order (tools/parity_stress.sh: order-dependent reads of '
                          'uninitialised device memory show up as failures that follow the seed)')


def pytest_collection_modifyitems(config, items):
    seed = config.getoption('--lw-shuffle')
    if seed is not None:
        import random
        random.Random(seed).shuffle(items)


@pytest.fixture(scope='session', autouse=True)
def _build_checkers():
    """TEST INFRASTRUCTURE: make sure the C oracle (and, where the reference sources exist, the
    reference build under oracle/_ref) is compiled before any test runs."""
    subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle'), 'all'], check=True,
                   stdout=subprocess.DEVNULL)


@pytest.fixture(scope='session')
def hip_lib():
    # torch bundles its own HIP runtime: when both live in one process (the RCCL test, bench.py)
    # torch has to initialise first, otherwise it reports "No HIP GPUs are available".
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass
    from lightweaver_amd import build, context
    build.build_hip()
    return context.load_library()


def _gpu_present():
    try:
        from lightweaver_amd import context
        return context.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope='session')
def gpu(hip_lib):
    """GPU tests must run the HIP path; with no device they fail (never silently pass)."""
    if hip_lib.lwhip_device_count() < 1:
        pytest.fail('no gfx950 device visible: -m gpu tests need a GPU box')
    return True


@pytest.fixture(autouse=True)
def _gpu_tests_initialise_torch_first(request):
    """Every test marked `gpu` goes through the `gpu` fixture, whether or not it names it: torch's HIP
    runtime has to be initialised before the library is first loaded in the process."""
    if request.node.get_closest_marker('gpu') is not None:
        request.getfixturevalue('gpu')
