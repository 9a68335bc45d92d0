import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch; torch.cuda.init()
import pytest
# run the preceding GPU tests of the file in-process, then the shard test with prints
import tests.test_prd as T
class G: pass
for name in ['test_hip_prd_golden', 'test_hip_prd_single_call', 'test_hip_prd_device_resident_matches_drop_in', 'test_hip_prd_single_call_refused_on_a_shard']:
    fn = getattr(T, name)
    try:
        fn(None) if fn.__code__.co_argcount else fn()
        print(name, 'ok')
    except Exception as e:
        print(name, 'EXC', repr(e)[:200])
    try:
        T.test_hip_prd_on_wavelength_shards(None)
        print('  shards ok after', name)
    except AssertionError as e:
        print('  shards FAIL after', name, str(e)[:300])
