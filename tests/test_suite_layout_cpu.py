"""Host-side checks of the GPU suite's own plumbing (no GPU needed): the bf16 child's selection and the fallback-route child's de-selection
name tests by substring -- a renamed test would silently drop out of (or into) a child run."""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))


def _gpu_test_names():
    names = []
    for f in sorted(os.listdir(HERE)):
        if f.startswith('test_') and f.endswith('.py'):
            names += re.findall(r'^def (test_[A-Za-z0-9_]+)\(', open(os.path.join(HERE, f)).read(), flags=re.M)
    return names


def test_every_name_of_the_bf16_child_selection_matches_a_test():
    from tests.conftest import BF16_CHILD_SELECT
    names = _gpu_test_names()
    for token in BF16_CHILD_SELECT.split(' or '):
        assert any(token in n for n in names), 'BF16_CHILD_SELECT names %r, which matches no test function' % token


def test_the_fallback_route_child_names_existing_tests_and_switches():
    src = open(os.path.join(HERE, 'test_hip_gemm.py')).read()
    names = _gpu_test_names()
    for token in re.findall(r'not ([a-z0-9_]+)', src[src.index('def test_kernel_suite_on_the_fallback_routes'):src.index('WS_SHAPES = [')]):
        if token in ('gpu',):
            continue
        assert any(token in n for n in names), token
    # every forced switch is one the library actually reads (csrc: evk_tunable("NAME", default))
    from tests.test_hip_gemm import FALLBACK_ROUTES
    csrc = os.path.join(os.path.dirname(HERE), 'evoke_amd', 'csrc')
    read = set()
    for f in os.listdir(csrc):
        if f.endswith(('.hip', '.h')):
            read |= set(re.findall(r'evk_tunable\("(EVK_[A-Z0-9_]+)"', open(os.path.join(csrc, f)).read()))
    for k in FALLBACK_ROUTES:
        assert k == 'EVK_EXPERIMENTAL' or k in read, '%s is forced by the fallback-route child but no kernel route reads it' % k
