import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: long CPU test')


GOLDEN = os.path.join(REPO, 'tests', 'golden')


@pytest.fixture(scope='session')
def tokenizer():
    from evoke_amd.tokenizer import load_tokenizer
    return load_tokenizer(os.path.join(GOLDEN, 'iu_xray_wordlevel_uncased_tokenizer.json'))
