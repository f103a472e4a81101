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


# ---- the bf16-storage build's pass over the parity tests (tests/test_model_gpu.py::test_bf16_storage_build_passes_the_gpu_suite) -------------
# The storage format is fixed per process, so that pass is a child interpreter.  It used to run INSIDE its test (275 s of a 660 s suite, the
# parent idle meanwhile); now the child is started when collection has finished and works beside the parent's own tests -- two processes on the
# card, far below the box's limit -- and the test only waits for its verdict.  What the child runs: every model-level parity test against its
# reference and ONE representative per kernel family (the other parametrisations exercise host logic and tile selection that do not depend on
# the storage format); what it leaves to the default build's pass: multi-process reducer tests, serving-loop driving modes, trainer / optimizer
# order tests, the forced-tile re-run.  (ONE child beside the parent: with the fallback-route child of tests/test_hip_gemm.py started at the same time
# the three processes oversubscribe the box's 16-CPU share -- the oracle legs of the two-rank tests then take 140 s instead of 20, the suite 717 s.)
BF16_CHILD_SELECT = ' or '.join([
    # model level
    'finetune_matches_reference', 'pretrain_matches_reference', 'training_forward_log_probabilities', 'trunk_follows_bf16_emulation',
    'inference_trunk_with_batchnorm', 'beam_search_matches_reference', 'beam_search_follows_the_reference_decisions',
    'beam_search_16bit_recurrence', 'distilgpt2_backend_matches', 'finetune_with_distilgpt2', 'loss_parity_at_realistic', 'full_size_step_properties',
    'full_size_beam_decode', 'edge_geometries_match_oracle', 'pipelined_generation_equals_per_batch',
    # one representative per kernel family
    'test_gemm_nt', 'test_gemm_tn_dw_accumulate', 'test_weight_gradient_kernel_tn', 'test_conv_fwd_dgrad_wgrad', 'test_stem', 'test_weight_stationary_pointwise',
    'test_strip_gemm_forward', 'test_halo_conv3x3_forward', 'test_halo_conv3x3_weight_gradient', 'test_stride2_conv3x3', 'test_layernorm', 'test_conditional_layernorm',
    'test_linear_fwd_bwd', 'test_attention', 'test_batchnorm_train', 'test_maxpool_and_patch_mean', 'test_embedding_and_nll', 'test_decode_rowblock',
    'test_beam_step_kernel', 'test_contrastive_losses', 'test_relational_memory_step_matches_oracle', 'test_rm_decode_step_f32', 'test_optim_step_matches_torch',
    'test_optim_group_step', 'test_native_trunk_matches_module_walk', 'test_decode_attention_matches_reference', 'test_linear_with_layernorm_in_the_operand',
    'test_preprocess_matches_pillow', 'test_dgrad_gate_statistics', 'test_step_graph_replays_the_eager_trajectory', 'test_dynamic_loss_scale_skips'])
BF16_CHILD = {}


def start_bf16_child():
    import subprocess
    import tempfile
    log = tempfile.NamedTemporaryFile('w+', prefix='evk_bf16_child_', suffix='.log', delete=False)
    cmd = [sys.executable, '-m', 'pytest', os.path.join(REPO, 'tests'), '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider', '-k', BF16_CHILD_SELECT,
           '--durations', '8']
    BF16_CHILD.update(proc=subprocess.Popen(cmd, env=dict(os.environ, EVK_STORE='bf16'), stdout=log, stderr=subprocess.STDOUT, cwd=REPO), log=log)


def pytest_collection_finish(session):
    names = {it.name for it in session.items}
    if session.config.option.collectonly:
        return
    import torch
    if not torch.cuda.is_available():
        return
    if ('test_bf16_storage_build_passes_the_gpu_suite' in names and len(names) > 40          # the whole GPU suite, not a hand-picked test
            and os.environ.get('EVK_STORE', 'f16').lower() != 'bf16' and os.environ.get('EVK_BF16_CHILD_INLINE') != '1'):
        start_bf16_child()


def pytest_sessionfinish(session, exitstatus):
    """A session that ends before the bf16 test has joined its child (-x after an earlier failure, a keyboard interrupt) must not leave the
    child running on the GPU."""
    proc = BF16_CHILD.get('proc')
    if proc is not None and proc.poll() is None:
        proc.kill()
        try:
            proc.wait(timeout=30)
        except Exception:          # noqa: BLE001
            pass
    log = BF16_CHILD.get('log')
    if log is not None:
        try:
            log.close()
            os.unlink(log.name)
        except OSError:
            pass
    BF16_CHILD.clear()
