"""Default argument dictionary of the engine's FineTune / Pretrain: the keys main_224.py / main_384.py pass after merging
config/finetune_config.yaml with modules/utils.py's parser defaults (the values below are those defaults and YAML entries for the
released EVOKE configuration: 3-layer R2Gen decoder, d_model 512, relational memory 3 x 512, BERT-6 text encoder, 2048-wide heads).
bench.py, __graft_entry__.smoke() and the tests start from it; a caller's own dict overrides any key."""
import os


def tunable(name, default):
    """Environment switches that select a measured ALTERNATIVE to a default of the engine (older decode paths, stream priorities, capture modes:
    INTEGRATION.md section 4 lists them) count only under EVK_EXPERIMENTAL=1 -- the same rule as csrc's evk_tunable -- so that a stray variable
    cannot move a production run onto a route the default test pass does not cover.  Product-level switches (EVK_STORE, EVK_GRAD_SYNC,
    EVK_DECODE_DEPTH, ...) are read directly."""
    return os.environ.get(name, default) if os.environ.get('EVK_EXPERIMENTAL', '0') == '1' else default


ARGS = dict(
    resnet_checkpoint='', text_checkpoint=None, fusion_checkpoint=None, vocab_size=1444, encoder_hidden_size=768,
    encoder_num_hidden_layers=6, output_dim=2048, fusion_num_heads=8, sk_fusion_num_layers=1, max_seq_len=100,
    is_multiview_learning=True, is_add_indication=True, instance_temp=0.5, region_temp=0.5, num_layers=3, d_model=512,
    d_ff=512, d_vf=2048, num_heads=8, dropout=0.0, drop_prob_lm=0.5, use_bn=0, rm_num_slots=3, rm_num_heads=8,
    rm_d_model=512, sample_method='beam_search', beam_size=3, temperature=1.0, sample_n=1, group_size=1,
    output_logsoftmax=1, decoding_constraint=0, block_trigrams=1, length_penalty='', diversity_lambda=0.5, suppress_UNK=0)

# the IU X-ray word-level tokenizer the reference ships (config/tokenizer/iu_xray_wordlevel_uncased_tokenizer.json), kept as a data
# fixture under tests/golden; the MIMIC-CXR vocabulary is trained on the fly by the reference and is not available offline
DEFAULT_TOKENIZER_JSON = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden',
                                      'iu_xray_wordlevel_uncased_tokenizer.json')


def load_default_tokenizer():
    from .tokenizer import load_tokenizer
    return load_tokenizer(DEFAULT_TOKENIZER_JSON)


def load_state_by_key(model, state, device='cuda'):
    """Load a {state_dict key: tensor} mapping by key, refusing unexpected or missing keys (HF's non-persistent position_ids
    excepted), and move the model to `device`."""
    res = model.load_state_dict(state, strict=False)
    if res.unexpected_keys or not all(k.endswith('position_ids') for k in res.missing_keys):
        raise RuntimeError('state does not fit the model: unexpected %s missing %s' % (res.unexpected_keys[:5], res.missing_keys[:5]))
    return model.to(device)
