"""Tokenizer boundary -- mirrors modules/tokenizers_new.py:45-66 (build_my_tokenizer).

Tokenisation itself stays HF `tokenizers` (WordLevel + Whitespace pre-tokenizer);
the engine only needs the ids the reference's FineTune/Pretrain read from it: vocab size, [BOS]/[EOS]/[PAD].
"""
import os

from tokenizers import Tokenizer


def load_tokenizer(tokenizer_path):
    """Tokenizer.from_file + the two special tokens the reference appends (tokenizers_new.py:64-65)."""
    tok = Tokenizer.from_file(tokenizer_path)
    tok.add_special_tokens(['[BOS]', '[EOS]'])
    return tok


def build_my_tokenizer(tokenizer_dir='config/tokenizer', model='wordlevel', data_name='mimic_cxr', ann_path=None,
                       tokenizer_type='uncased', is_same_tokenizer=False):
    """Same signature / file-name rule as the reference; training a missing tokenizer from an annotation
    file is caller-side data preparation (out of scope) and raises instead."""
    if is_same_tokenizer:
        data_name = 'mimic_cxr'
    path = os.path.join(tokenizer_dir, '%s_%s_%s_tokenizer.json' % (data_name.lower(), model.lower(), tokenizer_type.lower()))
    if not os.path.exists(path):
        raise FileNotFoundError('%s not found; train it with the reference tooling (tokenizers_new.py:26-42)' % path)
    return load_tokenizer(path)
