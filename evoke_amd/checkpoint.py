"""Checkpoint interoperability with the reference.

* The reference saves `{'epoch', 'state_dict', 'optimizer', 'monitor_best'}` (modules/trainer_v0401.py:160-166), resumes by key
  (`_resume_checkpoint`, 181-189) and warm-starts stage 2 from stage 1 through a key-and-shape filter (`_load_checkpoint`, 191-202).
  `manifest()` / `optimizer_manifest()` describe a model / optimizer the way such a file does -- key, shape, dtype, in order -- and are
  compared against manifests written from the imported reference (tests/golden/manifest_*.json.gz).
* `cvt2distilgpt2` warm start (README.md:129): that checkpoint is a Lightning file whose `state_dict` holds the generator under
  `decoder.encoder_decoder.decoder.*` (HF GPT2LMHeadModel with cross-attention inside an EncoderDecoderModel) next to a CvT image
  encoder under `encoder.*`.  models/language_encoder/language_model.py:215-220 builds `self.decoder = Decoder()` with an
  `encoder_decoder` attribute precisely so that those keys line up; under FineTune the module sits at `text_decoder`, so the map is
  `decoder.encoder_decoder.decoder.X -> text_decoder.decoder.encoder_decoder.decoder.X`.  Everything else in the file (the CvT
  encoder, HF's causal-mask buffers `attn.bias` / `attn.masked_bias` / `crossattention.bias`, the tied `lm_head.weight`) has no
  counterpart and is reported, not loaded; tensors whose shape differs (the vocabulary-sized embeddings under another tokenizer,
  the reference passes ignore_mismatched_sizes=True at :182) are reported too, exactly like `_load_checkpoint` does.
"""
import torch

CVT2DISTILGPT2_PREFIX = 'decoder.encoder_decoder.decoder.'
ENGINE_GPT2_PREFIX = 'text_decoder.decoder.encoder_decoder.decoder.'
_HF_BUFFERS = ('.attn.bias', '.attn.masked_bias', '.crossattention.bias', '.crossattention.masked_bias')


def manifest(state_dict):
    """[(key, shape, dtype)] in state_dict order."""
    return [(k, list(v.shape), str(v.dtype).replace('torch.', '')) for k, v in state_dict.items()]


def optimizer_manifest(opt_state_dict):
    """param_groups -> number of parameters and the hyper-parameter names; state -> per parameter index the tensor entries' shapes."""
    groups = [{'n_params': len(g['params']), 'first': g['params'][0] if g['params'] else None,
               'hyper': sorted(k for k in g if k != 'params')} for g in opt_state_dict['param_groups']]
    state = {int(i): {k: (list(v.shape) if torch.is_tensor(v) else None) for k, v in ent.items()} for i, ent in opt_state_dict['state'].items()}
    return {'param_groups': groups, 'state': state}


def map_cvt2distilgpt2_keys(state_dict):
    """-> ({engine key: tensor}, [keys of the file without a counterpart]) for a cvt2distilgpt2 state_dict."""
    out, skipped = {}, []
    for k, v in state_dict.items():
        if not k.startswith(CVT2DISTILGPT2_PREFIX) or k.endswith(_HF_BUFFERS):
            skipped.append(k)
            continue
        name = k[len(CVT2DISTILGPT2_PREFIX):]
        if name == 'lm_head.weight':           # tied to transformer.wte.weight (GPT2LMHeadModel): one tensor in the engine
            skipped.append(k)
            continue
        out[ENGINE_GPT2_PREFIX + name] = v
    return out, skipped


def filter_by_shape(current, loaded):
    """trainer_v0401.py:196-199: keep what exists with an equal shape -> (valid dict, invalid key set)."""
    valid = {k: v for k, v in loaded.items() if k in current and tuple(v.shape) == tuple(current[k].shape)}
    return valid, {k for k in loaded if k not in valid}


def load_cvt2distilgpt2(model, checkpoint):
    """Warm-start `model.text_decoder` (the distilgpt2 backend, args['text_decoder'] = 'distilgpt2') from a cvt2distilgpt2 checkpoint:
    `checkpoint` = path, the loaded file, or its state_dict.  Returns {'loaded': [...], 'invalid': [...], 'skipped': [...]}."""
    if isinstance(checkpoint, str):
        checkpoint = torch.load(checkpoint, map_location='cpu')
    sd = checkpoint.get('state_dict', checkpoint)
    mapped, skipped = map_cvt2distilgpt2_keys(sd)
    current = model.state_dict()
    valid, invalid = filter_by_shape(current, mapped)
    if not valid:
        raise RuntimeError('no tensor of the checkpoint fits the model: is text_decoder the distilgpt2 backend?')
    current.update(valid)
    model.load_state_dict(current, strict=False)
    return {'loaded': sorted(valid), 'invalid': sorted(invalid), 'skipped': skipped}
