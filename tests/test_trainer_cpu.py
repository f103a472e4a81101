"""Step harness (evoke_amd/trainer.py) host logic on the CPU: checkpoint format / resume / shape-filtered load and the
epoch bookkeeping of modules/trainer_v0401.py, exercised with a toy model and a stock torch optimizer."""
import os

import torch
import torch.nn as nn

from evoke_amd.trainer import Trainer, filter_state_for


class Toy(nn.Module):
    def __init__(self, d=4):
        super().__init__()
        self.a = nn.Linear(d, d)
        self.b = nn.Linear(d, 1)
        self.calls = []

    def forward(self, images, ids, masks, pids, inc_ids=None, inc_masks=None, mode='train'):
        self.calls.append('inc' if inc_ids is not None else 'no_inc')
        loss = self.b(torch.tanh(self.a(images))).pow(2).mean()
        return {'lm': loss, 'all_loss': loss}


def _loader(n, inc, d=4):
    out = []
    for i in range(n):
        g = torch.Generator().manual_seed(i + (100 if inc else 0))
        b = [['id%d' % i], torch.randn(2, d, generator=g), torch.zeros(2, 3, dtype=torch.long), torch.ones(2, 3, dtype=torch.long), ['p0', 'p1']]
        if inc:
            b += [torch.zeros(2, 3, dtype=torch.long), torch.ones(2, 3, dtype=torch.long)]
        out.append(tuple(b))
    return out


def test_filter_state_for():
    cur = {'x': torch.zeros(2, 3), 'y': torch.zeros(4)}
    valid, invalid = filter_state_for(cur, {'x': torch.ones(2, 3), 'y': torch.ones(5), 'z': torch.ones(1)})
    assert set(valid) == {'x'} and invalid == {'y', 'z'}


def test_finetune_epoch_order_and_mean(tmp_path):
    torch.manual_seed(0)
    m = Toy()
    opt = torch.optim.RAdam(m.parameters(), lr=1e-2)
    tr = Trainer(m, opt, {'result_dir': str(tmp_path)}, log=lambda s: None)
    # reference value of the running mean: same model/optimizer stepped by hand in FTrainer's order
    m2 = Toy()
    m2.load_state_dict(m.state_dict())
    opt2 = torch.optim.RAdam(m2.parameters(), lr=1e-2)
    tot = 0.0
    for b in _loader(2, True) + _loader(3, False):
        opt2.zero_grad()
        loss = m2(*b[1:])['all_loss']
        loss.backward()
        opt2.step()
        tot += loss.item()
    log = tr.train_epoch_finetune(_loader(2, True), _loader(3, False), epoch=3)
    assert m.calls == ['inc', 'inc', 'no_inc', 'no_inc', 'no_inc']
    assert log['epoch'] == 3 and abs(log['train_loss'] - tot / 5) < 1e-6


def test_checkpoint_format_resume_and_partial_load(tmp_path):
    torch.manual_seed(1)
    m = Toy()
    opt = torch.optim.RAdam(m.parameters(), lr=1e-2)
    args = {'result_dir': str(tmp_path), 'monitor_mode': 'max', 'monitor_metric': 'BLEU_4', 'save_period': 1}
    tr = Trainer(m, opt, args, log=lambda s: None)
    tr.train_epoch_finetune(None, _loader(2, False))
    best, stop = tr.end_of_epoch(1, {'val_BLEU_4': 0.25})
    assert best and not stop
    cur = os.path.join(str(tmp_path), 'checkpoint', 'current_checkpoint.pth')
    assert os.path.exists(cur) and os.path.exists(os.path.join(str(tmp_path), 'checkpoint', 'model_best.pth'))
    ck = torch.load(cur)
    assert set(ck) == {'epoch', 'state_dict', 'optimizer', 'monitor_best'} and ck['epoch'] == 1 and ck['monitor_best'] == 0.25
    best, _ = tr.end_of_epoch(2, {'val_BLEU_4': 0.10})
    assert not best and tr.not_improved == 1

    m2 = Toy()
    opt2 = torch.optim.RAdam(m2.parameters(), lr=1e-2)
    tr2 = Trainer(m2, opt2, dict(args, resume=cur), log=lambda s: None)
    assert tr2.start_epoch == 3 and tr2.mnt_best == 0.25
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])
    assert opt2.state_dict()['state'][0]['exp_avg'].abs().sum() > 0

    m3 = Toy(d=4)
    m3.b = nn.Linear(4, 2)                       # shape mismatch on b.*: must be skipped, a.* loaded
    before = m3.b.weight.clone()
    tr3 = Trainer(m3, torch.optim.RAdam(m3.parameters()), {'result_dir': str(tmp_path)}, log=lambda s: None)
    invalid = tr3.load_checkpoint(cur)
    assert invalid == {'b.weight', 'b.bias'}
    assert torch.equal(m3.a.weight, m.a.weight) and torch.equal(m3.b.weight, before)
