"""CPU: host-side optimiser plumbing against torch's own implementations — the plateau scheduler
the reference configures (cli/train.py:142-146), its warm-up rule (cli/train.py:189-191), and the
torch.optim.Adam state-dict layout its checkpoints carry (cli/train.py:321-336)."""
import torch

from edgedict_amd.optim import FusedAdam, ReduceLROnPlateau, WarmupLR


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))


def test_plateau_scheduler_matches_torch_on_a_noisy_curve():
    ref_opt = torch.optim.Adam(_net().parameters(), lr=5e-4)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(ref_opt, patience=1, factor=0.5, min_lr=1e-6)
    mine_opt = FusedAdam(_net(), lr=5e-4)
    mine = ReduceLROnPlateau(mine_opt, patience=1, factor=0.5, min_lr=1e-6)
    g = torch.Generator().manual_seed(3)
    loss = 10.0
    for i in range(60):
        loss = max(0.5, loss * (0.97 if i % 7 else 1.05)) + 0.05 * float(torch.randn(1, generator=g))
        ref.step(loss)
        mine.step(loss)
        assert abs(ref_opt.param_groups[0]["lr"] - mine_opt.param_groups[0]["lr"]) < 1e-15, i
    assert mine_opt.param_groups[0]["lr"] < 5e-4          # it did reduce at least once
    sd = mine.state_dict()
    again = ReduceLROnPlateau(mine_opt, patience=1, factor=0.5, min_lr=1e-6)
    again.load_state_dict(sd)
    assert again.best == mine.best and again.num_bad_epochs == mine.num_bad_epochs


def test_warmup_rule():
    opt = FusedAdam(_net(), lr=5e-4)
    w = WarmupLR(opt, 5e-4, 10000)
    assert abs(w.step(1) - 5e-8) < 1e-20
    assert abs(w.step(5000) - 2.5e-4) < 1e-18
    assert w.step(10000) == 5e-4
    opt.param_groups[0]["lr"] = 1e-4                      # e.g. reduced by the plateau scheduler
    assert w.step(10001) == 1e-4                          # warm-up no longer touches it


def test_adam_state_dict_round_trips_through_torch_adam():
    net_a, net_b = _net(), _net()
    opt = FusedAdam(net_a, lr=3e-4)
    opt.step_count = 17
    opt.m.uniform_(-1, 1)
    opt.v.uniform_(0, 1)
    sd = opt.state_dict()
    ref = torch.optim.Adam(net_b.parameters(), lr=1.0)
    ref.load_state_dict(sd)                               # torch accepts our checkpoint
    assert ref.param_groups[0]["lr"] == 3e-4
    for i, p in enumerate(net_b.parameters()):
        st = ref.state[p]
        assert int(st["step"]) == 17
        off, n = opt.flat.offsets[i], p.numel()
        assert torch.equal(st["exp_avg"].flatten(), opt.m[off:off + n])
    back = FusedAdam(_net(), lr=1.0)
    back.load_state_dict(ref.state_dict())                # and we accept torch's
    assert back.step_count == 17 and back.param_groups[0]["lr"] == 3e-4
    for off, p in zip(opt.flat.offsets, opt.flat.params):     # (alignment padding is not state)
        n = p.numel()
        assert torch.equal(back.m[off:off + n], opt.m[off:off + n])
        assert torch.equal(back.v[off:off + n], opt.v[off:off + n])
