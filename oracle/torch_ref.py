"""oracle/torch_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference's CPU path restated with PyTorch ops (BASELINE.md 4 / SURVEY 8d "CPU baseline beside it"): the same
operators the reference calls -- Conv2d, eval-mode BatchNorm2d/1d, ReLU, F.conv2d + untied bias, Linear, tanh,
softmax (bokego/nnet.py:31-57, 73-113, 175-180, 16) -- assembled from a state_dict with the reference's names, so
that timing it on the GPU box's host cores times the reference's arithmetic on the reference's backend (oneDNN /
MKL) without the reference travelling.  Pinned against tests/golden/nets.npz by tests/test_oracle.py.
Only tests/ and bench.py's cpu_baseline leg import this; the product never does.
"""
import numpy as np
import torch
import torch.nn.functional as F

_CONV = (0, 3, 6, 9, 12, 15, 18)


def _t(sd, name):
    v = sd[name]
    return (v.detach().clone() if isinstance(v, torch.Tensor) else torch.from_numpy(np.array(v, dtype=np.float32))).float()


class _Trunk(torch.nn.Module):
    """conv.{0,3,..,18} + BatchNorm conv.{1,4,..,19} + ReLU, then the untied-bias 1x1 conv.21 -> [B,1,9,9]."""

    def __init__(self, sd):
        super().__init__()
        self.blocks = torch.nn.ModuleList()
        for l, c in enumerate(_CONV):
            k = 5 if l == 0 else 3
            conv = torch.nn.Conv2d(27 if l == 0 else 128, 128, k, padding=k // 2)
            bn = torch.nn.BatchNorm2d(128)
            with torch.no_grad():
                conv.weight.copy_(_t(sd, f"conv.{c}.weight")); conv.bias.copy_(_t(sd, f"conv.{c}.bias"))
                bn.weight.copy_(_t(sd, f"conv.{c + 1}.weight")); bn.bias.copy_(_t(sd, f"conv.{c + 1}.bias"))
                bn.running_mean.copy_(_t(sd, f"conv.{c + 1}.running_mean"))
                bn.running_var.copy_(_t(sd, f"conv.{c + 1}.running_var"))
            self.blocks.append(torch.nn.Sequential(conv, bn, torch.nn.ReLU()))
        self.register_buffer("head_w", _t(sd, "conv.21.weight").reshape(1, 128, 1, 1))
        self.register_buffer("head_b", _t(sd, "conv.21.bias").reshape(1, 1, 9, 9))

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return F.conv2d(x, self.head_w) + self.head_b


class TorchPolicy(torch.nn.Module):
    def __init__(self, sd):
        super().__init__()
        self.trunk = _Trunk(sd)
        self.eval()

    @torch.no_grad()
    def forward(self, x):
        return self.trunk(x).reshape(-1, 81)


class TorchValue(torch.nn.Module):
    def __init__(self, sd):
        super().__init__()
        self.trunk = _Trunk(sd)
        self.bn = torch.nn.BatchNorm2d(1)
        self.lin1 = torch.nn.Linear(81, 64)
        self.lin_bn = torch.nn.BatchNorm1d(64)
        self.lin2 = torch.nn.Linear(64, 1)
        with torch.no_grad():
            for mod, pre in ((self.bn, "bn"), (self.lin_bn, "lin_bn")):
                mod.weight.copy_(_t(sd, pre + ".weight").reshape(-1)); mod.bias.copy_(_t(sd, pre + ".bias").reshape(-1))
                mod.running_mean.copy_(_t(sd, pre + ".running_mean").reshape(-1))
                mod.running_var.copy_(_t(sd, pre + ".running_var").reshape(-1))
            for mod, pre in ((self.lin1, "lin1"), (self.lin2, "lin2")):
                mod.weight.copy_(_t(sd, pre + ".weight").reshape(mod.weight.shape)); mod.bias.copy_(_t(sd, pre + ".bias").reshape(-1))
        self.eval()

    @torch.no_grad()
    def forward(self, x):
        h = torch.relu(self.bn(self.trunk(x))).reshape(-1, 81)
        return torch.tanh(self.lin2(torch.relu(self.lin_bn(self.lin1(h))))).reshape(-1)


def leaf_eval(policy, value, x):
    """one leaf-eval batch as the reference's policy_dist + value do it: logits -> softmax, and the value."""
    lg = policy(x)
    return lg, torch.softmax(lg, dim=1), value(x)
