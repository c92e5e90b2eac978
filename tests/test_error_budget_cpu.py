"""The CPU restatement of the kernel's summation order (tools/emu, tools/error_budget.py): it reproduces its own committed
fixture (which tests/test_gpu_parity.py pins to the kernel's bits on the GPU) and shows what the order is for -- the round-4
order is closer to the float64 evaluation of the reference than the single chain of rounds 1-3."""
import os
import sys

import numpy as np

from conftest import GOLDEN, REPO

sys.path.insert(0, os.path.join(REPO, "tools"))


def test_emulation_reproduces_its_fixture_and_the_two_chain_order_is_closer_to_float64():
    import error_budget as E
    chk = np.load(os.path.join(GOLDEN, "emu_check.npz"))
    sw = np.load(os.path.join(GOLDEN, "sweep_worst.npz"))
    sets = E.weight_sets()
    P, V = E.Net(sets["A"][0]), E.Net(sets["A"][1])
    x = chk["features"][:24]
    assert np.array_equal(x, sw["features_A"][:24])
    r4 = P.run(x, *E.SHIPPED)
    r3 = P.run(x, 0, 0, 0, 0)
    assert np.array_equal(r4, chk["logits_r4"][:24]) and np.array_equal(r3, chk["logits_r3"][:24])
    assert np.array_equal(V.run(x, *E.SHIPPED)[1][:, 0], chk["value_pre_tanh_r4"][:24])
    lg64, lgref = sw["logits_f64_A"][:24], sw["logits_A"][:24]
    d4, d3 = np.abs(r4 - lg64).max(), np.abs(r3 - lg64).max()
    assert d4 < 4.5e-5 and d4 < 0.75 * d3, (d4, d3)
    assert np.abs(r4 - lgref).max() < 6.5e-5
