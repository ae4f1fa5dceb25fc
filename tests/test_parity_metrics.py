"""The local parity metrics of tests/parity_utils.py catch what the global max-abs / max-abs figure lets through
(VERDICT r2 weak 3): a stem 40 dB below the loudest one that is entirely wrong, and a quiet passage that is wrong."""
import numpy as np
import pytest

import parity_utils as pu


def _stems(seed=0, n=6 * 4096):
    rng = np.random.default_rng(seed)
    ref = rng.standard_normal((4, 2, n))
    ref[2] *= 0.01  # one stem 40 dB down
    ref[0, :, 3 * 4096:4 * 4096] *= 0.003  # a quiet passage in a loud stem
    return ref


def test_global_metric_misses_a_wrong_quiet_stem_and_the_local_ones_catch_it():
    ref = _stems()
    got = ref + 1e-7 * np.random.default_rng(1).standard_normal(ref.shape)
    pu.assert_local_parity(got, ref)  # fp32-class error passes
    bad = got.copy()
    bad[2] = -ref[2]  # the quiet stem entirely wrong
    assert pu.relerr(bad, ref) < 3e-2  # a 200 % error of that stem is a 2 % blip for max|a-b| / max|b| ...
    worst, sdr = pu.local_errors(bad, ref)
    assert worst > 1.0 and sdr < 0.0  # ... blockwise error 2x the stem's scale, SDR of that stem -6 dB
    with pytest.raises(AssertionError):
        pu.assert_local_parity(bad, ref)


def test_quiet_passage_is_judged_on_its_own_scale_down_to_the_floor():
    ref = _stems()
    bad = ref.copy()
    bad[0, :, 3 * 4096:4 * 4096] *= 1.5  # 50 % error inside the quiet passage only (0.15 % of the stem's max-abs)
    assert pu.relerr(bad, ref) < 1e-2
    worst, _ = pu.local_errors(bad, ref)
    assert worst > 0.1
    with pytest.raises(AssertionError):
        pu.assert_local_parity(bad, ref)


def test_channel_metric_for_taps():
    rng = np.random.default_rng(2)
    r = rng.standard_normal((8, 50, 7))
    r[5] *= 0.02
    g = r.copy()
    assert pu.channel_relerr(g, r) == 0.0
    g[5] *= 1.2
    assert pu.relerr(g, r) < 5e-3 and pu.channel_relerr(g, r) > 0.1
