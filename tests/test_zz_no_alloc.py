"""vsm_match_pair_noalloc (include/vsmatch.h): the entry point that allocates nothing -- its scratch
is a static device array -- against vsm_match_pair, through the C ABI, bit-exact (integer/index
work).  Two calls in a row on the same GPU, the second with a smaller input (the array keeps the
first call's contents), further shapes, the size limit and the zero-size case."""
import importlib

import numpy as np
import pytest

from oracle import binding
from tests import cases

pytestmark = pytest.mark.gpu
vsm = importlib.import_module("visual-slam_amd")


def _noalloc(m, torch, d1, d2, thr, ratio, cap=None):
    dev = torch.device("cuda", 0)
    t1 = torch.from_numpy(d1.view(np.int64).copy()).to(dev)
    t2 = torch.from_numpy(d2.view(np.int64).copy()).to(dev)
    cap = max(1, min(len(d1), len(d2))) if cap is None else cap
    out = torch.full((cap, 2), -7, dtype=torch.int32, device=dev)
    cnt = torch.full((1,), -7, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()  # inputs complete before the library's own stream reads them
    m.match_pair_noalloc(t1.data_ptr(), len(d1), t2.data_ptr(), len(d2), thr, ratio, out.data_ptr(), cap, cnt.data_ptr())
    torch.cuda.synchronize()  # (device-wide: covers the context's non-blocking stream)
    n = int(cnt.cpu()[0])
    return n, out.cpu().numpy()


def test_no_alloc_entry_point_equals_match_pair():
    import torch
    with vsm.Matcher(0) as m:
        # two calls in a row, the second smaller: 1500 x 1400, then 300 x 777
        big = cases.synth.orb_like_pair(1500, 1400, seed=41)
        small = cases.build_case("planted_300x777")[:2]
        for d1, d2 in (big, small):
            want = m.match(d1, d2, 70, 1.2)
            n, out = _noalloc(m, torch, d1, d2, 70, 1.2)
            assert n == len(want) and np.array_equal(out[:n], want)
            assert np.all(out[n:] == -7)  # nothing written past the list
            assert np.array_equal(want, binding.port().match(d1, d2, 70, 1.2))
        assert len(m.match(*big, 70, 1.2)) > 500
        # the reference's golden pair, real descriptors (second bests next to the best), other parameters
        d1, d2 = cases.ex3()
        for thr, ratio in ((70, 1.2), (90, 1.0), (256, 1.5)):
            want = m.match(d1, d2, thr, ratio)
            n, out = _noalloc(m, torch, d1, d2, thr, ratio)
            assert n == len(want) and np.array_equal(out[:n], want), (thr, ratio)
        # every scan shape the knob reaches, ragged sizes, the largest size
        for variant, (n1, n2) in ((0, (8192, 8000)), (0x2000000, (1000, 1084)), (0x0301, (64, 33)), (1, (5, 2000)), (0, (1, 1))):
            a, b = cases.synth.orb_like_pair(n1, n2, seed=n1 + n2)
            m.set_variant(variant)
            try:
                want = m.match(a, b, 70, 1.2)
                n, out = _noalloc(m, torch, a, b, 70, 1.2)
            finally:
                m.set_variant(0)
            assert n == len(want) and np.array_equal(out[:n], want), (hex(variant), n1, n2)
        # capacity smaller than the list: the count is the need, the first `cap` pairs are stored
        want = m.match(*big, 70, 1.2)
        n, out = _noalloc(m, torch, *big, 70, 1.2, cap=10)
        assert n == len(want) and np.array_equal(out, want[:10])
        # an empty side: count 0, nothing else touched
        n, out = _noalloc(m, torch, big[0], big[1][:0], 70, 1.2, cap=4)
        assert n == 0 and np.all(out == -7)
        # sizes beyond the fixed scratch are refused with a clear error, nothing is enqueued
        with pytest.raises(vsm.VsmError) as e:
            m.match_pair_noalloc(1, 8193, 1, 10, 70, 1.2, 1, 1, 1)
        assert e.value.code == vsm.matcher.VSM_ERR_INVALID and "8192" in str(e.value)
        m.set_variant(0xff01)  # 255 forced train splits of 8192 x 8192 do not fit the scratch
        try:
            with pytest.raises(vsm.VsmError) as e:
                m.match_pair_noalloc(1, 8192, 1, 8192, 70, 1.2, 1, 1, 1)
            assert "fixed scratch" in str(e.value)
        finally:
            m.set_variant(0)


def test_no_alloc_two_contexts_one_gpu_in_turn():
    """Every context on a GPU uses that GPU's one array (same address from hipGetSymbolAddress): calls
    in turn, each finished before the next, give the right lists."""
    import torch
    d1, d2 = cases.synth.orb_like_pair(700, 900, seed=5)
    with vsm.Matcher(0) as a, vsm.Matcher(0) as b:
        want = a.match(d1, d2, 70, 1.2)
        for m in (a, b, a):
            n, out = _noalloc(m, torch, d1, d2, 70, 1.2)
            assert n == len(want) and np.array_equal(out[:n], want)
