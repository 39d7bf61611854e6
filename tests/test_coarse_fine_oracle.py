"""CPU checks of the oracle's intended coarse -> fine resampling (oracle.sample_pdf_intended: the reference's sample_pdf is dead
code, src/nerf.py:1745-1779, so there is no golden -- parity unpinned; these tests pin the restatement against an independent
scalar loop and against the properties an inverse cdf must have)."""
import bisect

import numpy as np
import torch

import oracle as O


def _scalar_loop(ts, w, N, u=None):
    """the same reading, one ray at a time in Python floats (fp64)"""
    T, R = w.shape
    out = np.zeros((N, R))
    uu = O.linspace01_f32(N).double().numpy() if u is None else None
    for r in range(R):
        wp = [float(w[i, r]) + 1e-5 for i in range(T - 1)]
        s = sum(wp)
        cdf = [0.0]
        acc = 0.0
        for x in wp:
            acc += x / s
            cdf.append(acc)
        for j in range(N):
            x = float(uu[j]) if u is None else float(u[j, r])
            ind = bisect.bisect_right(cdf, x)
            below, above = max(ind - 1, 0), min(ind, T - 1)
            den = cdf[above] - cdf[below]
            if den < 1e-5:
                den = 1.0
            t = (x - cdf[below]) / den
            out[j, r] = float(ts[below]) + t * (float(ts[above]) - float(ts[below]))
    return out


def test_sample_pdf_intended_matches_a_scalar_loop():
    g = torch.Generator().manual_seed(0)
    ts = torch.linspace(2, 6, 17)
    w = torch.rand(17, 9, generator=g) ** 3
    w[:, 0] = 0
    for u in (None, torch.rand(11, 9, generator=g)):
        got = O.sample_pdf_intended(ts, w, 11, u).numpy()
        ref = _scalar_loop(ts.numpy(), w.numpy(), 11, None if u is None else u.numpy())
        assert np.abs(got - ref).max() <= 1e-12


def test_linspace01_is_torchs_linspace_to_the_last_bit_or_one():
    assert O.linspace01_f32(1).tolist() == torch.linspace(0, 1, 1).tolist() == [0.0]  # (a single draw sits at u = 0)
    for N in (2, 5, 64, 128, 200):
        a, b = O.linspace01_f32(N), torch.linspace(0, 1, N, dtype=torch.float)
        assert float((a - b).abs().max()) <= 6e-8 and float(a[0]) == 0.0 and float(a[-1]) == 1.0


def test_sample_pdf_intended_properties():
    ts = torch.linspace(2, 6, 33)
    flat = torch.full((33, 1), 1 / 33.0)
    s = O.sample_pdf_intended(ts, flat, 65)[:, 0]
    # equal mass per interval: the inverse cdf of the deterministic draw is the uniform grid itself
    assert float((s - torch.linspace(2, 6, 65, dtype=torch.float64)).abs().max()) <= 1e-6
    peak = torch.zeros(33, 1)
    peak[10, 0] = 1.0
    s = O.sample_pdf_intended(ts, peak, 64)[:, 0]
    inside = ((s >= ts[10]) & (s <= ts[11])).float().mean()
    assert inside >= 0.95                                   # the interval that holds the mass gets the samples
    assert float(s.min()) >= 2.0 - 1e-9 and float(s.max()) <= 6.0 + 1e-9
    assert bool((s[1:] >= s[:-1]).all())                    # monotone in u
    m = O.merge_ts_intended(ts.double(), s[:, None])[:, 0]
    assert m.shape[0] == 33 + 64 and bool((m[1:] >= m[:-1]).all())
    # the last row of `weights` (the 1e10 interval) never enters: changing it changes nothing
    w2 = peak.clone(); w2[-1, 0] = 7.0
    assert torch.equal(O.sample_pdf_intended(ts, w2, 64), O.sample_pdf_intended(ts, peak, 64))
