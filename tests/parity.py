"""Shared comparison helpers.  Tolerance is the north star's (BASELINE.json): every channel
within +-1 LSB and >= 99.9 % of pixels identical; the raw float image is additionally compared
bit for bit (NaN == NaN: x86 and gfx950 produce different NaN payloads)."""
import numpy as np

MAX_LSB = 1
MIN_IDENTICAL = 0.999


def raw_equal_mask(a, b):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    return same.all(axis=-1)


def compare_rgb8(test, ref):
    test = np.asarray(test).astype(np.int16)
    ref = np.asarray(ref).astype(np.int16)
    assert test.shape == ref.shape, (test.shape, ref.shape)
    diff = np.abs(test - ref)
    ident = (diff == 0).all(axis=-1)
    return dict(max_lsb=int(diff.max()) if diff.size else 0,
                identical=float(ident.mean()) if ident.size else 1.0,
                differing=int((~ident).sum()))


def assert_rgb8_parity(test, ref, what=""):
    r = compare_rgb8(test, ref)
    assert r["max_lsb"] <= MAX_LSB and r["identical"] >= MIN_IDENTICAL, (what, r)
    return r
