"""Host half of the on-device preprocessing (no GPU): the restated Pillow coefficient tables reproduce PIL's own
`Image.resize((448, 448))` bit for bit when driven through a numpy model of the two integer passes, and the
rescale+normalise table equals the HF image processor's output.  PIL / transformers ARE the reference's dependencies
for this step (groma/eval/run_groma.py:78-80), so this pins the oracle to the real thing."""
import numpy as np
import pytest

from groma_amd.preprocess import pil_coefficients, normalise_table, PRECISION_BITS

PIL = pytest.importorskip("PIL.Image")


def numpy_resize(img, S):
    """the arithmetic of csrc/preprocess.hip in numpy (test-side model, int64 accumulators cannot overflow)"""
    def one_pass(a, n_in):
        b, c = pil_coefficients(n_in, S)
        out = np.zeros((a.shape[0], S, 3), dtype=np.uint8)
        for xx in range(S):
            x0, n = b[xx]
            acc = (a[:, x0:x0 + n, :].astype(np.int64) * c[xx, :n, None].astype(np.int64)).sum(1) + (1 << (PRECISION_BITS - 1))
            out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
        return out
    H, W, _ = img.shape
    t = one_pass(img, W) if W != S else img
    return one_pass(t.transpose(1, 0, 2), H).transpose(1, 0, 2) if H != S else t


@pytest.mark.parametrize("H,W", [(480, 640), (333, 500), (700, 37), (448, 448), (200, 448), (100, 120)])
def test_coefficients_reproduce_pil_resize(H, W):
    img = np.random.default_rng(H * 1000 + W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    ref = np.asarray(PIL.fromarray(img).resize((448, 448)))
    assert np.array_equal(numpy_resize(img, 448), ref)


def test_coefficient_tables_are_normalised_and_fit_int32():
    for n in (37, 448, 640, 4000):
        b, c = pil_coefficients(n, 448)
        assert (b[:, 0] >= 0).all() and ((b[:, 0] + b[:, 1]) <= n).all() and (b[:, 1] <= c.shape[1]).all()
        assert np.abs(c.sum(1) - (1 << PRECISION_BITS)).max() <= c.shape[1]  # rounding of each tap
        assert np.abs(c).max() * 255 * 1.0 < 2 ** 31  # a single product fits; sums stay < 2^31 (|sum k| ~ 2^22 * 255 * 1.3)


def test_normalise_table_equals_hf_processor():
    tr = pytest.importorskip("transformers")
    try:
        proc = tr.BitImageProcessor(do_resize=False, do_center_crop=False, do_rescale=True, rescale_factor=1 / 255,
                                    do_normalize=True, image_mean=[0.485, 0.456, 0.406], image_std=[0.229, 0.224, 0.225],
                                    do_convert_rgb=True)
    except Exception as e:  # pragma: no cover
        pytest.skip(f"image processor unavailable: {e}")
    a = np.random.default_rng(3).integers(0, 256, (64, 48, 3), dtype=np.uint8)
    out = proc.preprocess(PIL.fromarray(a), return_tensors="np")["pixel_values"][0]
    lut = normalise_table()
    ref = np.stack([lut[c][a[..., c]] for c in range(3)])
    assert np.array_equal(np.asarray(out, dtype=np.float32), ref)


def test_cv2_tables_and_oracle_properties():
    """the mmdet / cv2 route (oracle/cv2_pipeline.py; parity unpinned: OpenCV is not in this image): the host tables of the
    product equal the oracle's, and the restated resize has the properties OpenCV's has -- identity at equal size, constants
    preserved, exact 2x down-scale = 2x2 box average, coefficients sum to 2048 +- 1, Normalize = mmcv's double arithmetic."""
    from groma_amd import preprocess as PP
    from oracle import cv2_pipeline as CV
    for src, dst in ((640, 448), (480, 448), (37, 448), (448, 448), (1333, 448)):
        o1, c1 = PP.cv2_linear_tables(src, dst, True)
        o2, c2 = CV.linear_tables(src, dst)
        assert np.array_equal(o1, o2) and np.array_equal(c1, c2)
        o3, c3 = PP.cv2_linear_tables(src, dst, False)
        o4, c4 = CV.linear_tables_y(src, dst)
        assert np.array_equal(o3, o4) and np.array_equal(c3, c4)
        assert np.all(np.abs(c1.astype(int).sum(1) - 2048) <= 1) and o1.min() >= 0 and o1.max() <= src - 1
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    assert np.array_equal(CV.resize_linear_u8(img, 61, 83), img)
    assert (CV.resize_linear_u8(np.full((20, 30, 3), 77, np.uint8), 448, 448) == 77).all()
    big = rng.integers(0, 256, (896, 896, 3), dtype=np.uint8)
    s = big.astype(np.int32)
    assert np.array_equal(CV.resize_linear_u8(big, 448, 448), ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2))
    x = CV.imnormalize(img, PP.MMDET_MEAN, PP.MMDET_STD, True)
    ref = (img[..., ::-1].astype(np.float64) - np.array(PP.MMDET_MEAN)) / np.array(PP.MMDET_STD)
    assert x.dtype == np.float32 and np.abs(x - ref).max() < 1e-6
