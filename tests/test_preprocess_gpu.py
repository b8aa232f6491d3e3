"""On-device preprocessing vs PIL / the HF processor on the MI355X: bit-exact uint8 resize, exact normalised floats."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL.Image")


@pytest.mark.parametrize("H,W", [(480, 640), (333, 500), (1000, 37), (448, 448), (200, 448), (448, 300), (1536, 2048)])
def test_resize_bit_exact_vs_pil(dev, H, W):
    from groma_amd.preprocess import ImagePreprocessor
    img = np.random.default_rng(H + W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    pre = ImagePreprocessor(448, device=dev)
    got = pre.resize_u8(torch.from_numpy(img).to(dev)).cpu().numpy()
    ref = np.asarray(PIL.fromarray(img).resize((448, 448)))
    assert np.array_equal(got, ref)


def test_pipeline_equals_pil_resize_plus_processor(dev):
    from groma_amd.preprocess import ImagePreprocessor, normalise_table
    rng = np.random.default_rng(11)
    imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in [(375, 500), (640, 427), (448, 448)]]
    pre = ImagePreprocessor(448, device=dev)
    out = pre([torch.from_numpy(a) for a in imgs])
    assert out.shape == (3, 3, 448, 448) and out.dtype == torch.float32
    lut = normalise_table()
    for i, a in enumerate(imgs):
        r8 = np.asarray(PIL.fromarray(a).resize((448, 448)))
        ref = np.stack([lut[c][r8[..., c]] for c in range(3)])
        assert np.array_equal(out[i].cpu().numpy(), ref)


def test_preprocessed_batch_feeds_the_model(dev):
    """the tensor is what GromaModel.forward(images=...) consumes: same logits as feeding the host-side reference tensor"""
    from groma_amd.preprocess import ImagePreprocessor, normalise_table
    from groma_amd import synth
    from tests import util
    cfg, sd, tk = util.tiny_setup(seed=0)
    model = util.device_model(cfg, sd)
    _, ids = synth.make_inputs(cfg, tk, bs=1, seed=7)
    a = np.random.default_rng(5).integers(0, 256, (300, 400, 3), dtype=np.uint8)
    S = cfg.image_size
    pre = ImagePreprocessor(S, device=dev)
    x_dev = pre([torch.from_numpy(a)])
    r8 = np.asarray(PIL.fromarray(a).resize((S, S)))
    lut = normalise_table()
    x_ref = torch.from_numpy(np.stack([lut[c][r8[..., c]] for c in range(3)]))[None]
    torch.manual_seed(1); l1, _ = model.forward(input_ids=ids.clone(), images=x_dev)
    torch.manual_seed(1); l2, _ = model.forward(input_ids=ids.clone(), images=x_ref)
    assert torch.equal(l1, l2)


@pytest.mark.parametrize("H,W", [(480, 640), (448, 448), (896, 896), (37, 53), (1000, 333), (449, 447)])
def test_mmdet_cv2_pipeline_bit_exact_vs_oracle(dev, H, W):
    """Resize((448,448)) + Normalize(to_rgb) of the eval datasets (R: groma/data/datasets/refcoco_rec.py:38-65) on the device
    against the numpy restatement of OpenCV's 8-bit bilinear + mmcv.imnormalize: resized bytes and normalised floats
    bit-identical (integer arithmetic / IEEE double arithmetic on both sides)."""
    from groma_amd.preprocess import MmdetTestPipeline
    from oracle import cv2_pipeline as CV
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    pipe = MmdetTestPipeline(448)
    u8 = pipe.resize_u8(torch.from_numpy(img)).cpu().numpy()
    assert np.array_equal(u8, CV.resize_linear_u8(img, 448, 448))
    out = pipe([torch.from_numpy(img)])[0].cpu().numpy()
    assert np.array_equal(out, CV.mmdet_test_pipeline(img, 448))
