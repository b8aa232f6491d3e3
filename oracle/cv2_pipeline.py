"""
oracle/cv2_pipeline.py -- TEST INFRASTRUCTURE ONLY.  CPU (numpy) restatement of the image pipeline of the reference's eval
datasets (R: groma/data/datasets/refcoco_rec.py:38-65; same in the other mmdet-style datasets):

    LoadImageFromFile          cv2.imread -> uint8 HWC, BGR
    Resize(img_scale=(448, 448), keep_ratio=False)    mmcv.imresize(img, (448, 448), interpolation='bilinear', backend='cv2')
                                                      = cv2.resize(img, (448, 448), interpolation=cv2.INTER_LINEAR)
                                                      (R: mmcv/mmcv/image/geometric.py:51-101)
    Normalize(mean=[.485,.456,.406]*255, std=[.229,.224,.225]*255, to_rgb=True)
                                                      mmcv.imnormalize: float32 copy, cv2.cvtColor(BGR2RGB), cv2.subtract(img, mean),
                                                      cv2.multiply(img, 1/std)   (R: mmcv/mmcv/image/photometric.py:9-45)
    Pad(size_divisor=448)      no-op at 448x448;  DefaultFormatBundle: HWC -> CHW float tensor

PARITY PIN STATUS: **parity unpinned** for the resize.  Its arithmetic lives in OpenCV (`opencv-python`, a dependency of mmcv
that the reference does not pin and that is absent from /root/reference and from this image: cv2 cannot be imported, no
golden vector exists).  What is restated here is OpenCV's published 8-bit bilinear algorithm (modules/imgproc/src/resize.cpp):
  * source coordinate  fx = float((dx + 0.5) * scale - 0.5),  sx = floor(fx),  fx -= sx;  taps clamped at the borders
    (sx < 0 -> sx = 0, fx = 0;  sx >= src - 1 -> sx = src - 1, fx = 0 in x; rows are clipped to [0, src - 1] in y);
  * fixed-point coefficients  short(cvRound((1 - fx) * 2048)), short(cvRound(fx * 2048))  (INTER_RESIZE_COEF_BITS = 11,
    cvRound = round-half-to-even);
  * horizontal pass in int32:  D = S[sx] * a0 + S[sx + 1] * a1;
  * vertical pass:  dst = uchar((((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2);
  * exact 2x down-scale in both directions is rerouted by cv::resize to the INTER_AREA fast path: (s00+s01+s10+s11+2) >> 2.
The Normalize step is pinned by the reference's own source (photometric.py above): with a non-integer per-channel scalar
OpenCV evaluates subtract / multiply in double and stores float32, i.e. y = f32(f64(f32(f64(u8) - mean)) * (1 / f64(std))).
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def linear_tables(src, dst):
    """-> (ofs int32 [dst], coef int16 [dst, 2]) for one axis; x-style border handling (fx zeroed at both ends)"""
    scale = src / dst  # double (cv::resize: scale_x = 1. / inv_scale_x with inv_scale_x = (double)dst / src)
    inv = dst / src
    scale = 1.0 / inv
    ofs = np.zeros(dst, dtype=np.int32)
    coef = np.zeros((dst, 2), dtype=np.int16)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            s, f = 0, np.float32(0.0)
        if s >= src - 1:
            s, f = src - 1, np.float32(0.0)
        ofs[d] = s
        c0 = np.float32(np.float32(1.0) - f)
        coef[d, 0] = int(np.rint(np.float32(c0 * np.float32(COEF_SCALE))))   # cvRound: half to even
        coef[d, 1] = int(np.rint(np.float32(f * np.float32(COEF_SCALE))))
    return ofs, coef


def linear_tables_y(src, dst):
    """rows: OpenCV keeps fy at the borders and clips the ROW INDICES instead (resizeGeneric_Invoker: clip(sy + k))"""
    inv = dst / src
    scale = 1.0 / inv
    ofs = np.zeros(dst, dtype=np.int32)
    coef = np.zeros((dst, 2), dtype=np.int16)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        ofs[d] = s
        c0 = np.float32(np.float32(1.0) - f)
        coef[d, 0] = int(np.rint(np.float32(c0 * np.float32(COEF_SCALE))))
        coef[d, 1] = int(np.rint(np.float32(f * np.float32(COEF_SCALE))))
    return ofs, coef


def resize_linear_u8(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR) for uint8 HWC"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, C = img.shape
    if H == out_h and W == out_w:
        return img.copy()
    if H == 2 * out_h and W == 2 * out_w:  # INTER_AREA fast path
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    xo, xa = linear_tables(W, out_w)
    yo, yb = linear_tables_y(H, out_h)
    s = img.astype(np.int32)
    x1 = np.minimum(xo + 1, W - 1)
    hor = s[:, xo, :] * xa[None, :, 0, None].astype(np.int32) + s[:, x1, :] * xa[None, :, 1, None].astype(np.int32)  # [H, out_w, C]
    y0 = np.clip(yo, 0, H - 1)
    y1 = np.clip(yo + 1, 0, H - 1)
    b0 = yb[:, 0].astype(np.int32)[:, None, None]
    b1 = yb[:, 1].astype(np.int32)[:, None, None]
    out = ((((b0 * (hor[y0] >> 4)) >> 16) + ((b1 * (hor[y1] >> 4)) >> 16) + 2) >> 2)
    return out.astype(np.uint8)  # in range by construction (convex combination of 8-bit values)


def imnormalize(img_u8_bgr, mean, std, to_rgb=True):
    """mmcv.imnormalize (R: mmcv/mmcv/image/photometric.py:9-45) -> float32 HWC"""
    x = img_u8_bgr.astype(np.float32)
    if to_rgb:
        x = x[..., ::-1]
    mean64 = np.asarray(mean, dtype=np.float64).reshape(1, 1, -1)
    stdinv = 1.0 / np.asarray(std, dtype=np.float64).reshape(1, 1, -1)
    x = (x.astype(np.float64) - mean64).astype(np.float32)
    return (x.astype(np.float64) * stdinv).astype(np.float32)


def mmdet_test_pipeline(img_u8_bgr, size=448, mean=(0.485 * 255, 0.456 * 255, 0.406 * 255),
                        std=(0.229 * 255, 0.224 * 255, 0.225 * 255)):
    """-> float32 [3, size, size] (what the dataset hands to GromaModel.forward(images=...))"""
    r = resize_linear_u8(img_u8_bgr, size, size)
    return np.ascontiguousarray(imnormalize(r, mean, std, True).transpose(2, 0, 1))
