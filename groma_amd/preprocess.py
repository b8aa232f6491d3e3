"""Image preprocessing on the device: uint8 RGB image of any size -> the f32 [3, 448, 448]
tensor `GromaModel.forward(images=...)` takes.

Reference behaviour (groma/eval/run_groma.py:78-80, groma/data/datasets/groma.py:95-96):
    raw_image = Image.open(...).convert('RGB').resize((448, 448))                       # PIL BICUBIC, aspect ignored
    image = vis_processor.preprocess(raw_image, return_tensors='pt')['pixel_values']     # x * (1/255), (x - mean) / std
The host computes only what Pillow itself computes on the host -- the per-output-pixel window and its fixed-point
coefficients (a few KB) -- with Pillow's arithmetic; the pixels never leave the GPU (csrc/preprocess.hip).
The resized 8-bit image is bit-identical to PIL's (tests/test_preprocess_gpu.py); the normalised values equal the
numpy expression of the HF processor exactly (256-entry fp32 table per channel).
No CPU fallback: this module drives the HIP library only; `pil_coefficients` is also what the tests check against PIL.
"""
import math

import numpy as np
import torch

from . import _lib, ops

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # HF IMAGENET_DEFAULT_MEAN / STD (the DINOv2 processor's defaults)
IMAGENET_STD = (0.229, 0.224, 0.225)
PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_coefficients(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter (support 2) over the full input range:
    -> (bounds int32 [out, 2] = (first input index, count), coef int32 [out, ksize])."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        n = xmax - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(n)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(n):
            v = w[x] / ww if ww != 0.0 else w[x]
            coef[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, n)
    return bounds, coef


def normalise_table(mean=IMAGENET_MEAN, std=IMAGENET_STD, rescale_factor=1.0 / 255.0):
    """fp32 [3, 256]: the HF processor's `rescale` then `normalize` evaluated for every 8-bit value with its numpy
    expression: float32(uint8 * scale) -> (x - float32(mean)) / float32(std)."""
    u = np.arange(256, dtype=np.uint8)
    x = (u * rescale_factor).astype(np.float32)
    m = np.asarray(mean, dtype=np.float32)[:, None]
    s = np.asarray(std, dtype=np.float32)[:, None]
    return ((x[None, :] - m) / s).astype(np.float32)


class ImagePreprocessor:
    def __init__(self, size=448, mean=IMAGENET_MEAN, std=IMAGENET_STD, rescale_factor=1.0 / 255.0, device="cuda"):
        self.size, self.device = int(size), torch.device(device)
        self.lut = torch.from_numpy(normalise_table(mean, std, rescale_factor)).to(self.device).contiguous()
        self._coef = {}

    def _tables(self, n):
        t = self._coef.get(n)
        if t is None:
            b, c = pil_coefficients(n, self.size)
            t = (torch.from_numpy(b).to(self.device), torch.from_numpy(c).to(self.device), c.shape[1], int(b[0, 0]),
                 int((b[:, 0] + b[:, 1]).max()))
            self._coef[n] = t
        return t

    def resize_u8(self, img):
        """uint8 [H, W, 3] (device) -> uint8 [size, size, 3]: PIL Image.resize((size, size)) of the same pixels."""
        return self._run(img, want_u8=True, want_f32=False)[0]

    def _run(self, img, want_u8, want_f32, out_f32=None):
        lib = _lib.load()
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise ValueError("expected a uint8 [H, W, 3] RGB image")
        img = img.to(self.device).contiguous()
        H, W, S = img.shape[0], img.shape[1], self.size
        bx, cx, kx, _, _ = self._tables(W)
        by, cy, ky, _, _ = self._tables(H)
        st = ops._stream()
        if W != S:  # Pillow skips a pass whose size does not change
            tmp = torch.empty((H, S, 3), dtype=torch.uint8, device=self.device)
            _lib.check(lib.gr_resize_h_u8(ops._p(img), ops._p(tmp), ops._p(bx), ops._p(cx), H, W, S, kx, st), "gr_resize_h_u8")
        else:
            tmp = img
        o8 = torch.empty((S, S, 3), dtype=torch.uint8, device=self.device) if want_u8 else None
        of = (out_f32 if out_f32 is not None else torch.empty((3, S, S), dtype=torch.float32, device=self.device)) if want_f32 else None
        if H == S:  # identity vertical pass: window (yy, 1) with coefficient 1.0
            by = torch.stack([torch.arange(S, dtype=torch.int32), torch.ones(S, dtype=torch.int32)], 1).to(self.device).contiguous()
            cy = torch.full((S, 1), 1 << PRECISION_BITS, dtype=torch.int32, device=self.device)
            ky = 1
        _lib.check(lib.gr_resize_v_norm(ops._p(tmp), ops._p(o8), ops._p(of), ops._p(by), ops._p(cy), ops._p(self.lut), H, S, S,
                                        ky, st), "gr_resize_v_norm")
        return o8, of

    def __call__(self, images):
        """list of uint8 [H_i, W_i, 3] tensors (any sizes; host or device) -> f32 [B, 3, size, size] on the device"""
        out = torch.empty((len(images), 3, self.size, self.size), dtype=torch.float32, device=self.device)
        for i, im in enumerate(images):
            if not torch.is_tensor(im):
                im = torch.from_numpy(np.ascontiguousarray(np.asarray(im)))
            self._run(im, want_u8=False, want_f32=True, out_f32=out[i])
        return out


# ---------------------------------------------------------------------------------------------------------------------
# The eval datasets' route (groma/data/datasets/refcoco_rec.py:38-65 and the other mmdet-style datasets):
#   LoadImageFromFile (cv2.imread: uint8 HWC BGR) -> Resize((448, 448), keep_ratio=False) = cv2.resize INTER_LINEAR
#   -> Normalize(mean * 255, std * 255, to_rgb=True) = mmcv.imnormalize -> Pad(size_divisor=448) (no-op) -> CHW tensor.
MMDET_MEAN = (0.485 * 255, 0.456 * 255, 0.406 * 255)
MMDET_STD = (0.229 * 255, 0.224 * 255, 0.225 * 255)
_CV_COEF_SCALE = 2048  # INTER_RESIZE_COEF_SCALE (11 bits)


def cv2_linear_tables(src, dst, clamp_taps):
    """OpenCV's per-axis bilinear tables (imgproc/resize.cpp): tap index and the two 11-bit coefficients
    short(cvRound((1 - f) * 2048)), short(cvRound(f * 2048)) with f = float((d + 0.5) * scale - 0.5) - floor(.).
    clamp_taps=True is the x axis (index clamped, f zeroed at both borders); rows keep f and clip indices in the kernel."""
    scale = 1.0 / (dst / src)
    ofs = np.zeros(dst, dtype=np.int32)
    coef = np.zeros((dst, 2), dtype=np.int16)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if clamp_taps:
            if s < 0:
                s, f = 0, np.float32(0.0)
            if s >= src - 1:
                s, f = src - 1, np.float32(0.0)
        ofs[d] = s
        coef[d, 0] = int(np.rint(np.float32(np.float32(np.float32(1.0) - f) * np.float32(_CV_COEF_SCALE))))  # cvRound
        coef[d, 1] = int(np.rint(np.float32(f * np.float32(_CV_COEF_SCALE))))
    return ofs, coef


class MmdetTestPipeline:
    """uint8 [H, W, 3] **BGR** images (what cv2.imread / LoadImageFromFile yields; any size) -> f32 [B, 3, 448, 448] on the
    device, through one fused kernel per image (csrc/preprocess.hip::cv2_resize_norm_kernel)."""

    def __init__(self, size=448, mean=MMDET_MEAN, std=MMDET_STD, to_rgb=True, device="cuda"):
        self.size, self.device, self.to_rgb = int(size), torch.device(device), bool(to_rgb)
        self.mean = torch.tensor(mean, dtype=torch.float64, device=self.device)
        self.stdinv = (1.0 / torch.tensor(std, dtype=torch.float64)).to(self.device)
        self._tab = {}

    def _tables(self, n, clamp):
        key = (n, clamp)
        if key not in self._tab:
            o, c = cv2_linear_tables(n, self.size, clamp)
            self._tab[key] = (torch.from_numpy(o).to(self.device), torch.from_numpy(c).to(self.device).contiguous())
        return self._tab[key]

    def _run(self, img, out_f32=None, want_u8=False):
        lib = _lib.load()
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise ValueError("expected a uint8 [H, W, 3] BGR image")
        img = img.to(self.device).contiguous()
        H, W, S = img.shape[0], img.shape[1], self.size
        xo, xa = self._tables(W, True)
        yo, yb = self._tables(H, False)
        o8 = torch.empty((S, S, 3), dtype=torch.uint8, device=self.device) if want_u8 else None
        _lib.check(lib.gr_cv2_resize_norm(ops._p(img), H, W, ops._p(xo), ops._p(xa), ops._p(yo), ops._p(yb), ops._p(o8),
                                          ops._p(out_f32), ops._p(self.mean), ops._p(self.stdinv), int(self.to_rgb), S, S,
                                          ops._stream()), "gr_cv2_resize_norm")
        return o8

    def resize_u8(self, img):
        """cv2.resize(img, (size, size), interpolation=cv2.INTER_LINEAR) of the same pixels"""
        return self._run(img, None, want_u8=True)

    def __call__(self, images):
        out = torch.empty((len(images), 3, self.size, self.size), dtype=torch.float32, device=self.device)
        for i, im in enumerate(images):
            if not torch.is_tensor(im):
                im = torch.from_numpy(np.ascontiguousarray(np.asarray(im)))
            self._run(im, out[i])
        return out
