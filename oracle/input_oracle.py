"""TEST INFRASTRUCTURE -- CPU restatement (numpy) of the reference's per-sample input preparation, SURVEY 8(f) rank 3:
`core/loader.py:104-219` (`handDataset.process_data`: affine augmentation, brightness, flip with left/right swap,
BGR->RGB, /255, ImageNet normalisation, root-relative + bone-length label normalisation) and its helpers
`utils/manoutils.py:150-260` (`imgUtils.get_affine_mat`, `get_rotation_mat3d`, `data_augmentation`, `add_noise`).
Only tests/ (and bench/smoke checkers) may import this file; the product path never does.

PARITY STATUS
  * Everything except the image warp is PINNED: tests/golden/make_input_golden.py executes the reference's own
    `process_data` (with cv2 / torchvision / imgaug import stubs) and tests/test_input_pipeline.py compares this file with
    those fixtures.
  * `cv.warpAffine` lives in a third-party dependency that is absent here (opencv_python==4.7.0.72, README.md:34,
    requirements.txt:22).  `warp_affine_u8` restates its published algorithm for 8-bit INTER_LINEAR / BORDER_CONSTANT
    (modules/imgproc/src/imgwarp.cpp: `warpAffine` + `WarpAffineInvoker` + `remapBilinear`): the matrix is inverted in
    double precision, source coordinates are fixed point with 10 fractional bits (+ half of 1/32 for rounding) and reduced
    to 5 fractional bits, the four bilinear weights are 15-bit fixed point built from the 1/32-quantised fractions, the
    result is (sum + 2^14) >> 15, out-of-range taps read the border value 0.  In the golden generator the stub
    `cv2.warpAffine` IS this function, so the warp itself is "parity unpinned" against OpenCV (none to compare with); what the
    fixtures pin is every step around it.  Its GEOMETRY (forward matrix inverted inside, axis order, pixel-centre convention,
    border rule, bilinear weights) is cross-checked against an implementation that shares no code with it --
    scipy.ndimage.affine_transform, order 1, 'grid-constant' -- on the reference's own augmentation matrices: equal to 1 grey
    level wherever the four taps are inside the image (tests/test_input_pipeline.py::
    test_warp_restatement_agrees_with_an_independent_bilinear_resampler).  What stays unpinned is OpenCV's exact fixed-point
    rounding (1/32-pixel coordinates, 15-bit weights): a +-1 level question.
"""
import math

import numpy as np

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], np.float32)
BONE_LENGTH = 0.095             # dataset/dataset_utils.py:9
ROOT_JOINT = 9                  # core/loader.py:186 (middle-finger MCP)


def get_rotation_mat(center, theta):
    """utils/manoutils.py:150-169 (note the reference's pi = 3.14159)."""
    t = theta * (3.14159 / 180)
    r = np.zeros((3, 3), dtype='float32')
    r[0, 0] = math.cos(t)
    r[0, 1] = -math.sin(t)
    r[1, 0] = math.sin(t)
    r[1, 1] = math.cos(t)
    r[2, 2] = 1.0
    tt = np.matmul((np.identity(3, dtype='float32') - r), center)
    r[0, 2] = tt[0]
    r[1, 2] = tt[1]
    return r


def get_scale_mat(center, scale):
    """utils/manoutils.py:138-148."""
    s = np.identity(3, dtype='float32')
    s[0, 0] = scale
    s[1, 1] = scale
    t = np.matmul((np.identity(3, dtype='float32') - s), center)
    s[0, 2] = t[0]
    s[1, 2] = t[1]
    return s


def get_rotation_mat3d(theta):
    """utils/manoutils.py:171-180."""
    t = theta * (3.14159 / 180)
    r = np.zeros((3, 3), dtype='float32')
    r[0, 0] = math.cos(t)
    r[0, 1] = -math.sin(t)
    r[1, 0] = math.sin(t)
    r[1, 1] = math.cos(t)
    r[2, 2] = 1.0
    return r


def get_affine_mat(theta, scale, u, v, height, width):
    """utils/manoutils.py:182-194: translate(u, v) . scale-about-centre . rotate-about-centre, float32."""
    center = np.array([width / 2, height / 2, 1], dtype='float32')
    trans = np.identity(3, dtype='float32')
    trans[0, 2] = u
    trans[1, 2] = v
    return np.matmul(trans, np.matmul(get_scale_mat(center, scale), get_rotation_mat(center, theta)))


def _cv_round(x):
    """cvRound / saturate_cast<int>(double): round half to even, saturating."""
    return np.clip(np.rint(x), -2147483648.0, 2147483647.0).astype(np.int64)


def invert_affine(M):
    """The in-place inversion cv::warpAffine applies to a forward matrix (imgwarp.cpp `warpAffine`, double precision)."""
    m = np.asarray(M, np.float64).reshape(6).copy()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def warp_affine_u8(src, M, dsize):
    """cv.warpAffine(src, M, dsize) for uint8 HxWxC, flags = INTER_LINEAR, borderMode = BORDER_CONSTANT (0)."""
    return warp_with_inverse(src, invert_affine(M), dsize)


def warp_with_inverse(src, m, dsize):
    """The destination walk of cv.warpAffine given the already inverted matrix m[6] (float64)."""
    src = np.asarray(src)
    assert src.dtype == np.uint8 and src.ndim == 3
    H, W, C = src.shape
    dw, dh = dsize
    m = np.asarray(m, np.float64).reshape(6)
    AB_BITS, INTER_BITS = 10, 5
    AB_SCALE = 1 << AB_BITS
    round_delta = AB_SCALE // (1 << INTER_BITS) // 2
    xs = np.arange(dw, dtype=np.float64)
    ys = np.arange(dh, dtype=np.float64)
    adelta = _cv_round(m[0] * xs * AB_SCALE)
    bdelta = _cv_round(m[3] * xs * AB_SCALE)
    X0 = _cv_round((m[1] * ys + m[2]) * AB_SCALE) + round_delta
    Y0 = _cv_round((m[4] * ys + m[5]) * AB_SCALE) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)            # saturate_cast<short>
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    fx = X & 31
    fy = Y & 31
    # 15-bit weights from the 1/32-quantised fractions: ((32-fx)(32-fy), fx(32-fy), (32-fx)fy, fx fy) * 32 -- exact, sum 2^15
    w00 = (32 - fx) * (32 - fy) * 32
    w01 = fx * (32 - fy) * 32
    w10 = (32 - fx) * fy * 32
    w11 = fx * fy * 32

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        return v * ok[..., None]

    acc = (tap(sy, sx) * w00[..., None] + tap(sy, sx + 1) * w01[..., None] +
           tap(sy + 1, sx) * w10[..., None] + tap(sy + 1, sx + 1) * w11[..., None])
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def add_brightness(img_u8, a, b):
    """imgUtils.add_noise with noise = 0 (core/loader.py:134-137): a[3] * img + b in float64, clip, truncate to uint8."""
    out = np.asarray(a, np.float64)[None, None, :] * img_u8.astype(np.float32) + float(b)
    return np.clip(out, 0, 255.0).astype(np.uint8)


def image_tensors(img_u8):
    """core/loader.py:176-180: (BGR/255 CHW, ImageNet-normalised RGB CHW), float32."""
    ori = (img_u8.astype(np.float32) / np.float32(255)).transpose(2, 0, 1)
    rgb = (img_u8[..., ::-1].astype(np.float32) / np.float32(255)).transpose(2, 0, 1)
    norm = (rgb - IMAGENET_MEAN[:, None, None]) / IMAGENET_STD[:, None, None]
    return ori, norm.astype(np.float32)


def process_data(img, hand_dict, train, params=None, bright=None, bone_length=BONE_LENGTH):
    """core/loader.py:104-219 with the random draws made explicit: params = (theta, scale, u, v, flip), bright = (a[3], b).
    Returns the reference's 11-tuple as numpy arrays."""
    l2 = [hand_dict['left']['verts2d'], hand_dict['left']['joints2d'], hand_dict['right']['verts2d'], hand_dict['right']['joints2d']]
    l3 = [hand_dict['left']['verts3d'], hand_dict['left']['joints3d'], hand_dict['right']['verts3d'], hand_dict['right']['joints3d']]
    flip = False
    if train:
        theta, scale, u, v, flip = params
        S = img.shape[0]
        A = get_affine_mat(theta, scale, u, v, S, S)
        img = warp_affine_u8(img, A[0:2, :], (S, S))
        l2 = [np.matmul(p, A[0:2, 0:2].T) + A[0:2, 2:3].T for p in l2]
        R = get_rotation_mat3d(theta)
        l3 = [np.matmul(p, R.T) for p in l3]
        img = add_brightness(img, bright[0], bright[1])
    if flip:
        img = img[:, ::-1]
    ori, norm = image_tensors(img)
    root_left, root_right = l3[1][ROOT_JOINT], l3[3][ROOT_JOINT]
    root_rel = root_right - root_left
    l3 = [l3[0] - root_left, l3[1] - root_left, l3[2] - root_right, l3[3] - root_right]
    if bone_length is not None:
        length = (np.linalg.norm(l3[1][ROOT_JOINT] - l3[1][0]) + np.linalg.norm(l3[3][ROOT_JOINT] - l3[3][0])) / 2
        s = bone_length / length
        root_rel = root_rel * s
        l3 = [p * s for p in l3]
    root_rel = np.asarray(root_rel, np.float32).copy()
    l2 = [np.asarray(p, np.float32).copy() for p in l2]
    l3 = [np.asarray(p, np.float32).copy() for p in l3]
    if flip:
        root_rel[1:] = -root_rel[1:]                    # sic: y and z, not x (core/loader.py:205)
        for i in range(4):
            l2[i][:, 0] = img.shape[0] - l2[i][:, 0]
            l3[i][:, 0] = -l3[i][:, 0]
        v2d_r, j2d_r, v2d_l, j2d_l = l2
        v3d_r, j3d_r, v3d_l, j3d_l = l3
    else:
        v2d_l, j2d_l, v2d_r, j2d_r = l2
        v3d_l, j3d_l, v3d_r, j3d_r = l3
    return ori, norm, v2d_l, j2d_l, v2d_r, j2d_r, v3d_l, j3d_l, v3d_r, j3d_r, root_rel
