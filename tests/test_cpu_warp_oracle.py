"""Known-answer tests of the oracle's perspective warp (oracle/csrc/oracle_kernels.c: xo_warp_perspective_u8 / _f32 — the plain-C restatement of
OpenCV's documented INTER_LINEAR / BORDER_CONSTANT fixed-point scheme; PARITY UNPINNED vs OpenCV itself, which is absent from the reference tree and the
image).  These pin the restatement to the closed-form cases the scheme must satisfy, independently of the HIP kernel it later checks."""
import numpy as np

from oracle import xpoint_oracle as xo


def _img(h, w, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)


def test_identity_and_integer_translation_are_exact_copies():
    img = _img(48, 200)
    assert np.array_equal(xo.warp_perspective(img, np.eye(3)), img)
    assert np.array_equal(xo.warp_perspective(img.astype(np.float32), np.eye(3)), img.astype(np.float32))
    T = np.array([[1, 0, 5], [0, 1, 3], [0, 0, 1.0]])            # dst(x, y) = src(x - 5, y - 3); zeros where the source is outside
    o = xo.warp_perspective(img, T)
    assert np.array_equal(o[3:, 5:], img[:-3, :-5]) and not o[:3].any() and not o[:, :5].any()
    o = xo.warp_perspective(img, np.linalg.inv(T), inverse_map=True)          # WARP_INVERSE_MAP with the inverse = the same picture
    assert np.array_equal(o[3:, 5:], img[:-3, :-5])


def test_fractional_shift_weights_and_rounding():
    img = _img(16, 40, 1)
    f = img.astype(np.float32)
    # shift by 1/4 pixel in x: dst(x) = 0.75 src(x) + 0.25 src(x - 1)   (ax = 24 -> weights (8, 24) / 32)
    T = np.array([[1, 0, 0.25], [0, 1, 0], [0, 0, 1.0]])
    of = xo.warp_perspective(f, T)
    assert np.array_equal(of[:, 1:], f[:, 1:] * np.float32(0.75) + f[:, :-1] * np.float32(0.25))
    ou = xo.warp_perspective(img, T)
    ref = (img[:, 1:].astype(np.int64) * 24576 + img[:, :-1].astype(np.int64) * 8192 + 16384) >> 15       # 15-bit weights, round half up
    assert np.array_equal(ou[:, 1:], ref.astype(np.uint8))
    # the border column mixes with the constant 0
    assert np.array_equal(ou[:, 0], ((img[:, 0].astype(np.int64) * 24576 + 16384) >> 15).astype(np.uint8))
    # coordinates are quantised to 1/32 pixel, round half to even: a shift of 1/64 rounds to 0 (X = 32 x - 0.5 -> even), 3/64 to 2/32
    T[0, 2] = 1.0 / 64
    assert np.array_equal(xo.warp_perspective(f, T), f)
    T[0, 2] = 3.0 / 64
    of = xo.warp_perspective(f, T)
    assert np.array_equal(of[:, 1:], f[:, 1:] * np.float32(30 / 32) + f[:, :-1] * np.float32(2 / 32))


def test_out_of_image_singular_and_projective_cases():
    img = _img(33, 70, 2)
    far = np.array([[1, 0, 1000.0], [0, 1, 0], [0, 0, 1.0]])
    assert not xo.warp_perspective(img, far).any()
    sing = np.array([[1, 2, 3], [2, 4, 6], [0, 0, 1.0]])          # det = 0 -> zero matrix -> W = 0 -> every pixel reads source (0, 0) with weight 1
    assert np.all(xo.warp_perspective(img, sing) == img[0, 0])
    # a projective map: spot-check one pixel against the definition evaluated here (inverse map in double, 1/32 quantisation, 4 taps)
    M = np.array([[0.9, 0.05, 4.0], [-0.04, 1.1, 2.0], [2e-4, -1e-4, 1.0]])
    o = xo.warp_perspective(img.astype(np.float32), M)
    Mi = np.linalg.inv(M)
    for (x, y) in ((10, 7), (64, 20), (69, 32)):                  # incl. a pixel of the second 64-wide block
        v = Mi @ np.array([x, y, 1.0])
        X, Y = int(np.rint(v[0] / v[2] * 32)), int(np.rint(v[1] / v[2] * 32))
        sx, sy, ax, ay = X >> 5, Y >> 5, (X & 31) / 32.0, (Y & 31) / 32.0
        def tap(px, py):
            return float(img[py, px]) if 0 <= px < 70 and 0 <= py < 33 else 0.0
        ref = tap(sx, sy) * (1 - ay) * (1 - ax) + tap(sx + 1, sy) * (1 - ay) * ax + tap(sx, sy + 1) * ay * (1 - ax) + tap(sx + 1, sy + 1) * ay * ax
        assert abs(float(o[y, x]) - ref) < 1e-3, (x, y, float(o[y, x]), ref)


def test_multichannel_dsize_and_quantisation_helper():
    rgb = np.random.default_rng(3).integers(0, 256, (20, 30, 3), dtype=np.uint8)
    M = np.array([[1.05, 0.02, 1.5], [0.01, 0.97, -0.75], [0, 0, 1.0]])
    o = xo.warp_perspective(rgb, M, (50, 12))
    assert o.shape == (12, 50, 3)
    for c in range(3):
        assert np.array_equal(o[..., c], xo.warp_perspective(np.ascontiguousarray(rgb[..., c]), M, (50, 12)))
    g = np.array([[-0.5, 0.0, 0.5, 0.999, 1.0, 7.0]], dtype=np.float32)
    assert xo.to_u8_image(g).tolist() == [[0, 0, 127, 254, 255, 255]]
