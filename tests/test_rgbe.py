"""Radiance RGBE (.hdr) codec -- the env-map file format on the input side of the path (Common/rgbe.c)."""
import numpy as np
import pytest


def hdr_bytes(w, h, body):
    return b"#?RADIANCE\n# a comment\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n" + f"-Y {h} +X {w}\n".encode() + body


def test_rgbe_known_pixels_flat(ptlib):
    # rgbe2float: mantissa * 2^(e-136) (rgbe.c:93-106); e == 0 is black
    px = bytes([128, 64, 32, 129, 255, 0, 1, 136, 10, 20, 30, 0, 200, 100, 50, 120])
    img = ptlib.rgbe_decode(hdr_bytes(4, 1, px))           # width < 8: flat pixels
    assert img.shape == (1, 4, 3)
    exp = np.array([[128, 64, 32], [255, 0, 1], [0, 0, 0], [200, 100, 50]], np.float64) * np.array([[2.0 ** -7], [1.0], [0], [2.0 ** -16]])
    assert np.array_equal(img[0], exp.astype(np.float32))


def test_rgbe_rle_scanlines(ptlib):
    w, h = 16, 2
    rng = np.random.default_rng(0)
    chans = rng.integers(0, 256, (h, 4, w), dtype=np.uint8)
    chans[:, 3] = 128 + rng.integers(-3, 3, (h, w))
    body = b""
    for y in range(h):
        body += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            d = chans[y, c]
            if c == 0:      # one run of 5 then literals
                d[:5] = d[0]
                body += bytes([128 + 5, d[0]]) + bytes([w - 5]) + d[5:].tobytes()
            else:           # literals only, in two blocks
                body += bytes([7]) + d[:7].tobytes() + bytes([w - 7]) + d[7:].tobytes()
    img = ptlib.rgbe_decode(hdr_bytes(w, h, body))
    f = np.ldexp(1.0, chans[:, 3].astype(int) - 136)
    exp = np.stack([chans[:, 0] * f, chans[:, 1] * f, chans[:, 2] * f], -1).astype(np.float32)
    assert np.array_equal(img, exp)


def test_rgbe_round_trip(ptlib):
    sky = ptlib.synthetic_sky(64, 32, 3)
    enc = ptlib.rgbe_encode(sky)
    dec = ptlib.rgbe_decode(enc)
    assert dec.shape == sky.shape
    m = sky.max(-1, keepdims=True)
    assert (np.abs(dec - sky) <= m / 128 + 1e-30).all()       # 8-bit mantissa shared exponent
    assert len(enc) < sky.size                                  # run-length coding does compress the flat sky
    # encode(decode(x)) is a fixed point
    assert ptlib.rgbe_decode(ptlib.rgbe_encode(dec)).tobytes() == dec.tobytes()
    # narrow images are written flat
    small = ptlib.rgbe_decode(ptlib.rgbe_encode(sky[:3, :5]))
    assert small.shape == (3, 5, 3)


@pytest.mark.parametrize("bad", [b"", b"P6 1 1 255\n", b"#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 16\n\x02\x02\x00\x10\x90",
                                 b"#?RGBE\n\n-Y 1 +X 1\n\x00\x00\x00\x00", b"#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 16\n\x02\x02\x00\x11" + bytes(80)])
def test_rgbe_rejects_malformed(ptlib, bad):
    with pytest.raises(ptlib.MiptError):
        ptlib.rgbe_decode(bad)
