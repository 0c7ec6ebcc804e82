"""The library's own PNG / baseline-JPEG decoders (csrc/host/image_decoders.cpp) against Pillow's on the same bytes.
PNG is lossless: bit-exact.  JPEG decoders may differ in the last bits of the inverse DCT, and Pillow interpolates subsampled
chroma where this decoder replicates it: exact to a few levels at 4:4:4, close on average with subsampling."""
import importlib
import io

import numpy as np
import pytest

pt = importlib.import_module("metal-pathtracer-arm64_amd")
Image = pytest.importorskip("PIL.Image")


def _png(arr, mode, **kw):
    buf = io.BytesIO()
    Image.fromarray(arr, mode).save(buf, "PNG", **kw)
    return buf.getvalue()


def test_png_colour_types_filters_and_block_types():
    rng = np.random.default_rng(0)
    smooth = np.linspace(0, 255, 64 * 48 * 3).reshape(48, 64, 3).astype(np.uint8)          # Sub / Up / Paeth filters pay off here
    noise = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)                                # ragged size, filter None mostly
    cases = [(smooth, "RGB", {}), (noise, "RGBA", {}), (noise[..., 0].copy(), "L", {}), (noise[..., :2].copy(), "LA", {}),
             (smooth, "RGB", {"compress_level": 0}), (smooth, "RGB", {"optimize": True}), (noise, "RGBA", {"compress_level": 9})]
    for arr, mode, kw in cases:
        data = _png(arr, mode, **kw)
        want = np.array(Image.open(io.BytesIO(data)).convert("RGBA"))
        assert np.array_equal(pt.decode_image(data), want), (mode, kw)
    # palette (with fewer than 256 entries), 1-bit, 16-bit (high byte kept)
    pal = Image.fromarray(smooth, "RGB").quantize(16)
    buf = io.BytesIO()
    pal.save(buf, "PNG")
    assert np.array_equal(pt.decode_image(buf.getvalue()), np.array(pal.convert("RGBA")))
    bw = Image.fromarray((rng.random((19, 33)) > 0.5).astype(np.uint8) * 255).convert("1")
    buf = io.BytesIO()
    bw.save(buf, "PNG")
    assert np.array_equal(pt.decode_image(buf.getvalue())[..., 0], np.array(bw.convert("L")))
    deep = rng.integers(0, 65536, (20, 30), dtype=np.uint16)
    buf = io.BytesIO()
    Image.fromarray(deep).save(buf, "PNG")
    assert np.array_equal(pt.decode_image(buf.getvalue())[..., 0], (deep >> 8).astype(np.uint8))


def test_png_rejects_what_it_does_not_support_and_corrupt_streams():
    data = _png(np.zeros((8, 8, 3), np.uint8), "RGB")
    with pytest.raises(pt.PtrError):
        pt.decode_image(data[:40])                                 # truncated
    broken = bytearray(data)
    broken[-30] ^= 0xFF                                            # inside the IDAT stream
    with pytest.raises(pt.PtrError):
        pt.decode_image(bytes(broken))
    with pytest.raises(pt.PtrError, match="unknown image format"):
        pt.decode_image(b"GIF89a" + bytes(32))


def test_jpeg_baseline_against_pillow():
    yy, xx = np.mgrid[0:72, 0:96]
    img = (np.stack([xx + yy, 1.5 * (xx + yy), 0.5 * (xx + yy)], axis=-1) % 256).astype(np.uint8)
    for subsampling, max_diff in ((0, 4), (1, 64), (2, 96)):       # 4:4:4, 4:2:2, 4:2:0
        for quality in (95, 60):
            buf = io.BytesIO()
            Image.fromarray(img, "RGB").save(buf, "JPEG", quality=quality, subsampling=subsampling)
            got = pt.decode_image(buf.getvalue())[..., :3].astype(int)
            want = np.array(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")).astype(int)
            diff = np.abs(got - want)
            assert got.shape == want.shape and diff.mean() < 1.5 and diff.max() <= max_diff, (subsampling, quality, diff.max(), diff.mean())
    gray = np.random.default_rng(1).integers(0, 256, (41, 57), dtype=np.uint8)      # ragged size: partial MCUs at the right / bottom edges
    buf = io.BytesIO()
    Image.fromarray(gray, "L").save(buf, "JPEG", quality=90)
    got = pt.decode_image(buf.getvalue())
    assert np.abs(got[..., 0].astype(int) - np.array(Image.open(io.BytesIO(buf.getvalue()))).astype(int)).max() <= 4 and (got[..., 3] == 255).all()
    buf = io.BytesIO()
    Image.fromarray(img, "RGB").save(buf, "JPEG", progressive=True)
    with pytest.raises(pt.PtrError, match="progressive"):
        pt.decode_image(buf.getvalue())


def test_decoders_survive_hostile_files():
    import struct
    import zlib

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    # an 8x8 RGB header whose IDAT stream expands to 64 MB of zeros (a few dozen KB compressed): refused at the size the image needs
    bomb = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 8, 8, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(64 << 20), 9)) +
            chunk(b"IEND", b""))
    with pytest.raises(pt.PtrError):
        pt.decode_image(bomb)
    # every prefix of a valid PNG and of a valid JPEG is either decoded or refused with a message - never a crash
    png = _png(np.arange(16 * 16 * 3, dtype=np.uint8).reshape(16, 16, 3), "RGB")
    buf = io.BytesIO()
    Image.fromarray(np.arange(24 * 24 * 3, dtype=np.uint8).reshape(24, 24, 3), "RGB").save(buf, "JPEG", quality=80)
    jpeg = buf.getvalue()
    for data in (png, jpeg):
        for cut in range(0, len(data), 7):
            try:
                pt.decode_image(data[:cut])
            except pt.PtrError:
                pass
    # a JPEG that ends on a scan header of length 2 (no component count behind it)
    sos = jpeg.index(b"\xff\xda")
    with pytest.raises(pt.PtrError):
        pt.decode_image(jpeg[:sos] + b"\xff\xda\x00\x02")
