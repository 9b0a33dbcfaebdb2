"""Corpus of Radiance RGBE (.hdr) byte strings -- well-formed, odd and damaged -- shared by tests/test_rgbe.py (live
differential test against the reference's own Common/rgbe.c in oracle/_ref) and make_rgbe_ref_vectors.py (which stores
what that reference reader answers for each of them as tests/golden/rgbe_ref_vectors.npz).  The files are written by a
small encoder of this module's own (neither the product's nor the reference's), so that the three sides are independent."""
import numpy as np

HDR = b"#?RADIANCE\n"
FMT = b"FORMAT=32-bit_rle_rgbe\n"


def header(w, h, magic=HDR, before=b"", after=b"", blank=b"\n", size=None):
    return magic + before + FMT + after + blank + (size if size is not None else f"-Y {h} +X {w}\n".encode())


def rle_channel(d, mode):
    """one channel of one scanline; mode 'runs': greedy runs >= 2 (max 127) + literal blocks (max 128); 'lit': literals only"""
    out, i, n = b"", 0, len(d)
    while i < n:
        r = 1
        while mode == "runs" and i + r < n and r < 127 and d[i + r] == d[i]:
            r += 1
        if r >= 2:
            out += bytes([128 + r, d[i]]); i += r
            continue
        j = i
        while j < n and j - i < 128 and (mode == "lit" or j + 1 >= n or d[j + 1] != d[j]):
            j += 1
        j = max(j, i + 1)
        out += bytes([j - i]) + bytes(d[i:j]); i = j
    return out


def rle_body(px, mode="runs"):
    """px: uint8 [h, w, 4] -> new-style run-length scanlines"""
    h, w, _ = px.shape
    out = b""
    for y in range(h):
        out += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            out += rle_channel(px[y, :, c], mode)
    return out


def pixels(w, h, seed, patchy=True):
    rng = np.random.default_rng(seed)
    px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    px[..., 3] = 120 + rng.integers(0, 16, (h, w))
    if patchy:       # constant stretches, so that run-length coding has runs to find, and some black (e = 0) pixels
        for y in range(h):
            x = 0
            while x < w:
                n = int(rng.integers(1, max(2, w // 2)))
                if rng.random() < 0.6:
                    px[y, x:x + n] = px[y, x]
                if rng.random() < 0.1:
                    px[y, x:x + n, 3] = 0
                x += n
    return px


def corpus(fuzz=300):
    """list of (name, bytes)"""
    out = []
    add = lambda name, b: out.append((name, bytes(b)))
    p16 = pixels(16, 3, 1)
    p200 = pixels(200, 2, 2)          # runs longer than 127 and literal blocks longer than 128 get split
    p300 = pixels(300, 2, 3, patchy=False)
    flat5 = pixels(5, 4, 4)
    # ---- well-formed
    add("flat_w5", header(5, 4) + flat5.tobytes())
    add("flat_w7", header(7, 2) + pixels(7, 2, 5).tobytes())
    add("rle_w8", header(8, 2) + rle_body(pixels(8, 2, 6)))
    add("rle_w16_runs", header(16, 3) + rle_body(p16))
    add("rle_w16_literals", header(16, 3) + rle_body(p16, "lit"))
    add("rle_w200", header(200, 2) + rle_body(p200))
    add("rle_w300_literals", header(300, 2) + rle_body(p300, "lit"))
    add("flat_data_in_rle_width", header(16, 3) + p16.tobytes().replace(b"\x02\x02", b"\x02\x03"))
    wide = pixels(0x8000, 1, 7)
    add("flat_w32768", header(0x8000, 1) + wide.tobytes())                   # width > 0x7fff: never run-length encoded
    add("rle_w32767", header(0x7fff, 1) + rle_body(np.repeat(pixels(1, 1, 8), 0x7fff, axis=1)))
    add("trailing_bytes", header(5, 4) + flat5.tobytes() + b"garbage")
    # ---- header variants (rgbe.c:134-190)
    add("no_magic", header(5, 4, magic=b"") + flat5.tobytes())
    add("other_first_line", header(5, 4, magic=b"hello world\n") + flat5.tobytes())
    add("gamma_exposure_before_format", header(5, 4, before=b"# made by a test\nGAMMA=2.2\nEXPOSURE=0.5\n") + flat5.tobytes())
    add("exposure_after_format", header(5, 4, after=b"EXPOSURE=1.0\n") + flat5.tobytes())
    add("comment_after_format", header(5, 4, after=b"# c\n") + flat5.tobytes())
    add("two_blank_lines", header(5, 4, blank=b"\n\n") + flat5.tobytes())
    add("no_blank_line", header(5, 4, blank=b"") + flat5.tobytes())
    add("blank_line_before_format", header(5, 4, before=b"\n") + flat5.tobytes())
    add("no_format", HDR + b"\n-Y 4 +X 5\n" + flat5.tobytes())
    add("format_other", HDR + b"FORMAT=32-bit_rle_xyze\n\n-Y 4 +X 5\n" + flat5.tobytes())
    add("format_crlf", HDR + b"FORMAT=32-bit_rle_rgbe\r\n\r\n-Y 4 +X 5\r\n" + flat5.tobytes())
    add("format_trailing_space", HDR + b"FORMAT=32-bit_rle_rgbe \n\n-Y 4 +X 5\n" + flat5.tobytes())
    add("long_comment_200", header(5, 4, before=b"#" + b"x" * 200 + b"\n") + flat5.tobytes())
    add("long_comment_126", header(5, 4, before=b"#" + b"x" * 125 + b"\n") + flat5.tobytes())       # 127 bytes with the newline
    add("long_comment_127", header(5, 4, before=b"#" + b"x" * 126 + b"\n") + flat5.tobytes())       # the newline becomes a line of its own
    add("long_line_ending_in_format", header(5, 4, before=b"y" * 127) + flat5.tobytes())             # 127 bytes, then FORMAT= starts a fresh fgets
    add("nul_in_comment", header(5, 4, before=b"# a\x00b\n") + flat5.tobytes())
    add("nul_line", header(5, 4, before=b"\x00abc\n") + flat5.tobytes())
    add("nul_after_format", HDR + b"FORMAT=32-bit_rle_rgbe\n\x00\n\n-Y 4 +X 5\n" + flat5.tobytes())
    add("size_extra_text", header(5, 4, size=b"-Y 4 +X 5 and more\n") + flat5.tobytes())
    add("size_leading_space", header(5, 4, size=b" -Y 4 +X 5\n") + flat5.tobytes())
    add("size_tabs", header(5, 4, size=b"-Y\t4\t+X\t5\n") + flat5.tobytes())
    add("size_no_spaces", header(5, 4, size=b"-Y4+X5\n") + flat5.tobytes())
    add("size_plus_y", header(5, 4, size=b"+Y 4 +X 5\n") + flat5.tobytes())
    add("size_x_first", header(5, 4, size=b"+X 5 -Y 4\n") + flat5.tobytes())
    add("size_only_y", header(5, 4, size=b"-Y 4\n") + flat5.tobytes())
    add("size_no_newline_at_eof", header(5, 4, size=b"-Y 4 +X 5"))
    add("size_zero_width", header(0, 4) + flat5.tobytes())
    add("size_zero_height", header(5, 0) + flat5.tobytes())
    add("size_negative", header(-5, 4) + flat5.tobytes())
    add("size_huge", header(30000, 30000) + flat5.tobytes())
    add("empty", b"")
    add("magic_only", HDR)
    add("header_only", header(5, 4))
    add("ppm", b"P6 1 1 255\n\x00\x00\x00")
    # ---- damaged pixel data (rgbe.c:327-412)
    good = header(16, 3) + rle_body(p16)
    hl = len(header(16, 3))
    for cut in (1, 3, 4, 5, 6, 9, 20, len(good) - hl - 1):
        add(f"rle_truncated_{cut}", good[:hl + cut])
    add("flat_truncated", (header(5, 4) + flat5.tobytes())[:-1])
    add("flat_truncated_to_header", header(5, 4) + flat5.tobytes()[:3])
    add("rle_wrong_width", header(16, 1) + bytes([2, 2, 0, 17]) + bytes(80))
    add("rle_width_high_bit", header(16, 1) + bytes([2, 2, 0x80, 16]) + bytes(80))         # (rgbe[2] & 0x80): read flat
    add("rle_run_overruns", header(16, 1) + bytes([2, 2, 0, 16, 128 + 17, 9]) + bytes(80))
    add("rle_literal_overruns", header(16, 1) + bytes([2, 2, 0, 16, 17]) + bytes(80))
    add("rle_zero_count", header(16, 1) + bytes([2, 2, 0, 16, 0, 0]) + bytes(80))
    add("rle_count_128_literal", header(200, 1) + bytes([2, 2, 0, 200, 128]) + bytes(range(128)) + bytes([72]) + bytes(72) + (bytes([128 + 100, 7]) * 2) * 3)
    add("rle_second_scanline_flat", header(16, 3) + rle_body(p16[:1]) + p16[1:].tobytes().replace(b"\x02\x02", b"\x03\x02"))
    add("rle_run_then_eof", header(16, 1) + bytes([2, 2, 0, 16, 128 + 16]))
    # ---- seeded mutations of three good files
    rng = np.random.default_rng(12345)
    seeds = [good, header(5, 4, before=b"GAMMA=1\n") + flat5.tobytes(), header(200, 2) + rle_body(p200)]
    for k in range(fuzz):
        b = bytearray(seeds[k % 3])
        for _ in range(int(rng.integers(1, 4))):
            if not b:
                break
            op, at = int(rng.integers(0, 4)), int(rng.integers(0, len(b)))
            if k % 2 == 0:
                at = min(at, int(rng.integers(0, 60)))          # half of them in or near the header
            if op == 0:
                b[at] = int(rng.integers(0, 256))
            elif op == 1:
                del b[at]
            elif op == 2:
                b.insert(at, int(rng.integers(0, 256)))
            else:
                del b[at:]
        add(f"fuzz_{k}", b)
    return out
