"""Known-answer vectors restated from the reference's test crate
(test/tests.rs).  Each entry cites the lines it comes from."""

# (name, compressed bytes, expected output) -- test/tests.rs:232-317
DECODE_KATS = [
    ("decompress_copy_close_to_end_1",
     bytes([27, 0b000010_00, 1, 2, 3, 0b000_000_10, 3, 0, 0b010110_00]
           + list(range(4, 27))),
     bytes([1, 2, 3, 1] + list(range(4, 27)))),
    ("decompress_copy_close_to_end_2",
     bytes([28, 0b000010_00, 1, 2, 3, 0b000_000_10, 3, 0, 0b010111_00]
           + list(range(4, 28))),
     bytes([1, 2, 3, 1] + list(range(4, 28)))),
]

# (name, input, (variant, fields...), bad_header) -- test/tests.rs:345-466
ERROR_KATS = [
    ("err_empty", b"", ("Empty",), False),
    ("err_header_mismatch", b"\x05\x00a", ("HeaderMismatch", 5, 1), False),
    ("err_varint1", b"\xFF", ("Header",), True),
    ("err_varint2", b"\xff" * 10 + b"\x00", ("Header",), True),
    ("err_varint3", b"\x80\x80\x80\x80\x10",
     ("TooBig", 4294967296, 4294967295), True),
    ("err_lit", b"\x02\x00hi", ("CopyRead", 1, 0), False),
    ("err_lit_big1", b"\x02\xechi", ("Literal", 60, 2, 2), False),
    ("err_lit_big2a", b"\x02\xf0hi", ("Literal", 4, 2, 2), False),
    ("err_lit_big2b", b"\x02\xf0hi\x00\x00\x00", ("Literal", 105, 4, 2),
     False),
    ("err_copy1", b"\x02\x00a\x01", ("CopyRead", 1, 0), False),
    ("err_copy2a", b"\x11\x00a\x3e", ("CopyRead", 2, 0), False),
    ("err_copy2b", b"\x11\x00a\x3e\x01", ("CopyRead", 2, 1), False),
    ("err_copy3a", b"\x11\x00a\x3f", ("CopyRead", 4, 0), False),
    ("err_copy3b", b"\x11\x00a\x3f\x00", ("CopyRead", 4, 1), False),
    ("err_copy3c", b"\x11\x00a\x3f\x00\x00", ("CopyRead", 4, 2), False),
    ("err_copy3d", b"\x11\x00a\x3f\x00\x00\x00", ("CopyRead", 4, 3), False),
    ("err_copy_offset_zero", b"\x11\x00a\x01\x00", ("Offset", 0, 1), False),
    ("err_copy_offset_big", b"\x11\x00a\x01\xFF", ("Offset", 255, 1), False),
    ("err_copy_len_big", b"\x05\x00a\x1d\x01", ("CopyWrite", 11, 4), False),
    # 32-bit-only in the reference (test/tests.rs:578-589); on a 64-bit
    # target the same inputs fail the literal bounds check with these fields.
    ("err_lit_len_overflow1", b"\x11\x00\x00\xfc\xfe\xff\xff\xff",
     ("Literal", 0xFFFFFFFF, 0, 16), False),
    ("err_lit_len_overflow2", b"\x11\x00\x00\xfc\xff\xff\xff\xff",
     ("Literal", 0x100000000, 0, 16), False),
]

# quickcheck witnesses, test/tests.rs:469-504
RANDOM1 = bytes([
    0, 0, 0, 0, 1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 4, 0, 0, 0, 5, 0, 0,
    1, 1, 0, 0, 1, 2, 0, 0, 2, 1, 0, 0, 2, 2, 0, 0, 0, 6, 0, 0, 3, 1, 0,
    0, 0, 7, 0, 0, 1, 3, 0, 0, 0, 8, 0, 0, 2, 3, 0, 0, 0, 9, 0, 0, 1, 4,
    0, 0, 1, 0, 0, 3, 0, 0, 1, 0, 1, 0, 0, 0, 10, 0, 0, 0, 0, 2, 4, 0, 0,
    2, 0, 0, 3, 0, 1, 0, 0, 1, 5, 0, 0, 6, 0, 0, 0, 0, 11, 0, 0, 1, 6, 0,
    0, 1, 7, 0, 0, 0, 12, 0, 0, 3, 2, 0, 0, 0, 13, 0, 0, 2, 5, 0, 0, 0, 3,
    3, 0, 0, 0, 1, 8, 0, 0, 1, 0, 1, 0, 0, 0, 4, 1, 0, 0, 0, 0, 14, 0, 0,
    0, 1, 9, 0, 0, 0, 1, 10, 0, 0, 0, 0, 1, 11, 0, 0, 0, 1, 0, 2, 0, 0, 0,
    1, 1, 1, 0, 0, 0, 0, 5, 1, 0, 0, 0, 1, 2, 1, 0, 0, 0, 0, 0, 2, 6, 0,
    0, 0, 0, 0, 1, 12, 0, 0, 0, 0, 0, 3, 4, 0, 0, 0, 0, 0, 7, 0, 0, 0, 0,
    0, 1, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0])
RANDOM2 = bytes([10, 2, 14, 13, 0, 8, 2, 10, 2, 14, 13, 0, 0, 0, 0, 0, 0, 0,
                 0, 0, 0, 0, 0])
RANDOM3 = bytes([0, 0, 0, 4, 1, 4, 0, 0, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                 0, 0])
RANDOM4 = bytes([
    0, 0, 0, 0, 1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 4, 0, 0, 0, 5, 0, 0,
    1, 1, 0, 0, 1, 2, 0, 0, 1, 3, 0, 0, 1, 4, 0, 0, 2, 1, 0, 0, 0, 4, 0,
    1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def small_copy_inputs():  # test/tests.rs:208-216
    return [b"aaaa" + b"b" * i + b"aaaabbbb" for i in range(32)]


def small_regular_inputs():  # test/tests.rs:218-229
    out, i = [], 1
    while i < 20000:
        out.append(bytes((j % 10) + ord("a") for j in range(i)))
        i += 23
    return out


# SURVEY App-B known-answer table: sha256 of the compressed bench inputs
CORPUS_SHA256 = {
    "zflat00_html": (102400, 22843, "c7c94425c2b3516cf3d1c9824391b8453beb544f38dfdfa90eb8126103234b5a"),
    "zflat01_urls": (702087, 335492, "8d578b8cbf000930c09c7d02a6b68aa76731e436a20a036e66c40f63b10ba5ad"),
    "zflat02_jpg": (123093, 123034, "4da5e82d77ebe3d77e4f827a294562df17b5dcf37dcdb30d516ee8544d3164a6"),
    "zflat03_jpg_200": (200, 146, "499f86aacbb68236fb9a9f9b45dec443c176d57d4c4eba2072dfdf3b1d92bf73"),
    "zflat04_pdf": (102400, 85304, "ad668e5050689de4486cca4851a67b81731ff77ae920dc78da2e5fc9ca36d7e5"),
    "zflat05_html4": (409600, 92234, "11e53110e963fa6dd4ef3d726cf2d897a7ab689c8edb90ccab3f462ef21872f3"),
    "zflat06_txt1": (152089, 88034, "d9b27949428e5678cd7a4f00baaba000612d180d9028d28a6ab3a5e308272869"),
    "zflat07_txt2": (125179, 77503, "4bf8701f8c369f13e679f52e938c8630d2a2920eba4003bfeeced8522d984aa9"),
    "zflat08_txt3": (426754, 234661, "5db82d2428a5b5c747dae15c9b219fffc8093c82a9cc8263bec750d261569c09"),
    "zflat09_txt4": (481861, 319267, "30915f0a26ae2b882e7d8a6951dc3e844c8dd615b0a1a21c6dd69e8c8f958337"),
    "zflat10_pb": (118588, 23335, "84356d0f45f9cf8547834eabaa8d4ec569c3e71c505828ab3321ffbd35370d11"),
    "zflat11_gaviota": (184320, 69526, "b6513d28c84b3715f02a2697ddb3f6b56aab8f09f0b5950075762912ae5ae8d9"),
}
