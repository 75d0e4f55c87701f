"""Raw Snappy streams no rust-snappy encoder would write: copy-4
elements, offsets beyond 64 KiB, copies that overlap their own output, every
literal length form.  A stream is built from explicit elements and comes with
the output a conforming decoder must produce (src/decompress.rs:130-343,
tag layout :415-474)."""
import random


def varint(n):
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def lit(data, form=None):
    """A literal element; `form` forces 0..4 extra length bytes."""
    n = len(data) - 1
    if form is None:
        form = 0 if n < 60 else (1 if n < 256 else (2 if n < 65536 else 3))
    if form == 0:
        assert n < 60
        return bytes([n << 2]) + data
    return bytes([(59 + form) << 2]) + n.to_bytes(form, "little") + data


def copy(offset, length, kind):
    """copy-1 (kind 1), copy-2 (kind 2) or copy-4 (kind 4)."""
    if kind == 1:
        assert 4 <= length <= 11 and offset < 2048
        return bytes([1 | ((length - 4) << 2) | ((offset >> 8) << 5),
                      offset & 0xFF])
    assert 1 <= length <= 64
    if kind == 2:
        assert offset < 65536
        return bytes([2 | ((length - 1) << 2)]) + offset.to_bytes(2, "little")
    return bytes([3 | ((length - 1) << 2)]) + offset.to_bytes(4, "little")


def build(seed, target):
    """(stream, expected output) of about `target` bytes."""
    rng = random.Random(seed)
    out = bytearray()
    body = bytearray()
    while len(out) < target:
        r = rng.random()
        if r < 0.25 or len(out) == 0:
            n = rng.choice([1, 2, 3, 15, 16, 17, 59, 60, 61, 64, 65, 255, 256,
                            257, 1000, rng.randrange(1, 5000)])
            if rng.random() < 0.02:
                n = rng.randrange(65536, 70000)  # 3 length bytes
            data = bytes(rng.randrange(256) for _ in range(min(n, 64)))
            data = (data * (n // len(data) + 1))[:n]
            form = None
            if n <= 60 and rng.random() < 0.1:
                form = rng.choice([1, 2, 3, 4])  # non-minimal length forms
            elif rng.random() < 0.05:
                form = 4
            body += lit(data, form)
            out += data
        else:
            length = rng.choice([1, 2, 3, 4, 5, 7, 8, 11, 12, 16, 31, 32, 33,
                                 63, 64, rng.randrange(1, 65)])
            far = rng.random()
            if far < 0.3:
                offset = rng.randrange(1, min(len(out), 70) + 1)  # overlaps
            elif far < 0.6:
                offset = rng.randrange(1, min(len(out), 5000) + 1)
            else:
                offset = rng.randrange(1, len(out) + 1)  # up to > 64 KiB back
            kinds = [4]
            if offset < 65536:
                kinds.append(2)
            if offset < 2048 and 4 <= length <= 11:
                kinds.append(1)
            body += copy(offset, length, rng.choice(kinds))
            for _ in range(length):
                out.append(out[-offset])
    return varint(len(out)) + bytes(body), bytes(out)


def cases():
    return [build(s, t) for s, t in
            [(1, 100), (2, 1000), (3, 5000), (4, 70000), (5, 150000),
             (6, 300000), (7, 66000), (8, 200000)]]
