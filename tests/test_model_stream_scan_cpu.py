"""The level-1 table of snapmi_decompress_stream two ways (tests/
model_stream_scan.py): one walk per (segment, entry) - the definition - and
the wavefront scheme of round 4's k_stream_scan (128-byte walks of all
entries, one trunk per segment, overrunning walks that join the next trunk,
links resolved downwards).  Every entry must be the definition's, on text,
on streams with long literals (chains through literal bytes never join), on
incompressible data, and on truncated / corrupted streams (entries that cannot
be followed)."""
import random

import pytest

import model_stream_scan as M
import oracle_lib as O

# (segment bytes, segments per scan wavefront): what a 2 GiB stream gets, what
# the streams of a small call get (stream_seg_log2 / stream_scan_segs in
# snapmi_api.hip), and sizes in between
GEOMETRIES = [(4096, 64), (1024, 64), (1024, 16), (1024, 8), (4096, 32)]


@pytest.fixture(params=GEOMETRIES, ids=lambda g: f"{g[0]}x{g[1]}")
def geom(request):
    seg, group = request.param
    old = M.SEG
    M.SEG = seg
    yield group
    M.SEG = old


def _check(comp, want_links=False, group=64):
    naive = M.naive_table(comp)
    stats = {}
    pooled = M.pooled_table(comp, stats, group)
    assert naive.keys() == pooled.keys()
    for k, v in naive.items():
        got = pooled[k]
        if v[0] is None and got[0] is not None:
            # the definition gave up after 8 192 elements behind the segment;
            # a walk that joined a trunk may know the answer: it must be true
            assert M.naive_walk(comp, k[0], k[0] * M.SEG + k[1], cap=None) == got
            continue
        assert got[0] == v[0], (k, v, got)
        if v[0] is not None:
            assert got[1] == v[1], (k, v, got)
    if want_links:
        assert stats["links"] > 0 and stats["others"] > 0, stats
    return stats


def test_scan_model_on_text_and_the_corpus_mix(geom):
    rnd = dict(O.corpus_round())
    text = rnd["zflat06_txt1"] + rnd["zflat00_html"] + rnd["zflat07_txt2"]
    _check(O.compress(text), want_links=True, group=geom)   # ~57 x 4 KiB
    mix = b"".join(d[:90000] for d in rnd.values())
    _check(O.compress(mix * 2), want_links=True, group=geom)


def test_scan_model_with_long_literals_and_incompressible_data(geom):
    rng = random.Random(3)
    rnd = dict(O.corpus_round())
    noise = lambda n: bytes(rng.randrange(256) for _ in range(n))  # noqa: E731
    data = bytearray()
    for n in (70000, 61, 5000, 65536, 300):
        data += noise(n) + rnd["zflat08_txt3"][rng.randrange(50000):][:40000]
    _check(O.compress(bytes(data)), group=geom)
    _check(O.compress(rnd["zflat02_jpg"] * 2), group=geom)  # 64 KiB literals
    _check(O.compress(bytes(rng.choice(b"abcd") for _ in range(150000))),
           group=geom)


def test_scan_model_on_streams_that_cannot_be_followed(geom):
    rng = random.Random(9)
    comp = O.compress(dict(O.corpus_round())["zflat09_txt4"][:250000])
    _check(comp[:len(comp) // 2], group=geom)
    _check(comp[:-1], group=geom)
    for _ in range(3):
        b = bytearray(comp)
        for _ in range(4):
            b[rng.randrange(len(b))] = rng.randrange(256)
        _check(bytes(b), group=geom)
