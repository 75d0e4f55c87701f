"""The round order of the lane-per-block match finder (k_match_blocks) and of
its speculating variant (k_match_blocks_spec: a probe's round also reads the
entry of the probe that follows a miss), as a model: tests/model_match_lane.py
reads and forwards table entries in the kernel's order, and the stream its
tokens encode to must be the oracle's - with and without speculation, on data
where consecutive positions share a table slot (runs, tiny alphabets: every
forwarding case) as well as on text."""
import random

import pytest

import model_match_lane as M
import oracle_lib as O


def _cases():
    rng = random.Random(11)
    blob = b"".join(p.read_bytes() for p in sorted(O.CORPUS.iterdir())
                    if p.stat().st_size > 70000)
    out = []
    for n in (17, 18, 31, 32, 33, 100, 255, 256, 257, 1000, 4096, 5000):
        out.append(bytes(n))
        for alpha in (1, 2, 3, 4, 16, 256):
            out.append(bytes(rng.randrange(alpha) for _ in range(n)))
        at = rng.randrange(0, len(blob) - n)
        out.append(blob[at:at + n])
    for n in (20000, 65535, 65536):
        at = rng.randrange(0, len(blob) - n)
        out.append(blob[at:at + n])
        out.append(bytes(rng.randrange(2) for _ in range(n)))
    unit = bytes(rng.randrange(256) for _ in range(37))
    out.append((unit * 400)[:12000])
    return out


@pytest.mark.parametrize("spec", [False, True])
def test_lane_rounds_give_the_oracle_stream(spec):
    for data in _cases():
        got, _ = M.compress_one_block_stream(data, spec)
        assert got == O.compress(data), (spec, len(data), data[:24].hex())


def test_speculation_saves_rounds_on_text():
    text = (O.CORPUS / "alice29.txt").read_bytes()[:65536]
    a, plain = M.compress_one_block_stream(text, False)
    b, spec = M.compress_one_block_stream(text, True)
    assert a == b == O.compress(text)
    assert spec < 0.8 * plain, (plain, spec)
