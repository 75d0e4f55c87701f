"""The lane-level model of the wave decoder (tests/model_decoder.py) against
the oracle: every stream the model accepts must decode to the oracle's bytes,
and wherever it stops ("irregular": stream tail, failed check) the output it
has produced so far must be a prefix of the truth, so that the sequential
decoder can take over at (s, d)."""
import random

import pytest

import foreign
import model_decoder as M2
import model_decoder3 as M3
import oracle_lib as O

M = M2


@pytest.fixture(params=[2, 3], autouse=True)
def generation(request):
    """every test runs over both kernels' models"""
    global M
    M = M2 if request.param == 2 else M3
    return request.param


def check(comp, data=None):
    try:
        want = O.decompress(comp)
    except O.SnapError:
        want = None
    st = M.Stats()
    res = M.decode(comp, st)
    if res[0] == "ok":
        assert want is not None, "model accepted what the oracle rejects"
        assert res[1] == want
    else:
        _, s, d, prefix = res
        if want is not None:
            assert prefix == want[:d], (s, d)
            if M is M3:
                # a valid stream leaves the third generation's wide path only
                # for its last kTail bytes of input
                hdr = M3.read_varint(comp)[1]
                assert len(comp) - hdr - s < M3.TAIL, (s, len(comp))
                return "ok", st
    if data is not None:
        assert want == data
    return res[0], st


def test_model_on_corpus():
    total = M.Stats()
    for name, data in O.corpus_round():
        kind, st = check(O.compress(data), data)
        assert kind == "ok"      # a valid stream never leaves the wide path
        total.windows += st.windows
        total.elements += st.elements
        if M is M2:
            total.rounds += st.rounds
        else:
            total.runs += st.runs
            total.trips += st.trips
            total.sweeps += st.sweeps
    assert total.windows > 1000
    if M is M2:
        # dependency rounds per window stay close to one on this corpus
        assert total.rounds / total.windows < 2.5
    else:
        print("windows", total.windows, "elements/window",
              total.elements / total.windows, "runs/window",
              total.runs / total.windows, "trips/window",
              total.trips / total.windows, "sweeps/window",
              total.sweeps / total.windows)
        assert total.elements / total.windows > 30
        assert total.sweeps / total.windows < 8


def test_model_on_structured_and_random():
    rng = random.Random(12)
    cases = [b"", b"a", b"a" * 100000, b"ab" * 50000, b"abc" * 40000,
             bytes(300000), bytes(range(256)) * 300,
             b"".join(bytes([rng.randrange(4)]) * rng.randrange(1, 40)
                      for _ in range(5000))]
    for _ in range(25):
        alpha = rng.choice([1, 2, 3, 4, 16, 256])
        n = rng.choice([200, 1000, 65536, 70000, rng.randrange(1, 150000)])
        cases.append(bytes(rng.choices(range(alpha), k=n)))
    for data in cases:
        kind, _ = check(O.compress(data), data)
        assert kind == "ok" or len(O.compress(data)) < 12, len(data)


def test_model_on_foreign_streams_and_long_literals():
    jpg = (O.CORPUS / "fireworks.jpeg").read_bytes()
    check(O.compress(jpg * 3), jpg * 3)          # 64 KiB literals
    # streams no 64 KiB-block encoder writes: copy-4, offsets beyond 64 KiB,
    # overlapping copies of every period, non-minimal literal lengths
    for comp, data in foreign.cases():
        assert check(comp, data)[0] == "ok"
    # elements that end exactly at, or are cut off by, the end of the input
    txt = (O.CORPUS / "alice29.txt").read_bytes()[:3000]
    comp = O.compress(txt)
    for cut in range(1, 40):
        check(comp[:-cut])


def test_model_on_corrupted_streams():
    rng = random.Random(4)
    base = O.compress((O.CORPUS / "html").read_bytes())
    for _ in range(150):
        bad = bytearray(base)
        for _ in range(rng.choice([1, 1, 2, 5])):
            bad[rng.randrange(3, len(bad))] = rng.randrange(256)
        check(bytes(bad))
    for cut in (4, 100, 1000, len(base) - 1):
        check(base[:cut])
