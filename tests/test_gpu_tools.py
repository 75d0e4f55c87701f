"""tools/szip (the szip-like front end over the host pipeline):
stream bytes equal to the oracle's restatement of write::FrameEncoder (what
the reference's szip produces for file input: 65536-byte chunks), round trips,
file naming and flags of szip/main.rs, errors behind the good bytes."""
import os
import subprocess
from pathlib import Path

import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
SZIP = ROOT / "tools" / "szip"


def run(args, data=None, **kw):
    return subprocess.run([str(SZIP)] + args, input=data, capture_output=True,
                          timeout=300, **kw)


def test_szip_stdin_stdout_equals_reference_framing(built):
    data = b"".join(d for _, d in O.corpus_round())
    p = run([], data)
    assert p.returncode == 0, p.stderr
    assert p.stdout == O.frame_compress(data)
    q = run(["-d"], p.stdout)
    assert q.returncode == 0 and q.stdout == data
    assert run([], b"").stdout == b""                # nothing in, nothing out
    assert run(["-d"], b"").stdout == b""


def test_szip_many_slabs_through_the_pipeline(built, tmp_path):
    """200 MiB = four 64 MiB slabs, two workers: slabs leave in order, the
    identifier is written once, the decoder cuts its slabs at chunk
    boundaries and carries the tail."""
    one = b"".join(d for _, d in O.corpus_round())
    data = (one * 72)[:200 << 20]
    f = tmp_path / "big.bin"
    f.write_bytes(data)
    p = run(["-k", "-v", str(f)])
    assert p.returncode == 0, p.stderr
    sz = tmp_path / "big.bin.sz"
    framed = sz.read_bytes()
    # chunk for chunk the oracle's bytes (spot checks: the oracle is slow)
    head = O.frame_compress(data[:1 << 20])
    assert framed[:len(head)] == head
    tailpos = (len(data) // 65536 - 8) * 65536
    tail = O.frame_compress(data[tailpos:])[10:]
    assert framed.endswith(tail)
    assert framed.count(b"\xff\x06\x00\x00sNaPpY") == 1
    st = f.stat()
    assert abs(sz.stat().st_mtime - st.st_mtime) < 2      # times preserved
    f.unlink()
    q = run(["-d", "-j", "3", str(sz)])
    assert q.returncode == 0, q.stderr
    assert not sz.exists()                                 # removed like gzip
    assert f.read_bytes() == data


def test_szip_flags_and_naming(built, tmp_path):
    f = tmp_path / "a.txt"
    f.write_bytes(b"hello " * 1000)
    assert run(["-k", str(f)]).returncode == 0
    assert run(["-k", str(f)]).returncode != 0            # exists, no -f
    assert run(["-k", "-f", str(f)]).returncode == 0
    assert run([str(tmp_path / "a.txt.sz")]).returncode != 0   # already .sz
    assert run(["-d", str(f)]).returncode != 0            # not .sz
    assert run([str(tmp_path)]).returncode != 0           # a directory
    # raw format (szip --raw): one raw stream
    r = run(["-r"], f.read_bytes())
    assert r.stdout == O.compress(f.read_bytes())
    assert run(["-r", "-d"], r.stdout).stdout == f.read_bytes()


def test_szip_error_after_the_good_chunks(built):
    data = (O.CORPUS / "alice29.txt").read_bytes()
    framed = bytearray(O.frame_compress(data))
    framed[len(framed) - 200] ^= 0xFF                      # in the last chunk
    with pytest.raises(O.SnapError) as oe:
        O.frame_decompress(bytes(framed))
    p = run(["-d"], bytes(framed))
    assert p.returncode != 0
    assert p.stdout == data[:131072]                       # two good chunks
    # the reference's Display text (src/error.rs:249-335; szip/main.rs:75-82)
    import rust_snappy_amd as R
    e = oe.value
    assert R.error.Error(e.kind, e.a, e.b, e.c).display().encode() in p.stderr
    p = run(["-d"], bytes(O.frame_compress(data)[:-5]))
    assert p.returncode != 0 and b"failed to fill whole buffer" in p.stderr
    assert p.stdout == data[:131072]
