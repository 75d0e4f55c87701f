"""GPU suite: the reference-side view of the native seam.  A C program written
against snappy-c.h and linked with -lsnappy (tests/seam_consumer.c: what
snappy-cpp/src/lib.rs:13-88 is to the reference's tests and its `cpp` bench
group, bench/src/bench.rs:117-153) is built against a directory in which
libsnappy.so IS libsnapmi.so - no source change, the snappy-cpp/build.rs:2
situation - and must round-trip the 12 bench inputs to the oracle's bytes,
report libsnappy's statuses for short buffers and broken streams, and keep
doing so with sixteen callers at once (round 5: concurrent calls are combined
into one batch launch)."""
import os
import random
import subprocess
import threading

import pytest

import oracle_lib as O
from conftest import ROOT

pytestmark = pytest.mark.gpu

SNAPPY_H = "/opt/conda/include"


@pytest.fixture(scope="module")
def consumer(built, tmp_path_factory):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(os.path.join(SNAPPY_H, "snappy-c.h")):
        pytest.skip("no snappy-c.h in this image")
    d = tmp_path_factory.mktemp("seam")
    lib = ROOT / "rust-snappy_amd" / "libsnapmi.so"   # the PRODUCT library
    os.symlink(lib, d / "libsnappy.so")
    exe = d / "seam_consumer"
    subprocess.check_call(
        ["gcc", "-O2", "-Wall", "-I", SNAPPY_H, "-o", str(exe),
         str(ROOT / "tests" / "seam_consumer.c"), f"-L{d}", "-lsnappy",
         "-lpthread", f"-Wl,-rpath,{d}", f"-Wl,-rpath,{lib.parent}"])
    ins = d / "in"
    ins.mkdir()
    for name, data in O.corpus_round():
        (ins / f"{name}.in").write_bytes(data)
        (ins / f"{name}.snappy").write_bytes(O.compress(data))
    # the program is bound to the GPU library, not to Google's
    ldd = subprocess.run(["ldd", str(exe)], capture_output=True,
                         text=True).stdout
    assert "libsnapmi" in ldd or str(d) in ldd, ldd
    assert "/opt/conda/lib/libsnappy" not in ldd, ldd
    return exe, ins


def test_seam_consumer_round_trips_the_bench_inputs(consumer):
    exe, ins = consumer
    r = subprocess.run([str(exe), "check", str(ins)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "seam check ok: 12 inputs" in r.stdout


def test_seam_consumer_with_sixteen_callers(consumer):
    """One call per file from 16 threads at once (every call checks its
    status and its length): the combined launches give each caller its own
    bytes, and the aggregate rate is far above one caller's."""
    exe, ins = consumer
    rows = {}
    for threads in (1, 16):
        r = subprocess.run([str(exe), "bench", str(ins), str(threads), "150"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        for line in r.stdout.splitlines():
            name, n, t, c, u = line.split()
            rows[(name, int(t))] = (float(c), float(u))
    assert len(rows) == 24, rows
    one = rows[("zflat06_txt1", 1)]
    many = rows[("zflat06_txt1", 16)]
    print("alice29.txt MB/s compress/uncompress: 1 caller", one,
          "16 callers", many)
    assert many[0] > 4 * one[0] and many[1] > 3 * one[1], (one, many)


def test_combined_calls_give_every_caller_the_oracles_bytes(built):
    """24 Python threads, each a different mix of inputs - corpus files,
    empty and one-byte inputs, random and runs - through snappy_compress /
    snappy_uncompress / validate at once: every result byte-equal to the
    oracle's, every broken stream refused, nothing crossed between callers."""
    import ctypes as C
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from rust_snappy_amd import _lib
    L = _lib.load()
    rnd = [d for _, d in O.corpus_round()]
    pool = rnd + [b"", b"a", bytes(100000), b"ab" * 70000,
                  rnd[6][:70000], rnd[2][:100], rnd[0][:65536]]
    want = [O.compress(d) for d in pool]
    errors = []

    def caller(seed):
        rng = random.Random(seed)
        try:
            for _ in range(25):
                k = rng.randrange(len(pool))
                data, comp = pool[k], want[k]
                cap = L.snappy_max_compressed_length(len(data))
                out = C.create_string_buffer(cap)
                n = C.c_size_t(cap)
                st = L.snappy_compress(data, len(data), out, C.byref(n))
                assert st == 0 and out.raw[:n.value] == comp, (k, st)
                back = C.create_string_buffer(max(len(data), 1))
                m = C.c_size_t(len(data))
                st = L.snappy_uncompress(comp, len(comp), back, C.byref(m))
                assert st == 0 and back.raw[:m.value] == data, (k, st)
                if len(comp) > 8:
                    bad = bytearray(comp)
                    bad[rng.randrange(3, len(bad))] ^= 0x55
                    try:
                        O.decompress(bytes(bad), len(data))
                        ok = True
                    except O.SnapError:
                        ok = False
                    m = C.c_size_t(len(data))
                    st = L.snappy_uncompress(bytes(bad), len(bad), back,
                                             C.byref(m))
                    assert (st == 0) == ok, (k, st, ok)
                    st = L.snappy_validate_compressed_buffer(
                        comp[:len(comp) // 2], len(comp) // 2)
                    assert st == 1, (k, st)
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=caller, args=(s,)) for s in range(24)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


@pytest.mark.gpu
@pytest.mark.parametrize("lib", ["test", "product"])
def test_single_launch_path_of_small_calls(built, lib):
    """A lone snappy_compress / snappy_uncompress of under 256 bytes runs as
    ONE kernel over pinned host memory (seam_tiny, snapmi_api.hip): every
    input length 1..255 over three alphabets compresses to the oracle's bytes
    and comes back; every small error KAT of the reference's decoder
    (test/tests.rs:345-466) is refused with SNAPPY_INVALID_INPUT; a short
    output buffer is SNAPPY_BUFFER_TOO_SMALL; a stream whose header promises
    more than 256 bytes (the batch path's) still decodes; and 300 such calls
    take a fraction of what round 5's path took (80 us each)."""
    import ctypes as C
    import random
    import time
    import kats
    import oracle_lib as O
    from rust_snappy_amd import _lib
    L = _lib.load_product() if lib == "product" else _lib.load()
    rng = random.Random(66)

    def press(d):
        cap = C.c_size_t(L.snappy_max_compressed_length(len(d)))
        out = C.create_string_buffer(cap.value)
        assert L.snappy_compress(bytes(d), len(d), out, C.byref(cap)) == 0
        return out.raw[:cap.value]

    def depress(c, room=None):
        n = C.c_size_t(0)
        if L.snappy_uncompressed_length(bytes(c), len(c), C.byref(n)) != 0:
            return 1, b""
        cap = C.c_size_t(n.value if room is None else room)
        out = C.create_string_buffer(max(cap.value, 1))
        rc = L.snappy_uncompress(bytes(c), len(c), out, C.byref(cap))
        return rc, out.raw[:cap.value] if rc == 0 else b""
    for n in range(1, 256):
        for alpha in (2, 16, 256):
            d = bytes(rng.choices(range(alpha), k=n))
            c = press(d)
            assert c == O.compress(d), (n, alpha)
            assert depress(c) == (0, d), (n, alpha)
    assert press(b"") == b"\x00" and depress(b"\x00") == (0, b"")
    for name, data, want, bad_header in kats.ERROR_KATS:
        if 0 < len(data) < 256:
            assert depress(data)[0] == 1, name
    zeros = O.compress(b"\x00" * 4096)          # 200 bytes in, 4 KiB out
    assert len(zeros) < 256 and depress(zeros) == (0, b"\x00" * 4096)
    c = press(b"hello hello hello hello")
    assert depress(c, room=5)[0] == 2
    d = bytes(rng.choices(range(4), k=200))
    c = press(d)
    t0 = time.perf_counter()
    for _ in range(300):
        assert press(d) == c
    tc = (time.perf_counter() - t0) / 300 * 1e6
    t0 = time.perf_counter()
    for _ in range(300):
        assert depress(c) == (0, d)
    td = (time.perf_counter() - t0) / 300 * 1e6
    print(f"\n200-byte calls through snappy-c.h ({lib}): compress {tc:.1f} us, "
          f"uncompress {td:.1f} us (ctypes call overhead included)")
    assert tc < 60 and td < 60, (tc, td)
