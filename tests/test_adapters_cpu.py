"""The streaming adapters' HOST logic without a GPU.

rust-snappy_amd/frame.py runs the reference's state machines (where
write::FrameEncoder cuts chunks, what read::FrameDecoder does with short
reads, errors and batches) on the host and hands batches of chunks to two
device calls.  Here those two calls - and the pinned buffers - are replaced by
a stand-in built on the oracle (test infrastructure, this file only), and the
adapter tests of the GPU suite (tests/test_gpu_frame.py: the same functions,
called with a dummy context) run against it.  What this proves is the host
side: chunk boundaries, batching, carry, "good bytes before the error", no
waiting on a short read, nothing lost when the reader fails.  The device calls
themselves are the GPU suite's business."""
import ctypes as C

import pytest

import oracle_lib as O
import test_gpu_frame as G

IDENT = b"\xff\x06\x00\x00sNaPpY"


class _FakeHostBuffer:
    """HostBuffer without hipHostMalloc."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self._b = bytearray(max(self.nbytes, 1))
        self.view = memoryview(self._b)[:self.nbytes]
        self.ptr = C.addressof((C.c_char * len(self._b)).from_buffer(self._b))

    def close(self):
        self.view = None


def _fake_encode_host(ctx, data, chunk_lens, ident=True):
    """snapmi_frame_encode_host: chunks as the caller cut them."""
    out, pos = bytearray(IDENT if ident else b""), 0
    for ln in (int(x) for x in chunk_lens):
        out += O.frame_compress(bytes(data[pos:pos + ln]))[10:]
        pos += ln
    assert pos == len(data)
    return bytes(out)


def _fake_encode_host_into(ctx, buf, chunk_lens, out, ident=True):
    f = _fake_encode_host(ctx, bytes(buf), chunk_lens, ident)
    out.view[:len(f)] = f
    return len(f)


def _fake_decode_host(ctx, data, out, continuation, final, stale):
    """snapmi_frame_decode_host (include/snapmi.h): the whole chunks at the
    start of `data`, as many data chunks as len(out) / 65536 allows; on an
    error the bytes in front of the failing chunk and consumed = 0."""
    from rust_snappy_amd.error import Error
    data = bytes(data)
    pos = w = nchunks = 0
    room = len(out) // 65536
    while pos < len(data):
        rem = len(data) - pos
        ln = int.from_bytes(data[pos + 1:pos + 4], "little") if rem >= 4 else 0
        if rem < 4 or rem - 4 < ln:          # the chunk is cut off
            if final:
                return w, 0, Error(64)
            break
        is_data = data[pos] in (0, 1)
        if is_data and nchunks == room:
            break
        piece = data[pos:pos + 4 + ln]
        try:
            dec = O.frame_decompress(
                piece if not (continuation or pos) else IDENT + piece)
        except O.SnapError as e:
            return w, 0, Error(64 if e.kind == -1 else e.kind, e.a, e.b, e.c)
        out[w:w + len(dec)] = dec
        w += len(dec)
        pos += 4 + ln
        nchunks += is_data
    return w, pos, None


@pytest.fixture
def fake(monkeypatch):
    from rust_snappy_amd import frame
    monkeypatch.setattr(frame, "encode_host", _fake_encode_host)
    monkeypatch.setattr(frame, "encode_host_into", _fake_encode_host_into)
    monkeypatch.setattr(frame, "decode_host", _fake_decode_host)
    monkeypatch.setattr(frame, "HostBuffer", _FakeHostBuffer)
    return object()          # the context: only the device calls look at it


@pytest.mark.parametrize("test", [
    G.test_frame_bytes_equal_oracle_on_corpus,
    G.test_frame_random_roundtrip,
    G.test_frame_flush_boundaries,
    G.test_read_frame_encoder_big_and_little_buffers,
    G.test_frame_encoder_write_state_machine,
    G.test_frame_encoder_emits_before_flush_and_into_inner_error,
    G.test_frame_decoder_streams_in_batches,
    G.test_frame_decoder_returns_good_chunks_before_the_error,
    G.test_read_frame_encoder_chunks_follow_the_reads,
    G.test_adapters_lose_nothing_when_the_reader_fails,
    G.test_frame_decoder_does_not_wait_for_a_full_batch,
    G.test_frame_decoder_readinto,
], ids=lambda f: f.__name__)
def test_adapter_host_logic(fake, test):
    test(fake)


def test_large_write_goes_out_where_it_lies(fake):
    """A write of at least DIRECT_MIN bytes with an empty block buffer is
    framed straight from the caller's memory (no queue): the same bytes."""
    import io
    from rust_snappy_amd import frame
    data = (O.CORPUS / "lcet10.txt").read_bytes() * 12      # 5 MB
    sink = io.BytesIO()
    enc = frame.FrameEncoder(sink, fake)
    enc.write_all(b"head")
    enc.flush()
    enc.write_all(memoryview(data))
    enc.write_all(b"tail")
    enc.into_inner()
    want = (O.frame_compress(b"head") + O.frame_compress(data)[10:]
            + O.frame_compress(b"tail")[10:])
    assert sink.getvalue() == want
