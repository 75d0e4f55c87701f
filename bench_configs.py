#!/usr/bin/env python3
"""Secondary benchmarks: BASELINE.json configs 3 and 5 (bench.py is the
headline config 2), the per-file rates of the reference's bench list and the
PCIe-inclusive rate of the host-buffer entry points.  One JSON line per
config on stdout; --only picks one.

  cfg3  FrameEncoder/FrameDecoder (framing + CRC32C kernel) on seeded
        synthetic English-like text, device resident.  Text: tokens of the
        four corpus text files drawn i.i.d. from their unigram distribution
        (torch.multinomial, seed 0x5EED5A4D), single spaces, a newline where a
        token crosses a 73-byte boundary; a period of --period-mib is generated
        on the device and tiled to --gib.
  cfg5  incompressible path: fireworks.jpeg as independent raw streams tiled
        to --gib (literal fast path; the one workload near the HBM roofline).
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

GIB = float(1 << 30)


SYNTH_SEED = 0x5EED5A4D53505931
_GAMMA = 0x9E3779B97F4A7C15
_M1, _M2 = 0xBF58476D1CE4E5B9, 0x94D049BB133111EB


def _s64(x):
    """a 64-bit pattern as the int64 torch computes with (wrapping)"""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x


def _vocabulary():
    """Whitespace-delimited tokens of the four text files of the corpus, in
    byte order, with their counts (the empirical unigram distribution)."""
    import oracle_lib as O
    words = []
    for name in ("alice29.txt", "asyoulik.txt", "lcet10.txt", "plrabn12.txt"):
        words += (O.CORPUS / name).read_bytes().split()
    vocab, counts = np.unique(np.array(words, dtype=object),
                              return_counts=True)
    return [bytes(w) for w in vocab], counts.astype(np.int64)


def synth_text_reference(period_bytes, seed=SYNTH_SEED):
    """The generator of SURVEY 8d, one token at a time on the CPU (small
    periods only: it pins synth_text below).  Token i is drawn with the i-th
    output of splitmix64(seed): r = (z >> 1) mod total count, looked up in
    the cumulative counts of the vocabulary; tokens are joined by single
    spaces; the separator behind the first token that passes column 72 is a
    newline."""
    vocab, counts = _vocabulary()
    cum = np.cumsum(counts)
    total = int(cum[-1])
    out = bytearray()
    state, col = seed, 0
    mask = (1 << 64) - 1
    while len(out) < period_bytes:
        state = (state + _GAMMA) & mask
        z = state
        z = ((z ^ (z >> 30)) * _M1) & mask
        z = ((z ^ (z >> 27)) * _M2) & mask
        z ^= z >> 31
        w = vocab[int(np.searchsorted(cum, (z >> 1) % total, side="right"))]
        out += w
        col += len(w)
        if col > 72:
            out += b"\n"
            col = 0
        else:
            out += b" "
            col += 1
    return bytes(out[:period_bytes])


def synth_text(dev, period_bytes, seed=SYNTH_SEED):
    """synth_text_reference on the device (a period is 10^7..10^8 tokens):
    counter-based splitmix64 in wrapping int64 arithmetic, a search in the
    cumulative counts, byte positions by a prefix sum - and the greedy line
    breaks, which are a chain (line k starts where line k-1 broke): "the line
    that starts at token i ends behind token next(i)" for every i, then the
    orbit of token 0 by pointer doubling."""
    vocab, counts = _vocabulary()
    maxlen = max(len(w) for w in vocab)
    table = np.zeros((len(vocab), maxlen), dtype=np.uint8)
    lens = np.array([len(w) for w in vocab], dtype=np.int64)
    for i, w in enumerate(vocab):
        table[i, :len(w)] = np.frombuffer(w, dtype=np.uint8)
    cum = torch.from_numpy(np.cumsum(counts)).to(dev)
    total = int(cum[-1].item())
    mean_len = float((lens * counts).sum() / counts.sum()) + 1.0
    ntok = int(period_bytes / mean_len * 1.02) + 64
    i = torch.arange(1, ntok + 1, dtype=torch.int64, device=dev)
    z = i * _s64(_GAMMA) + _s64(seed)                   # wraps mod 2^64

    def lsr(x, k):                                      # logical shift right
        return (x >> k) & ((1 << (64 - k)) - 1)
    z = (z ^ lsr(z, 30)) * _s64(_M1)
    z = (z ^ lsr(z, 27)) * _s64(_M2)
    z = z ^ lsr(z, 31)
    ids = torch.searchsorted(cum, lsr(z, 1) % total, right=True)
    del z, i
    d_len = torch.from_numpy(lens).to(dev)[ids]          # token lengths
    ends = torch.cumsum(d_len + 1, 0)                    # behind separator
    starts = ends - d_len - 1
    total_bytes = int(ends[-1].item())
    assert total_bytes >= period_bytes
    # a line that starts at token i (byte starts[i]) breaks behind the first
    # token j whose last byte passes column 72: ends[j] - 1 - starts[i] > 72
    brk = torch.searchsorted(ends, starts + 73, right=True)   # that token j
    nxt = torch.clamp(brk + 1, max=ntok - 1)             # next line's first
    is_start = torch.zeros(ntok, dtype=torch.bool, device=dev)
    is_start[0] = True
    jump = nxt
    while True:                                          # pointer doubling
        idx = is_start.nonzero().squeeze(1)
        before = idx.numel()
        is_start[jump[idx]] = True
        if int(is_start.sum().item()) == before:
            break
        jump = jump[jump]
    del jump
    is_start[-1] = True   # (the clamp's fixed point; behind the period)
    line_first = is_start.nonzero().squeeze(1)
    last_tok = brk[line_first]                           # gets the newline
    last_tok = last_tok[last_tok < ntok]
    out = torch.full((total_bytes,), 32, dtype=torch.uint8, device=dev)
    d_table = torch.from_numpy(table).to(dev)
    for k in range(maxlen):                              # k-th character
        m = d_len > k
        out[starts[m] + k] = d_table[ids[m], k]
    out[ends[last_tok] - 1] = 10
    return out[:period_bytes].contiguous()


def time_it(fn, steps, ctx):
    fn()
    ctx.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    ctx.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def cfg3(args, ctx, dev):
    from rust_snappy_amd import frame
    import oracle_lib as O
    pbytes = min(int(args.period_mib * (1 << 20)),
                 max(1 << 20, int(args.gib * GIB) // 65536 * 65536))
    period = synth_text(dev, pbytes)
    reps = max(1, int(args.gib * GIB / period.numel()))
    data = period.repeat(reps)
    n = data.numel()
    out, flen, index = frame.compress_device(ctx, data, want_index=True)
    # parity: the first 64 MiB (1024 chunks) against the oracle's restatement
    # of write::FrameEncoder, chunk for chunk; the timed decode below checks
    # every period of the round trip against the generator's period
    hn = min(64 << 20, n // 65536 * 65536)
    head = data[:hn].cpu().numpy().tobytes()
    want = O.frame_compress(head)
    k = hn // 65536
    cut = int(index[k].item())
    assert out[:cut].cpu().numpy().tobytes() == want[:cut], "framed bytes differ"
    import ctypes as C
    from rust_snappy_amd import _lib, raw
    L = _lib.load()
    out_len = torch.zeros(1, dtype=torch.int64, device=dev)
    err = torch.zeros(32, dtype=torch.uint8, device=dev)
    cap = frame.frame_max_len(n)

    def enc():
        rc = L.snapmi_frame_compress(ctx._h, C.c_void_p(data.data_ptr()), n,
                                     C.c_void_p(out.data_ptr()), cap,
                                     C.c_void_p(out_len.data_ptr()),
                                     C.c_void_p(index.data_ptr()))
        assert rc == 0

    te = time_it(enc, args.steps, ctx)
    # the input is not needed any more (64 GiB + 75 GiB of output capacity +
    # 64 GiB of decoded output would not leave room for the scratch)
    del data
    torch.cuda.empty_cache()
    back, m = frame.decompress_device(ctx, out, flen, index=index, out_cap=n)
    assert m == n, "frame round trip length"

    def dec():
        rc = L.snapmi_frame_decompress(ctx._h, C.c_void_p(out.data_ptr()),
                                       flen, C.c_void_p(back.data_ptr()), n,
                                       C.c_void_p(out_len.data_ptr()),
                                       C.c_void_p(err.data_ptr()),
                                       C.c_void_p(index.data_ptr()),
                                       index.numel() - 1)
        assert rc == 0

    def dec_walk():  # no side index: the device walks the chunk headers
        rc = L.snapmi_frame_decompress(ctx._h, C.c_void_p(out.data_ptr()),
                                       flen, C.c_void_p(back.data_ptr()), n,
                                       C.c_void_p(out_len.data_ptr()),
                                       C.c_void_p(err.data_ptr()), None, 0)
        assert rc == 0

    tw = time_it(dec_walk, 1, ctx)
    td = time_it(dec, args.steps, ctx)
    # round trip of the timed decode against the generator's period
    per = back[:reps * period.numel()].view(reps, period.numel())
    for r in range(reps):
        assert torch.equal(per[r], period), "frame round trip (timed)"
    types = out[index[:-1]]                  # chunk type bytes
    n_stored = int((types == 1).sum().item())
    return {"config": "cfg3 framed synthetic text (SURVEY 8d generator: "
                      "splitmix64 seed 0x5EED5A4D53505931, unigram tokens of "
                      "the four corpus texts, newline behind the first token "
                      "past column 72; period "
                      f"{period.numel() >> 20} MiB)",
            "gib": round(n / GIB, 3),
            "chunks": index.numel() - 1, "uncompressed_chunks": n_stored,
            "ratio": round(flen / n, 4),
            "frame_encode_gibs": round(n / GIB / te, 2),
            "frame_decode_gibs": round(n / GIB / td, 2),
            "encode_ms": round(te * 1e3, 2), "decode_ms": round(td * 1e3, 2),
            "decode_uses_side_index": True,
            "frame_decode_no_index_gibs": round(n / GIB / tw, 2),
            "decode_no_index_ms": round(tw * 1e3, 2)}


def stream_probe(t, write=True):
    """GB/s of a plain streaming pass over the uint8 tensor `t` - a fill
    (write) or a sum (read) by torch's own kernels: where in the device's
    memory a buffer lies decides 20 % of what streams into it (tests/hw/
    zone_stream.hip: writes run at 4.7 TB/s into the first 64 GiB a process is
    handed and at 5.2-6.3 TB/s behind them, reads at 5.7 / 6.0-6.3)."""
    v = t[:t.numel() // 8 * 8].view(torch.int64)
    for k in range(3):
        if k == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if write:
            v.fill_(0)
        else:
            v.sum()
    torch.cuda.synchronize()
    return round(2 * v.numel() * 8 / (time.perf_counter() - t0) / 1e9)


def raw_tiles(ctx, dev, blob, gib, steps, want=None, diag=None,
              back_first=False):
    """`blob` as independent raw streams tiled to `gib`: compress and
    decompress rates, first/last stream checked against `want`.  `diag`: a
    dict for the streaming rates of the buffers where they lie."""
    from rust_snappy_amd import batch, raw
    reps = max(1, int(gib * GIB / max(1, len(blob))))
    stride = (len(blob) + 15) // 16 * 16
    back = None
    if back_first:   # the decoder's output buffer before everything else
        back = batch.StreamBatch.empty(
            np.full(reps, len(blob), dtype=np.int64), dev)
    one = np.zeros(stride, dtype=np.uint8)
    one[:len(blob)] = np.frombuffer(blob, dtype=np.uint8)
    data = torch.from_numpy(one).to(dev).repeat(reps)
    offs = np.arange(reps, dtype=np.int64) * stride
    lens = np.full(reps, len(blob), dtype=np.int64)
    src = batch.StreamBatch(data, offs, lens)
    cap = raw.max_compress_len(len(blob))
    comp = batch.StreamBatch.empty(np.full(reps, cap, dtype=np.int64), dev)
    clens = torch.zeros(reps, dtype=torch.int64, device=dev)
    if back is None:
        back = batch.StreamBatch.empty(lens, dev)
    blens = torch.zeros(reps, dtype=torch.int64, device=dev)

    def enc():
        raw.compress_batch(ctx, src.d_ptrs, src.d_lens, comp.d_ptrs,
                           comp.d_lens, clens, None, host_in_lens=src.h_lens)

    def dec():
        raw.decompress_batch(ctx, comp.d_ptrs, clens, back.d_ptrs,
                             back.d_lens, blens, None)

    if diag is not None:
        diag.update({"write_gbs_into_compressed": stream_probe(comp.data),
                     "write_gbs_into_decoded": stream_probe(back.data),
                     "read_gbs_of_input": stream_probe(data, write=False)})
    enc()
    ctx.synchronize()
    if want is not None:
        assert comp.stream_bytes(0, int(clens[0])) == want
        assert comp.stream_bytes(reps - 1, int(clens[-1])) == want
    te = time_it(enc, steps, ctx)
    td = time_it(dec, steps, ctx)
    assert back.stream_bytes(reps // 2) == blob
    n = reps * len(blob)
    c = int(clens.sum().item())
    return n, c, reps, te, td


def cfg5(args, ctx, dev):
    import oracle_lib as O
    jpg = (O.CORPUS / "fireworks.jpeg").read_bytes()

    def row(n, c, reps, te, td, diag):
        return {"gib": round(n / GIB, 3), "streams": reps,
                "compress_gibs": round(n / GIB / te, 2),
                "decompress_gibs": round(n / GIB / td, 2),
                "compress_hbm_frac": round((n + c) / te / 8e12, 4),
                "decompress_hbm_frac": round((n + c) / td / 8e12, 4),
                "compress_ms": round(te * 1e3, 2),
                "decompress_ms": round(td * 1e3, 2), **diag}
    diag = {"free_gib_before": round(torch.cuda.mem_get_info(dev)[0] / GIB)}
    res = {"config": "cfg5 incompressible (fireworks.jpeg tiles)"}
    res.update(row(*raw_tiles(ctx, dev, jpg, args.gib, args.steps,
                              O.compress(jpg), diag), diag))
    # Round 5 left a riddle: this pure streaming path took 11.8 ms to decode
    # on one box and 14.0 on the next.  It is WHERE the buffers lie (see
    # stream_probe): the same call with every buffer allocated behind a
    # spacer of 96 GiB, and with the decoder's output buffer allocated first
    # (the part of the memory that takes writes slowest) - beside the row above, whose buffers lie wherever the allocator
    # put them after the configs that ran before.
    torch.cuda.empty_cache()
    d2 = {}
    res["decoded_buffer_allocated_first"] = row(
        *raw_tiles(ctx, dev, jpg, args.gib, args.steps, O.compress(jpg), d2,
                   back_first=True), d2)
    torch.cuda.empty_cache()
    need = 3.3 * args.gib + 96 + 8
    if torch.cuda.mem_get_info(dev)[0] / GIB > need:
        spacer = torch.empty(int(96 * GIB), dtype=torch.uint8, device=dev)
        d3 = {}
        res["behind_a_96_gib_spacer"] = row(
            *raw_tiles(ctx, dev, jpg, args.gib, args.steps, O.compress(jpg),
                       d3), d3)
        del spacer
        torch.cuda.empty_cache()
    return res


def files(args, ctx, dev):
    """The 12 inputs of the reference's bench list one at a time
    (bench/src/bench.rs:83-114; README.md:135-158 quotes them per file),
    each tiled to --gib as independent raw streams."""
    import oracle_lib as O
    rows = {}
    for bench_id, blob in O.corpus_round():
        n, c, reps, te, td = raw_tiles(ctx, dev, blob, args.gib, args.steps,
                                       O.compress(blob))
        rows[bench_id] = {"bytes": len(blob), "ratio": round(c / n, 4),
                          "streams": reps,
                          "compress_gibs": round(n / GIB / te, 2),
                          "decompress_gibs": round(n / GIB / td, 2)}
    return {"config": f"per-file rates, each tiled to {args.gib} GiB",
            "files": rows}


def round_tiles(ctx, dev, gib, steps, diag=None):
    """The 12-stream zflat/uflat round tiled to `gib` as independent raw
    streams (bench.py's workload at another size): compress and decompress
    seconds per pass, round 0 compared with the oracle's bytes, the round
    trip with the input.  `diag` (a dict) gets what explains a row from the
    outside: the first compress call on this context at this size on its own
    (it allocates, and places the lane tables), the placement's log, every
    timed call's kernel ms, hipMemGetInfo's free bytes around it all."""
    import oracle_lib as O
    from rust_snappy_amd import batch, raw
    rnd = O.corpus_round()
    offs, pos = [], 0
    for _, d in rnd:
        offs.append(pos)
        pos += (len(d) + 15) // 16 * 16
    one = np.zeros(pos, dtype=np.uint8)
    for (_, d), o in zip(rnd, offs):
        one[o:o + len(d)] = np.frombuffer(d, dtype=np.uint8)
    r_lens = np.array([len(d) for _, d in rnd], dtype=np.int64)
    rounds = max(1, int(round(gib * GIB / int(r_lens.sum()))))
    data = torch.from_numpy(one).to(dev).repeat(rounds)
    o_all = (np.arange(rounds, dtype=np.int64)[:, None] * pos
             + np.array(offs, dtype=np.int64)[None, :]).reshape(-1)
    lens = np.tile(r_lens, rounds)
    n = 12 * rounds
    src = batch.StreamBatch(data, o_all, lens)
    caps = np.array([raw.max_compress_len(int(x)) for x in r_lens],
                    dtype=np.int64)
    comp = batch.StreamBatch.empty(np.tile(caps, rounds), dev)
    clens = torch.zeros(n, dtype=torch.int64, device=dev)
    back = batch.StreamBatch.empty(lens, dev)
    blens = torch.zeros(n, dtype=torch.int64, device=dev)

    def enc():
        raw.compress_batch(ctx, src.d_ptrs, src.d_lens, comp.d_ptrs,
                           comp.d_lens, clens, None, host_in_lens=src.h_lens)

    def dec():
        raw.decompress_batch(ctx, comp.d_ptrs, clens, back.d_ptrs,
                             back.d_lens, blens, None)

    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    t0 = time.perf_counter()
    enc()
    ctx.synchronize()
    first_ms = (time.perf_counter() - t0) * 1e3
    cl = clens.cpu().numpy()
    for j, (_, d) in enumerate(rnd):
        assert comp.stream_bytes(j, int(cl[j])) == O.compress(d), j
        k = n - 12 + j
        assert comp.stream_bytes(k, int(cl[k])) == O.compress(d), k
    calls = []

    def enc_logged():
        t0 = time.perf_counter()
        enc()
        ctx.synchronize()
        calls.append(round((time.perf_counter() - t0) * 1e3, 2))
    te = time_it(enc_logged if diag is not None else enc, steps, ctx)
    enc_kernel = ctx.last_kernel()
    td = time_it(dec, steps, ctx)
    if diag is not None:
        diag.update({
            "first_call_ms": round(first_ms, 1),
            "call_ms": calls,            # the warm call, then the timed ones
            "kernel": enc_kernel,        # the compress side's dominant kernel
            "placement": ctx.table_probe_log(),
            "free_gib_before": round(free0 / GIB, 1),
            "free_gib_after": round(torch.cuda.mem_get_info(dev)[0] / GIB, 1)})
    for j, (_, d) in enumerate(rnd):
        assert back.stream_bytes(n - 12 + j) == d, j
    ub = rounds * int(r_lens.sum())
    return ub, int(cl.sum()), n, te, td


def sweep(args, ctx, dev):
    """The batch-size regime the headline hides: bench.py's workload (the
    12-stream round as independent raw streams) at 64 MiB / 256 MiB / 1 GiB /
    4 GiB per call - what an adapter's batch, a host slice or a frame of a
    few hundred MiB pays."""
    from rust_snappy_amd import raw
    rows = {}
    # (a context of its own: its lane tables are made for the first size
    # that needs them, which is not how the other configs meet theirs)
    own = raw.Context(ctx.device)
    for label, gib in (("64MiB", 1 / 16), ("256MiB", 0.25), ("1GiB", 1.0),
                       ("4GiB", 4.0)):
        if gib > args.gib:
            continue
        diag = {}
        ub, cb, n, te, td = round_tiles(own, dev, gib, max(args.steps, 3),
                                        diag)
        rows[label] = {"gib": round(ub / GIB, 4), "streams": n,
                       "compress_gibs": round(ub / GIB / te, 2),
                       "decompress_gibs": round(ub / GIB / td, 2),
                       "compress_ms": round(te * 1e3, 3),
                       "decompress_ms": round(td * 1e3, 3), **diag}
    own.close()
    torch.cuda.empty_cache()
    return {"config": "bench.py's workload (12-stream round, independent raw "
                      "streams, device resident) by batch size", "sizes": rows}


def budget(args, ctx, dev):
    """What the headline's memory setting buys: bench.py's workload at --gib
    with the lane tables allowed 75 % (bench.py), 33 % (the library's
    default) and 15 % of the free device memory - compress GiB/s, the bytes
    the tables hold afterwards and the most that was held while a placement
    was chosen (snapmi_table_probe_log)."""
    from rust_snappy_amd import _lib, raw
    rows = {}
    for pct in (75, 33, 15):
        c = raw.Context(ctx.device)
        c.set_option("lane_table_budget_pct", pct)
        free0 = torch.cuda.mem_get_info(dev)[0]
        diag = {}
        ub, cb, n, te, td = round_tiles(c, dev, args.gib, max(args.steps, 3),
                                        diag)
        log = c.table_probe_log()
        held = None
        if "held at most" in log:
            held = int(log.split("held at most")[1].split()[0])
        torch.cuda.empty_cache()
        free1 = torch.cuda.mem_get_info(dev)[0]
        rows[f"pct{pct}"] = {"compress_gibs": round(ub / GIB / te, 2),
                             "compress_ms": round(te * 1e3, 2),
                             "context_bytes": int(free0 - free1),
                             "token_scratch_bytes":
                                 c.info("token_scratch_bytes"),
                             "input_bytes": int(ub),
                             "held_at_most_during_placement": held,
                             "first_call_ms": diag["first_call_ms"],
                             "call_ms": diag["call_ms"],
                             "placement": diag["placement"]}
        c.close()
        torch.cuda.empty_cache()
    # ... and the explicit opt-in: snapmi_ctx_prepare(TOP_OF_MEMORY) before
    # the first batch (holds the whole device for a moment, include/snapmi.h).
    # NOTE what this row is: the call is made for a process that asks FIRST;
    # here it is the fourth context of a process that has allocated and freed
    # hundreds of GB, where "the far end" is wherever the allocator's free
    # lists say (the device reuses what was freed last): the row shows the
    # call's cost and that it works, not its best case
    # (profiles/r6_budget_chunks.txt: 71.6-71.8 GiB/s in a fresh process)
    c = raw.Context(ctx.device)
    rounds = max(1, int(round(args.gib * GIB / 2928571)))
    free0 = torch.cuda.mem_get_info(dev)[0]
    t0 = time.perf_counter()
    c.prepare(50 * rounds, top_of_memory=True)     # 50 blocks per round
    prep_ms = (time.perf_counter() - t0) * 1e3
    diag = {}
    ub, cb, n, te, td = round_tiles(c, dev, args.gib, max(args.steps, 3), diag)
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info(dev)[0]
    rows["prepared_top_of_memory"] = {
        "compress_gibs": round(ub / GIB / te, 2),
        "compress_ms": round(te * 1e3, 2),
        "context_bytes": int(free0 - free1),
        "prepare_ms": round(prep_ms, 1),
        "first_call_ms": diag["first_call_ms"], "call_ms": diag["call_ms"],
        "placement": diag["placement"]}
    c.close()
    torch.cuda.empty_cache()
    # ... and the token scratch: the pool at 100 % of the worst case (no
    # block can spill; round 5's layout in pages), and the block list matched
    # and encoded in two equal launches (lane_segment_blocks) - what the
    # context holds against what it costs
    for seg in (1 << 20, 98304):
        c = raw.Context(ctx.device)
        c.set_option("lane_table_budget_pct", 75)
        c.set_option("lane_segment_blocks", seg)
        if seg == 1 << 20:
            c.set_option("token_pool_pct", 100)
        free0 = torch.cuda.mem_get_info(dev)[0]
        diag = {}
        ub, cb, n, te, td = round_tiles(c, dev, args.gib, max(args.steps, 3),
                                        diag)
        torch.cuda.empty_cache()
        free1 = torch.cuda.mem_get_info(dev)[0]
        launches = -(-50 * rounds // seg)
        rows["token_pool_100_pct" if seg == 1 << 20 else
             f"segments_of_at_most_{seg}_blocks"] = {
            "launches": launches,
            "compress_gibs": round(ub / GIB / te, 2),
            "compress_ms": round(te * 1e3, 2),
            "context_bytes": int(free0 - free1),
            "token_scratch_bytes": c.info("token_scratch_bytes"),
            "blocks_spilled": c.info("token_blocks_spilled"),
            "input_bytes": int(ub), "call_ms": diag["call_ms"]}
        c.close()
        torch.cuda.empty_cache()
    return {"config": f"lane_table_budget_pct 75 / 33 / 15 at {args.gib:g} "
                      "GiB of bench.py's workload, snapmi_ctx_prepare("
                      "SNAPMI_PREPARE_TOP_OF_MEMORY), token_pool_pct 100, "
                      "and the batch in two launches (lane_segment_blocks)",
            "budgets": rows}


def tiny(args, ctx, dev):
    """The tiny-stream regime on its own: zflat03 (the first 200 bytes of
    fireworks.jpeg, bench/src/bench.rs:91) tiled to --gib = 10.7 M streams
    at 2 GiB; beside it the first 200 bytes of alice29.txt (tiny streams
    that do compress: literals and copies in every lane) and its first
    400 / 1 000 / 2 000 / 4 096 bytes."""
    import oracle_lib as O
    res = None
    for bench_id, blob in O.corpus_round():
        if len(blob) == 200:
            n, c, reps, te, td = raw_tiles(ctx, dev, blob, args.gib,
                                           args.steps, O.compress(blob))
            res = {"config": f"{bench_id} tiled to {args.gib} GiB",
                   "streams": reps, "ratio": round(c / n, 4),
                   "compress_gibs": round(n / GIB / te, 2),
                   "decompress_gibs": round(n / GIB / td, 2),
                   "compress_ms": round(te * 1e3, 2),
                   "decompress_ms": round(td * 1e3, 2)}
    # (400 .. 2 000 bytes: the three classes of k_compress_small; 4 KiB: a
    # one-block stream of the block kernels)
    for key, size in (("text_200", 200), ("text_400", 400),
                      ("text_1k", 1000), ("text_2k", 2000),
                      ("text_4k", 4096)):
        text = (O.CORPUS / "alice29.txt").read_bytes()[:size]
        n, c, reps, te, td = raw_tiles(ctx, dev, text, args.gib, args.steps,
                                       O.compress(text))
        res[key] = {"ratio": round(c / n, 4),
                    "compress_gibs": round(n / GIB / te, 2),
                    "decompress_gibs": round(n / GIB / td, 2)}
    return res


def seam(args, ctx, dev):
    """The reference-side view of the native seam (snappy-cpp/src/lib.rs:13-88,
    the `cpp` group of bench/src/bench.rs:117-153): tests/seam_consumer.c - a C
    program against snappy-c.h, linked with -lsnappy - built once against a
    directory whose libsnappy.so is a symlink to libsnapmi.so and once against
    Google's libsnappy 1.1.8, one call per file, 1 and 16 callers at once.
    MB/s of uncompressed bytes, [compress, uncompress]; args.gib = milliseconds
    per file, direction and thread count."""
    import oracle_lib as O
    import os
    import subprocess
    import tempfile
    hdr = "/opt/conda/include"
    if not os.path.exists(f"{hdr}/snappy-c.h"):
        return {"error": "no snappy-c.h in this image"}
    ms = max(20.0, float(args.gib))
    with tempfile.TemporaryDirectory() as d:
        lib = ROOT / "rust-snappy_amd" / "libsnapmi.so"
        os.symlink(lib, f"{d}/libsnappy.so")
        ins = f"{d}/in"
        os.mkdir(ins)
        for name, data in O.corpus_round():
            open(f"{ins}/{name}.in", "wb").write(data)
            open(f"{ins}/{name}.snappy", "wb").write(O.compress(data))
        src = str(ROOT / "tests" / "seam_consumer.c")
        builds = {"snapmi": [f"-L{d}", f"-Wl,-rpath,{d}",
                             f"-Wl,-rpath,{lib.parent}"]}
        if os.path.exists("/opt/conda/lib/libsnappy.so"):
            builds["libsnappy_1_1_8"] = ["-L/opt/conda/lib",
                                         "-Wl,-rpath,/opt/conda/lib"]
        res = {"config": "one call per bench input through snappy-c.h "
                         "(tests/seam_consumer.c, -lsnappy)",
               "unit": "MB/s uncompressed [compress, uncompress]",
               "ms_per_leg": ms}
        for key, flags in builds.items():
            exe = f"{d}/consumer_{key}"
            subprocess.check_call(["gcc", "-O2", "-I", hdr, "-o", exe, src]
                                  + flags + ["-lsnappy", "-lpthread"])
            p = subprocess.run([exe, "check", ins], capture_output=True,
                               text=True, timeout=120)
            if p.returncode != 0:
                return {"error": f"{key}: check failed: "
                                 + (p.stdout + p.stderr)[-200:]}
            rows = {}
            for threads in (1, 16):
                p = subprocess.run([exe, "bench", ins, str(threads), str(ms)],
                                   capture_output=True, text=True,
                                   timeout=300)
                if p.returncode != 0:
                    return {"error": f"{key}: bench with {threads} callers: "
                                     + (p.stdout + p.stderr)[-200:]}
                for line in p.stdout.splitlines():
                    name, n, t, c, u = line.split()
                    rows.setdefault(name, {})[f"callers_{t}"] = [
                        float(c), float(u)]
            res[key] = rows
        return res


def _host_corpus(gib):
    """The corpus round tiled into pinned host memory (snapmi_host_alloc)."""
    import oracle_lib as O
    from rust_snappy_amd import frame
    blob = b"".join(d for _, d in O.corpus_round())
    reps = max(1, int(gib * GIB / len(blob)))
    n = reps * len(blob)
    h_in = frame.HostBuffer(n)
    h_in.array.reshape(reps, len(blob))[:] = np.frombuffer(blob, dtype=np.uint8)
    return blob, n, h_in


def pcie(args, ctx, dev):
    """Host to host through the frame layer: snapmi_frame_encode_host and
    snapmi_frame_decode_host on pinned host buffers - what a host-side
    FrameEncoder / FrameDecoder pays per batch (slices in flight on three
    streams: copy in, kernels, copy out).  bench.py's `value` is device
    resident; this is never it."""
    import ctypes as C
    import oracle_lib as O
    from rust_snappy_amd import _lib, frame
    L = _lib.load()
    blob, n, h_in = _host_corpus(min(args.gib, 4.0))
    nch = (n + 65535) // 65536
    lens = np.full(nch, 65536, dtype=np.uint32)
    lens[-1] = n - (nch - 1) * 65536
    h_out = frame.HostBuffer(10 + n + 8 * nch)
    h_back = frame.HostBuffer(nch * 65536)
    flen = 0

    def enc():
        nonlocal flen
        flen = frame.encode_host_into(ctx, h_in.view, lens, h_out)

    def dec():
        stale = (C.c_uint8 * 10)()
        written, consumed = C.c_size_t(0), C.c_size_t(0)
        err = _lib.SnapmiError()
        rc = L.snapmi_frame_decode_host(
            ctx._h, C.c_void_p(h_out.ptr), flen, 2, stale,
            C.c_void_p(h_back.ptr), h_back.nbytes, C.byref(written),
            C.byref(consumed), C.byref(err))
        assert rc == 0 and written.value == n and consumed.value == flen, \
            (rc, written.value, consumed.value)

    te = time_it(enc, args.steps, ctx)
    td = time_it(dec, args.steps, ctx)
    assert bool((h_back.array[:n] == h_in.array).all()), "host round trip"
    head = O.frame_compress(bytes(h_in.view[:1 << 20]))
    assert bytes(h_out.view[:len(head)]) == head, "framed bytes vs the oracle"
    res = {"config": "host to host through snapmi_frame_encode_host / "
                     "_decode_host (pinned memory; slices pipelined over "
                     "copy-in, kernels, copy-out)",
           "gib": round(n / GIB, 3), "ratio": round(flen / n, 4),
           "frame_encode_gibs": round(n / GIB / te, 2),
           "frame_decode_gibs": round(n / GIB / td, 2),
           "encode_ms": round(te * 1e3, 2), "decode_ms": round(td * 1e3, 2)}
    for b in (h_in, h_out, h_back):
        b.close()
    return res


def adapters(args, ctx, dev):
    """The streaming adapters the north star keeps: FrameEncoder.write_all of
    one large host buffer (rust-snappy_amd/frame.py, the mirror of
    snap::write::FrameEncoder), FrameDecoder.read_to_end of the result (Python
    bytes) and FrameDecoder.readinto of it (the caller's pinned buffer)."""
    import io
    import oracle_lib as O
    from rust_snappy_amd import frame
    blob, n, h_in = _host_corpus(min(args.gib, 4.0))

    class Sink:
        def __init__(self):
            self.n, self.head = 0, bytearray()

        def write(self, b):
            if len(self.head) < (2 << 20):
                self.head += b[:2 << 20]
            self.n += len(b)
            return len(b)

    # one encoder, written to repeatedly (its pinned staging buffer is made
    # by the first large write, like a Vec that has grown)
    sink = Sink()
    enc = frame.FrameEncoder(sink, ctx)
    enc.write_all(h_in.view[:64 << 20])
    enc.flush()
    enc.DIRECT_MAX = 4 << 30    # one device call for the whole write
    enc.write_all(h_in.view)                    # staging grows here
    enc.flush()
    head0 = bytes(sink.head)
    best = None
    for _ in range(max(2, args.steps)):
        t0 = time.perf_counter()
        enc.write_all(h_in.view)
        enc.flush()
        t = time.perf_counter() - t0
        best = t if best is None or t < best else best
    sink = Sink()
    enc2 = frame.FrameEncoder(sink, ctx)
    enc2.write_all(h_in.view[:4 << 20])
    enc2.flush()
    want = O.frame_compress(bytes(h_in.view[:1 << 20]))
    assert bytes(sink.head[:len(want)]) == want, "adapter bytes vs the oracle"
    # decoder adapter on a smaller stream (it hands out Python bytes)
    m = min(n, 1 << 30)
    sink = io.BytesIO()
    enc = frame.FrameEncoder(sink, ctx)
    enc.write_all(h_in.view[:m])
    enc.flush()
    framed = sink.getvalue()
    # read_to_end: everything into one bytearray (the reference's Vec<u8>).
    # Its ceiling is the host's, not the device's: every byte of the result
    # is written once into memory the process has never touched - timed here
    # as a plain copy of the same size into a fresh bytearray
    t0 = time.perf_counter()
    fresh = bytearray(h_in.view[:m])
    t_fresh = time.perf_counter() - t0
    del fresh
    td = None
    for _ in range(2):
        dec = frame.FrameDecoder(io.BytesIO(framed), ctx)
        t0 = time.perf_counter()
        back = dec.read_to_end()
        t = time.perf_counter() - t0
        td = t if td is None or t < td else td
        assert len(back) == m and back[:1 << 20] == h_in.view[:1 << 20] and \
            back[m - (1 << 20):] == h_in.view[m - (1 << 20):m], \
            "adapter round trip"
        del back
        dec.close()

    def readinto_all(reader):
        """io::Read::read as the reference has it - into the caller's
        buffer (pinned, with room for a batch: the bytes come straight from
        the device call)"""
        dec = frame.FrameDecoder(reader, ctx)
        t0 = time.perf_counter()
        pos = 0
        while True:
            k = dec.readinto(h_back.view[pos:])
            if k == 0:
                break
            pos += k
        t = time.perf_counter() - t0
        assert pos == m and bytes(h_back.view[:1 << 20]) == bytes(
            h_in.view[:1 << 20]) and bytes(
            h_back.view[m - (1 << 20):m]) == bytes(
            h_in.view[m - (1 << 20):m]), "readinto round trip"
        dec.close()
        return t

    h_back = frame.HostBuffer(m + (256 << 20))
    # ... from Python bytes (io.BytesIO lends its buffer: no host copy, but
    # the copy to the device reads pageable memory) and from pinned memory
    # (HostReader: both ends pinned - what the device call itself does)
    ti = min(readinto_all(io.BytesIO(framed)) for _ in range(2))
    h_fr = frame.HostBuffer(len(framed))
    h_fr.view[:] = framed
    tp = min(readinto_all(frame.HostReader(h_fr.view)) for _ in range(2))
    h_fr.close()
    h_back.close()
    h_in.close()
    return {"config": "Python streaming adapters over the host-buffer calls: "
                      "FrameEncoder.write_all of one pinned buffer (no copy "
                      "on the Python side), FrameDecoder.read_to_end (one "
                      "bytearray), FrameDecoder.readinto a pinned buffer "
                      "from io.BytesIO and from a pinned HostReader",
            "gib": round(n / GIB, 3), "framed_bytes": sink_n(sink, framed),
            "frame_encoder_write_all_gibs": round(n / GIB / best, 2),
            "frame_decoder_read_to_end_gibs": round(m / GIB / td, 2),
            "fresh_bytearray_copy_gibs": round(m / GIB / t_fresh, 2),
            "frame_decoder_readinto_pinned_gibs": round(m / GIB / ti, 2),
            "frame_decoder_readinto_pinned_from_pinned_gibs": round(
                m / GIB / tp, 2),
            "decoder_gib": round(m / GIB, 3)}


def sink_n(sink, framed):
    return len(framed)


def stream(args, ctx, dev):
    """ONE raw stream of --gib (the 12 corpus files repeated): compressed as
    one Encoder::compress call (blocks in parallel), decoded by
    snapmi_decompress_stream (many wavefronts) and, for comparison, 1/16 of
    it as a batch of one stream: through snapmi_decompress_batch as it is
    (the long stream gets its pieces) and with batch_long_streams 0 (a
    single wavefront)."""
    import oracle_lib as O
    from rust_snappy_amd import batch, raw
    blob = b"".join(d for _, d in O.corpus_round())
    reps = max(1, int(min(args.gib, 3.0) * GIB / len(blob)))  # < 2^32 * 6/7
    one = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    data = one.repeat(reps)
    n = data.numel()
    src = batch.StreamBatch(data, np.array([0], dtype=np.int64),
                            np.array([n], dtype=np.int64))
    comp = batch.StreamBatch.empty(
        np.array([raw.max_compress_len(n)], dtype=np.int64), dev)
    clen = torch.zeros(1, dtype=torch.int64, device=dev)
    raw.compress_batch(ctx, src.d_ptrs, src.d_lens, comp.d_ptrs, comp.d_lens,
                       clen, None, host_in_lens=src.h_lens)
    ctx.synchronize()
    c = int(clen.item())
    back = torch.empty(n, dtype=torch.uint8, device=dev)
    out_len = torch.zeros(1, dtype=torch.int64, device=dev)
    err = torch.zeros(32, dtype=torch.uint8, device=dev)

    def dec():
        raw.decompress_stream(ctx, comp.data, c, back, out_len, err)

    td = time_it(dec, args.steps, ctx)
    assert raw.stream_decode_path(ctx) == 0
    assert int(out_len.item()) == n
    per = back.view(reps, len(blob))
    for r in range(0, reps, max(1, reps // 16)):
        assert torch.equal(per[r], one), "stream round trip"
    # a single wavefront on 1/16 of it
    m = (reps // 16 or 1) * len(blob)
    src1 = batch.StreamBatch(data, np.array([0], dtype=np.int64),
                             np.array([m], dtype=np.int64))
    raw.compress_batch(ctx, src1.d_ptrs, src1.d_lens, comp.d_ptrs,
                       comp.d_lens, clen, None, host_in_lens=src1.h_lens)
    ctx.synchronize()
    cap1 = torch.tensor([m], dtype=torch.int64, device=dev)
    optr = torch.tensor([back.data_ptr()], dtype=torch.int64, device=dev)

    def dec1():
        raw.decompress_batch(ctx, comp.d_ptrs, clen, optr, cap1, out_len, None)

    tb = time_it(dec1, 2, ctx)    # the batch call: the stream gets its pieces
    ctx.set_option("batch_long_streams", 0)
    try:
        t1 = time_it(dec1, 1, ctx)   # ... and a wavefront to itself
    finally:
        ctx.set_option("batch_long_streams", 1)
    return {"config": "one raw stream, device resident",
            "gib": round(n / GIB, 3), "ratio": round(c / n, 4),
            "decompress_stream_gibs": round(n / GIB / td, 2),
            "decompress_stream_ms": round(td * 1e3, 2),
            "batch_of_one_gibs": round(m / GIB / tb, 2),
            "one_wavefront_gibs": round(m / GIB / t1, 3),
            "one_wavefront_gib": round(m / GIB, 3)}


def cfg4(args, ctx, dev):
    """BASELINE config 4: the framed text stream of cfg3 sharded over the
    ranks of a torch.distributed job (launch with torch.distributed.run;
    world 1 works too).  Rank r frames periods [r*P/N, (r+1)*P/N) of the
    stream - a contiguous range of 64 KiB chunks, no exchange during compute -
    and the framed parts are gathered on rank 0 (shard.gatherv: sizes first,
    then grouped point-to-point into the root's buffer).  Parts after the
    first drop their 10-byte stream identifier, so the gathered bytes are the
    single-stream framing."""
    import os
    import torch.distributed as dist
    from rust_snappy_amd import frame, shard
    import oracle_lib as O
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("SNAPMI_OVERSUBSCRIBE") == "1":
            dist.init_process_group("gloo")  # N ranks on one GPU: proof run
        else:
            dist.init_process_group("nccl", device_id=dev)
    pbytes = min(int(args.period_mib * (1 << 20)),
                 max(1 << 20, int(args.gib * GIB / world) // 65536 * 65536))
    period = synth_text(dev, pbytes)
    assert period.numel() % 65536 == 0
    periods = max(world, int(args.gib * GIB / period.numel()))
    lo, hi = periods * rank // world, periods * (rank + 1) // world
    data = period.repeat(hi - lo)
    n = data.numel()

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    out, flen, _ = frame.compress_device(ctx, data)          # warm-up
    sync()
    t0 = time.perf_counter()
    out, flen, _ = frame.compress_device(ctx, data)
    sync()
    t_enc = time.perf_counter() - t0
    part = out[:flen] if rank == 0 else out[10:flen]
    del data
    if world > 1:
        # communicator and point-to-point channels are set up by the first
        # exchange: keep that out of the timed gather
        shard.gatherv(part[:1 << 20], dst=0)
        sync()
    t0 = time.perf_counter()
    whole = shard.gatherv(part, dst=0) if world > 1 else part
    sync()
    t_gather = time.perf_counter() - t0
    seen = [rank]
    if world > 1:
        seen = [None] * world
        dist.all_gather_object(seen, rank)
    res = None
    if rank == 0:
        # the head of the gathered stream against the oracle, chunk for chunk
        head = period[:1 << 20].cpu().numpy().tobytes()
        want = O.frame_compress(head)
        got = whole[:len(want)].cpu().numpy().tobytes()
        parity_ok = got == want  # (raised after the last barrier: a rank
        total = periods * period.numel()  # must not leave the others waiting)
        res = {"config": "cfg4 framed synthetic text sharded by chunk range",
               "n_gpus": world, "gib": round(total / GIB, 3),
               "framed_bytes": int(whole.numel()),
               "frame_encode_gibs_no_gather": round(total / GIB / t_enc, 2),
               "frame_encode_gibs_with_gather": round(
                   total / GIB / (t_enc + t_gather), 2),
               "encode_ms": round(t_enc * 1e3, 2),
               "gather_ms": round(t_gather * 1e3, 2),
               "gathered_bytes_from_peers": int(whole.numel() - part.numel()),
               # root ingress: 7 xGMI links x 153 GB/s
               "gather_gbs": (round((whole.numel() - part.numel()) / t_gather
                                    / 1e9, 1) if world > 1 else None),
               "gather_bound_gbs": 1071.0,
               "ranks_seen": sorted(seen)}
    if world > 1:
        dist.barrier()
    if rank == 0:
        assert parity_ok, "gathered framed bytes differ from the oracle's"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=8.0)
    ap.add_argument("--period-mib", type=float, default=1024.0,
                    help="period of the synthetic text (SURVEY 8d: 1 GiB)")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--only", default="")
    ap.add_argument("--plan", default="",
                    help="name:gib,... - run these configs at these sizes, "
                         "one JSON line each with a \"name\" key; a config "
                         "that fails prints {\"name\", \"error\"} and the "
                         "rest still run (bench.py's extras)")
    ap.add_argument("--option", action="append", default=[],
                    help="name=value: snapmi_ctx_set_option on the context "
                         "(experiments: --option tiny_stream_kernel=0)")
    args = ap.parse_args()
    import __graft_entry__ as g
    g.build()
    from rust_snappy_amd import raw
    import os
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SNAPMI_OVERSUBSCRIBE") == "1":
        local = 0  # bench.py --oversubscribe: every rank drives cuda:0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ctx = raw.Context(local)      # the library's defaults (extras.budget)
    for item in args.option:
        name, value = item.split("=")
        ctx.set_option(name, int(value))
    table = {"cfg3": cfg3, "cfg5": cfg5, "files": files, "pcie": pcie,
             "adapters": adapters, "stream": stream, "cfg4": cfg4,
             "tiny": tiny, "sweep": sweep, "budget": budget, "seam": seam}
    if args.plan:
        for item in args.plan.split(","):
            name, gib = item.split(":")
            args.gib = float(gib)
            t0 = time.perf_counter()
            try:
                res = table[name](args, ctx, dev)
            except Exception as e:  # noqa: BLE001 - reported, not hidden
                res = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
            if res is not None:
                res["name"] = name
                res["wall_s"] = round(time.perf_counter() - t0, 1)
                print(json.dumps(res), flush=True)
        return
    for name, fn in (("cfg3", cfg3), ("cfg5", cfg5), ("files", files),
                     ("pcie", pcie), ("adapters", adapters),
                     ("stream", stream), ("cfg4", cfg4)):
        if args.only != name and (args.only or name == "cfg4"):
            continue  # cfg4 only on request (it is the multi-rank config)
        res = fn(args, ctx, dev)
        if res is not None:
            print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
