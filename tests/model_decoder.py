"""Lane-level model (numpy, 64-wide arrays) of the windowed wave decoder in
rust-snappy_amd/csrc/snapmi_decompress.hip (k_decompress_streams, v2).

TEST INFRASTRUCTURE: the kernel's algorithm restated step by step so that its
logic - speculative decode per byte, element starts by mask doubling, the
2048-byte window cut, one lane-parallel copy step plus an in-order sweep of
the elements that depend on the window's own output, the 4 KiB LDS ring with
its "safe history" rule, 256-byte flushes, long literals, the hand-over to
the sequential decoder - can be checked on the CPU against the
oracle (tests/test_model_decoder_cpu.py).  It mirrors the kernel's variable
names; it is not used by the product.

decode(comp) -> ("ok", bytes) | ("irregular", s, d, prefix)
  "irregular": the wide path stopped at stream position s / output position
  d (a window boundary, or the tail of the stream) with `prefix` = out[:d]
  complete; the kernel then runs the sequential decoder from there, which
  owns every error report.
"""
import numpy as np

R = 4096          # ring bytes
WMAX = 2048       # output bytes per window at most
WAVE = 64

LANE = np.arange(WAVE)


def read_varint(b):
    v, shift = 0, 0
    for i, x in enumerate(b[:10]):
        if x < 0x80:
            return v | (x << shift), i + 1
        v |= (x & 0x7F) << shift
        shift += 7
    return None, 0


class Stats:
    def __init__(self):
        self.windows = self.rounds = self.far = self.wide = self.longlit = 0
        self.flush_partial = self.fence = self.elements = 0


def decode(comp, stats=None):
    st = stats or Stats()
    comp = bytes(comp)
    dst_len, hdr = read_varint(comp)
    assert hdr and dst_len is not None and dst_len < (1 << 32)
    src = np.frombuffer(comp[hdr:], dtype=np.uint8)
    src_len = len(src)
    out = np.zeros(dst_len + 64, dtype=np.uint8)      # "global memory"
    visible = 0   # out[:visible] is what completed stores have written
    ring = np.zeros(R + 16, dtype=np.uint8)           # + 16-byte mirror
    s = d = 0
    gflush = 0    # out[:gflush] has been stored (stores may be in flight)
    fenced = 0    # out[:fenced]: those stores are known complete
    ring_lo = 0   # ring holds out[max(ring_lo, d - R) : d]

    def ring_write(pos, data):
        for k, b in enumerate(data):
            ring[(pos + k) & (R - 1)] = b

    def mirror():
        ring[R:R + 16] = ring[0:16]

    def ring_read16(pos):
        a = pos & (R - 1)
        return ring[a:a + 16].copy()   # may run into the mirror, never past

    def flush_chunks():
        nonlocal gflush
        while gflush + 256 <= d:
            for k in range(256):
                out[gflush + k] = ring[(gflush + k) & (R - 1)]
            gflush += 256

    def flush_partial(upto):
        nonlocal gflush
        st.flush_partial += 1
        for p in range(gflush, upto):
            out[p] = ring[p & (R - 1)]
        gflush = upto

    def irregular():
        flush_partial(d)
        return ("irregular", s, d, bytes(out[:d]))

    if src_len < 8:
        return irregular()              # too short for the window loads
    while s < src_len:
        st.windows += 1
        rem = min(src_len - s, 1 << 20)
        # ---- speculative decode of "the element at src[s + lane]" ---------
        # (bytes behind the end of the input read as zero)
        w = np.array([int.from_bytes(src[s + i:s + i + 8].tobytes(), "little")
                      for i in range(WAVE)], dtype=np.uint64)
        tag = (w & np.uint64(0xFF)).astype(np.int64)
        b14 = ((w >> np.uint64(8)) & np.uint64(0xFFFFFFFF)).astype(np.int64)
        typ = tag & 3
        n6 = tag >> 2
        is_lit = typ == 0
        lnb = np.where(n6 >= 60, n6 - 59, 0)
        lmask = np.where(lnb == 4, 0xFFFFFFFF, (1 << (8 * lnb)) - 1)
        L = np.where(lnb > 0, (b14 & lmask) + 1, n6 + 1)
        hd = 1 + lnb
        long_ = is_lit & (L > 64)
        cnb = np.where(typ == 1, 1, np.where(typ == 2, 2, 4))
        clen = np.where(typ == 1, 4 + (n6 & 7), n6 + 1)
        off = np.where(typ == 1, ((tag >> 5) << 8) | (b14 & 0xFF),
                       np.where(typ == 2, b14 & 0xFFFF, b14))
        enc = np.where(is_lit, hd + L, 1 + cnb)
        olen = np.where(is_lit, L, clen)
        # the whole element lies inside the input; an extended literal
        # length is read as 4 bytes (src/decompress.rs:189-198)
        fits = (LANE < rem) & ~long_ & (enc <= rem - LANE) & \
            ~(is_lit & (lnb > 0) & (LANE + 5 > rem))
        # a window with 160 bytes of input in front of it needs none of the
        # per-lane tests (the kernel's uniform fast path)
        deep = rem >= 160
        if deep:
            assert (fits == ~long_).all()
        inner = rem >= 64 + 5 + 16      # speculative 16-byte literal loads
        nx = np.where(long_ | (LANE >= rem), WAVE,
                      np.minimum(LANE + enc, WAVE))
        # ---- element starts: S = orbit of lane 0 under nx ------------------
        # One dword per round (a ds_bpermute is what costs): the wave is two
        # halves of 32 positions; a lane's set R (32 bits, its own half)
        # holds the first nodes of its chain INCLUDING the frontier, which is
        # the set's highest bit because chains run forward.  One round:
        # R |= R[frontier].  After k rounds a set has 2^k + 1 nodes; a half
        # has at most 16 (an element has two bytes or more): four rounds.  A
        # lane whose element ends the chain (long literal, behind the input,
        # or leaving its half) fetches its own set.  The halves are strung
        # together afterwards.
        term = long_ | (LANE >= rem)
        nxl = LANE + enc
        stay = ~term & ((nxl ^ LANE) < 32)
        Rr = (1 << (LANE & 31)) | np.where(stay, 1 << (nxl & 31), 0)
        for k in range(4):
            top = np.array([int(x).bit_length() - 1 for x in Rr])
            Rr = Rr | Rr[(LANE & 32) + top]
        S = int(Rr[0])
        t0 = S.bit_length() - 1
        if not term[t0] and nxl[t0] < WAVE:
            S |= int(Rr[int(nxl[t0])]) << 32
        is_start = np.array([(S >> i) & 1 for i in range(WAVE)], dtype=bool)
        # reference walk, to check the doubling
        p, walk = 0, 0
        while p < WAVE:
            walk |= 1 << p
            p = int(nx[p])
        assert walk == S, (hex(walk), hex(S))
        elem = is_start & fits
        o = np.where(elem, olen, 0)
        incl = np.cumsum(o)
        keep = elem & (incl <= WMAX)
        # the window ends in front of the first start that is a long literal
        # or does not fit
        stop = is_start & ~fits
        if stop.any():
            keep &= LANE < int(LANE[stop][0])
        E = int(keep.sum())
        if E == 0:
            # lane 0 is a long literal (> 64 bytes): 256 B per instruction
            st.longlit += 1
            Lq, h0 = int(L[0]), int(hd[0])
            if (not long_[0]) or rem < h0 or src_len - (s + h0) < Lq or \
                    dst_len - d < Lq:
                return irregular()
            flush_partial(d)
            out[d:d + Lq] = src[s + h0:s + h0 + Lq]
            s += h0 + Lq
            d += Lq
            gflush = d
            ring_lo = d
            continue
        f_rel = incl - o
        last = int(LANE[keep][-1])
        W = int(incl[last])
        cur = last + int(enc[last])
        st.elements += E
        # ---- the reference's checks (src bounds hold by the TAIL rule) -----
        if d + W > dst_len:
            return irregular()
        dstp = d + f_rel
        cpy = keep & ~is_lit
        if (cpy & ((off == 0) | (off > dstp))).any():
            return irregular()
        assert not (keep & (s + LANE + enc > src_len)).any()
        # ---- expand ---------------------------------------------------------
        # ONE lane-parallel step: every element whose source is complete
        # before this window (literals; copies from in front of d) is copied
        # by its own lane, 16 bytes per trip.  Then the elements that read
        # this window's own output, and overlapping copies with a short
        # period, are swept in stream order, each by the whole wave (lane k =
        # byte k), so every source is complete when its element's turn comes.
        q = dstp - off                       # copy source position
        n = np.minimum(olen, off)            # source bytes a copy reads
        # (a lane stores whole 16-byte pieces: up to 15 bytes past the
        # window's output may be clobbered as well)
        safe_lo = max(ring_lo, d + W + 16 - R)   # ring history that no write
        ring_ok = q >= safe_lo                   # of this window can clobber
        far_ok = q + n <= gflush
        pad = (olen + 15) & ~15              # 16-byte loads stay in bounds
        if deep:
            assert (~(keep & is_lit) | (inner & (LANE + hd + pad <= rem))).all()
        lanewise = keep & np.where(
            is_lit, inner & (LANE + hd + pad <= rem),
            (q + n <= d) & (olen <= off) &
            np.where(ring_ok, True, far_ok & (q + 64 <= dst_len)))
        far = lanewise & ~is_lit & ~ring_ok

        def fence_for(limit):
            nonlocal visible, fenced
            if limit > fenced:
                st.fence += 1
                visible = gflush
                fenced = gflush
            assert limit <= visible

        if far.any():
            st.far += int(far.sum())
            fence_for(int((q + n)[far].max()))
        # Every lane stores WHOLE 16-byte pieces with one DS instruction; the
        # bytes past an element's end land on the following elements, whose
        # lanes are higher and whose stores of the same instruction therefore
        # win (tests/hw/lds_write_order.hip).  The pieces go last to first,
        # so the excess of an element's last piece is repaired by the first
        # pieces of its successors, which are written in the last trip.
        cmax = int(((olen[lanewise] + 15) // 16).max()) if lanewise.any() else 0
        for c in range(cmax - 1, -1, -1):
            act = lanewise & (16 * c < olen)
            reads = {}
            for i in LANE[act]:              # all loads of the trip first
                if is_lit[i]:
                    a = s + i + int(hd[i]) + 16 * c
                    assert a + 16 <= src_len
                    reads[i] = src[a:a + 16].copy()
                elif ring_ok[i]:
                    reads[i] = ring_read16(int(q[i]) + 16 * c)
                else:
                    a = int(q[i]) + 16 * c
                    assert a + 16 <= dst_len
                    reads[i] = out[a:a + 16].copy()
            for i in LANE[act]:              # one store instruction, lanes
                m = min(int(olen[i]) - 16 * c, 16)   # in ascending order
                wa = (int(dstp[i]) + 16 * c) & (R - 1)
                if wa + m > R:
                    continue                 # the element's bytes wrap: below
                ring[wa:wa + 16] = reads[i]  # may spill into the mirror
            for i in LANE[act]:              # (rare) exact, bytewise, wrapped
                m = min(int(olen[i]) - 16 * c, 16)
                wa = (int(dstp[i]) + 16 * c) & (R - 1)
                if wa + m > R:
                    ring_write(int(dstp[i]) + 16 * c, reads[i][:m])
        st.rounds += 1
        for i in LANE[keep & ~lanewise]:     # the sweep, in stream order
            st.wide += 1
            if is_lit[i]:                    # only at the end of the input
                a = s + i + int(hd[i])
                ring_write(int(dstp[i]), src[a:a + int(olen[i])])
                continue
            qi, oi, ni = int(q[i]), int(off[i]), int(olen[i])
            in_ring = qi >= safe_lo
            if not in_ring and qi + min(ni, oi) > gflush:
                # bytes the ring has lost and that are not stored yet (only
                # right after a long literal): store them, then read them
                flush_partial(int(dstp[i]))
            k = np.arange(ni)
            srcpos = qi + (k % oi)
            if in_ring:
                data = ring[srcpos & (R - 1)].copy()
            else:
                fence_for(qi + min(ni, oi))
                data = out[srcpos].copy()
            ring_write(int(dstp[i]), data)
        # the mirror once per window: its stores (whole 16-byte pieces, up to
        # d + W + 16) touched ring[0,16) or spilled over the ring's end.
        # Within the window nothing reads through the mirror what they
        # changed: such a source lies below safe_lo.
        a0 = d & (R - 1)
        if a0 < 16 or a0 + W + 16 > R:
            mirror()
        d += W
        s += cur
        flush_chunks()
    flush_partial(d)
    if d != dst_len:
        return ("irregular", s, d, bytes(out[:d]))
    return ("ok", bytes(out[:dst_len]))
