"""Lane-level model (numpy, 64-wide arrays) of k_decompress_streams3 in
rust-snappy_amd/csrc/snapmi_decompress.hip: the third-generation wave decoder
(256 compressed bytes per window, one ELEMENT per lane).

TEST INFRASTRUCTURE: the kernel's algorithm restated step by step so that its
logic can be checked on the CPU against the oracle
(tests/test_model_decoder_cpu.py).  It mirrors the kernel's variable names; it
is not used by the product.

What differs from the second generation (tests/model_decoder.py):

  * a window is G groups of 64 input bytes (G = 4): lane l looks at the tag
    bytes at s + 64 g + l only far enough to know how long the element there
    would be (no offsets, no lengths beyond the tag);
  * element starts: per group one 32-bit set per lane and half (positions of
    its own 32-byte sub-window), which contains the frontier as its highest
    bit; four rounds of R |= R[frontier]; the 2 G sub-windows are strung
    together by a scalar walk;
  * the starts are COMPACTED: the t-th element of the window goes to lane t
    (rank by population count, through a 64 G-byte table in LDS), and only
    there is the element decoded in full - every lane of the expensive part
    holds a real element;
  * literals of 61 bytes and more (a length byte behind the tag) always end a
    window and are copied by the whole wave;
  * the copy step: every element whose source is complete before the window
    is copied by its lane (16 bytes per trip, whole-piece stores resolved by
    lane order), then the LATE elements - copies that read the window's own
    output or overlap themselves - are moved one by one, in stream order, by
    the whole wave, byte by byte;
  * a window needs kTail bytes of input in front of it (all loads of the
    window then stay inside the input); the last windows of a stream are the
    sequential decoder's.

decode(comp) -> ("ok", bytes) | ("irregular", s, d, prefix)
"""
import numpy as np

R = 4096          # ring bytes
WMAX = 2048       # output bytes per window at most
WAVE = 64
G = 4             # groups of 64 input bytes per window (kG3)
# every load of a window stays inside the input when this much is left: the
# last position (64 G - 1), an element header (tag + 4 offset bytes / a tag
# and 60 literal bytes rounded up to whole 16-byte pieces)
TAIL = 64 * G + 1 + 64 + 16

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
        self.windows = self.runs = self.trips = self.sweeps = 0
        self.far = self.longlit = self.flush_partial = self.fence = 0
        self.elements = self.mirrors = 0


def top_bit(x):
    return int(x).bit_length() - 1


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
        st.mirrors += 1
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

    def fence_for(limit):
        nonlocal visible, fenced
        if limit > fenced:
            st.fence += 1
            visible = gflush
            fenced = gflush
        assert limit <= visible

    while s < src_len:
        rem = src_len - s
        if rem < TAIL:
            return irregular()          # the sequential decoder's
        st.windows += 1
        # ---- 1. how long is "the element at src[s + 64 g + lane]" ----------
        tagb = [src[s + 64 * g:s + 64 * g + WAVE].astype(np.int64)
                for g in range(G)]
        pos_g = [64 * g + LANE for g in range(G)]
        typ_g = [t & 3 for t in tagb]
        n6_g = [t >> 2 for t in tagb]
        lng_g = [(typ_g[g] == 0) & (n6_g[g] >= 60) for g in range(G)]
        enc_g = [np.where(typ_g[g] == 0, n6_g[g] + 2,
                          np.array([0, 2, 3, 5])[typ_g[g]]) for g in range(G)]
        nx_g = [pos_g[g] + enc_g[g] for g in range(G)]
        # ---- 2. element starts ------------------------------------------
        Rr = []
        for g in range(G):
            stay = ~lng_g[g] & ((nx_g[g] ^ pos_g[g]) < 32)
            r = (1 << (LANE & 31)) | np.where(stay, 1 << (nx_g[g] & 31), 0)
            for k in range(4):
                top = np.array([top_bit(x) for x in r])
                r = r | r[(LANE & 32) + top]
            Rr.append(r)
        S = [0] * G                      # start lanes per group
        k, e = 0, 0
        while True:
            g, half = k >> 1, k & 1
            Sk = int(Rr[g][32 * half + e])
            assert Sk & ((1 << e) - 1) == 0
            S[g] |= Sk << (32 * half)
            lt = 32 * half + top_bit(Sk)         # the sub-window's last start
            if lng_g[g][lt]:
                break
            nxa = int(nx_g[g][lt])
            if nxa >= 64 * G:
                break
            assert (nxa >> 5) > k
            k, e = nxa >> 5, nxa & 31
        # reference walk, to check the sets
        p, walk = 0, [0] * G
        while p < 64 * G:
            g, l = p >> 6, p & 63
            walk[g] |= 1 << l
            if lng_g[g][l]:
                break
            p = int(nx_g[g][l])
        assert walk == S, ([hex(x) for x in walk], [hex(x) for x in S])
        # ---- 3. compaction: the t-th start goes to lane t -------------------
        postab = np.zeros(64 * G, dtype=np.int64)
        base = 0
        for g in range(G):
            for l in range(WAVE):
                if (S[g] >> l) & 1:
                    rank = base + bin(S[g] & ((1 << l) - 1)).count("1")
                    postab[rank] = 64 * g + l
            base += bin(S[g]).count("1")
        n_el = min(base, WAVE)
        act = LANE < n_el
        pos = np.where(act, postab[:WAVE], 0)
        # ---- 4. the element, in full, one per lane --------------------------
        w = np.array([int.from_bytes(src[s + p:s + p + 8].tobytes(), "little")
                      for p in pos], dtype=np.uint64)
        tag = (w & np.uint64(0xFF)).astype(np.int64)
        b14 = ((w >> np.uint64(8)) & np.uint64(0xFFFFFFFF)).astype(np.int64)
        typ = tag & 3
        n6 = tag >> 2
        is_lit = typ == 0
        lng = is_lit & (n6 >= 60)
        cnb = np.where(typ == 1, 1, np.where(typ == 2, 2, 4))
        clen = np.where(typ == 1, 4 + (n6 & 7), n6 + 1)
        off = np.where(typ == 1, ((tag >> 5) << 8) | (b14 & 0xFF),
                       np.where(typ == 2, b14 & 0xFFFF, b14))
        olen = np.where(is_lit, n6 + 1, clen)
        enc = np.where(is_lit, n6 + 2, 1 + cnb)
        o = np.where(act & ~lng, olen, 0)
        incl = np.cumsum(o)
        keep = act & ~lng & (incl <= WMAX)
        if (act & lng).any():            # the window ends in front of it
            keep &= LANE < int(LANE[act & lng][0])
        E = int(keep.sum())
        if E == 0:
            # lane 0 is a literal with a length byte: moved by the whole wave
            assert lng[0]
            st.longlit += 1
            lnb = int(n6[0]) - 59
            Lq = (int(b14[0]) & ((1 << (8 * lnb)) - 1)) + 1
            h0 = 1 + lnb
            if src_len - (s + h0) < Lq or dst_len - d < Lq:
                return irregular()
            flush_partial(d)
            out[d:d + Lq] = src[s + h0:s + h0 + Lq]
            s += h0 + Lq
            d += Lq
            gflush = d
            ring_lo = d
            continue
        assert keep[:E].all()            # a prefix of the lanes
        f_rel = incl - o
        last = E - 1
        W = int(incl[last])
        cur = int(pos[last]) + int(enc[last])
        st.elements += E
        # ---- the reference's checks --------------------------------------
        if d + W > dst_len:
            return irregular()
        dstp = d + f_rel
        cpy = keep & ~is_lit
        if (cpy & ((off == 0) | (off > dstp))).any():
            return irregular()
        assert s + cur <= src_len
        # ---- 5. the copy step ---------------------------------------------
        q = dstp - off
        qe = q + olen
        safe_lo = max(ring_lo, d + W + 16 - R)
        in_ring = q >= safe_lo
        # A far source must have been stored (true by construction while the
        # ring is whole).  LATE elements are moved one by one, in stream
        # order, by the whole wave after everything else: copies that read
        # this window's own output (qe > d; that includes a copy that
        # overlaps itself) and sources neither in the ring nor stored.
        whole = ring_lo + R <= d + W + 16
        far_ok = (qe <= gflush) & (q + 64 <= dst_len)
        if whole:
            assert (~(cpy & ~in_ring) | far_ok).all()
        late = cpy & ((qe > d) | (~in_ring & ~far_ok))
        assert (~(cpy & (off < olen)) | late).all()
        lw = keep & ~late
        far = lw & ~is_lit & ~in_ring
        if far.any():
            st.far += int(far.sum())
            fence_for(int(qe[far].max()))
        # whole 16-byte pieces, last piece first, one store instruction per
        # trip (lanes applied in ascending order): the excess of a short
        # element lands on the elements behind it - higher lanes of the same
        # instruction, which win, or late elements, written afterwards
        st.runs += 1
        cmax = int(((olen[lw] + 15) // 16).max()) if lw.any() else 0
        for c in range(cmax - 1, -1, -1):
            st.trips += 1
            actc = lw & (16 * c < olen)
            reads = {}
            for i in LANE[actc]:             # all loads of the trip first
                if is_lit[i]:
                    p0 = s + int(pos[i]) + 1 + 16 * c
                    assert p0 + 16 <= src_len
                    reads[i] = src[p0:p0 + 16].copy()
                elif in_ring[i]:
                    reads[i] = ring_read16(int(q[i]) + 16 * c)
                else:
                    p0 = int(q[i]) + 16 * c
                    assert p0 + 16 <= dst_len
                    reads[i] = out[p0:p0 + 16].copy()
            for i in LANE[actc]:             # one store instruction
                m = min(int(olen[i]) - 16 * c, 16)
                wa = (int(dstp[i]) + 16 * c) & (R - 1)
                if wa + m > R:
                    continue                 # the element's bytes wrap: below
                ring[wa:wa + 16] = reads[i]  # may spill into the mirror
            for i in LANE[actc]:             # (rare) exact, bytewise, wrapped
                m = min(int(olen[i]) - 16 * c, 16)
                wa = (int(dstp[i]) + 16 * c) & (R - 1)
                if wa + m > R:
                    ring_write(int(dstp[i]) + 16 * c, reads[i][:m])
        for i in LANE[late]:                 # in stream order: lane k = byte k
            st.sweeps += 1
            qi, oi, ni = int(q[i]), int(off[i]), int(olen[i])
            inr = qi >= safe_lo
            if not inr and qi + min(ni, oi) > gflush:
                flush_partial(int(dstp[i]))
            kk = np.arange(ni)
            srcpos = qi + (kk % oi)
            if inr:
                data = ring[srcpos & (R - 1)].copy()
            else:
                fence_for(qi + min(ni, oi))
                data = out[srcpos].copy()
            ring_write(int(dstp[i]), data)
        # the mirror for the next window
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
