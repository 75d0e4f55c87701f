"""Round-level model of k_match_blocks / k_match_blocks_spec
(rust-snappy_amd/csrc/snapmi_compress.hip): one lane, one block.

What is modelled is the order of table reads and writes inside a round - the
entry of this round's probe and, with `spec`, the entry of the probe that
follows a miss are both read BEFORE anything the round writes, and what the
round wrote is forwarded in registers exactly as the kernel does it.  The
window, the stalls and the token buffers are not (they do not change what is
computed).  tests/test_model_match_cpu.py compares the stream the tokens
encode to with the oracle's."""

PROBE, CHAIN, EXTEND = 0, 1, 2


def _le32(b, i):
    return int.from_bytes(b[i:i + 4].ljust(4, b"\0"), "little")


def _hash(x, shift):
    return ((x * 0x1E35A7BD) & 0xFFFFFFFF) >> shift


def _common(a, b, limit):
    m = 0
    while m < limit and m < len(a) and m < len(b) and a[m] == b[m]:
        m += 1
    return m


def lane_tokens(block, spec):
    """Tokens (literal_len, copy_len, offset) of one block of >= 17 bytes,
    and the number of rounds it took."""
    n = len(block)
    shift, tsize = 24, 256
    while tsize < 16384 and tsize < n:
        shift -= 1
        tsize *= 2
    s_limit = n - 15
    first12 = block[0:12]
    table = {}               # slot -> (12 bytes at the position, position)
    tokens = []
    s, s_next, skip, mode, next_emit = 1, 2, 33, PROBE, 0
    p = c = mpos = mcand = 0
    rounds = 0
    while True:
        rounds += 1
        pos = p if mode == EXTEND else s
        q = block[pos - 1:pos + 11]          # bytes at pos - 1 (12 of them)
        r = block[pos:pos + 16]              # bytes at pos
        hprev = _hash(_le32(block, pos - 1), shift)
        hcur = _hash(_le32(block, pos), shift)
        # ---- loads of the round, before any of its stores
        if mode == EXTEND:
            A = block[c:c + 16]
        else:
            A = table.get(hcur)
        do_spec, A2, h2, t = False, None, 0, b""
        if spec and mode <= CHAIN:
            delta = 1 if mode == CHAIN else s_next - s
            if delta <= 3:
                do_spec = True
                t = block[pos + delta:pos + delta + 12]
                h2 = _hash(_le32(t, 0), shift)
                A2 = table.get(h2)
        matched = advance = tail = finished = False
        mend = 0

        def lookup(entry, at, bytes12):
            """The probe at position `at`: (hit, cand, common length up to
            12).  entry None = the reference's fresh table: position 0."""
            cand_bytes, cand = (entry if entry is not None else (first12, 0))
            if cand_bytes[0:4] == bytes12[0:4]:
                return True, cand, _common(cand_bytes, bytes12, 12)
            return False, cand, 0

        if mode <= CHAIN:
            was_chain = mode == CHAIN
            if was_chain:
                e_prev = (bytes(q[0:12]), s - 1)
                table[hprev] = e_prev
                if hprev == hcur:
                    A = e_prev
            e_cur = (bytes(r[0:12]), s)
            hit, cand, m = lookup(A, s, r[0:12])
            table[hcur] = e_cur
            if hit:
                mpos, mcand = s, cand
                if m < 12:
                    matched, mend = True, s + m
                else:
                    p, c, mode = s + 12, cand + 12, EXTEND
                    tail = p + 16 > n
            else:
                if was_chain:
                    s_next, skip = s + 1, 32
                advance = True
            if spec and advance and do_spec:
                s_old = s
                s = s_next
                step = skip >> 5
                s_next = s + step
                skip += step
                mode = PROBE
                advance = False
                if s_next > s_limit:
                    finished = True
                else:
                    assert s == s_old + (1 if was_chain else s - s_old)
                    if h2 == hcur:
                        A2 = e_cur
                    elif was_chain and h2 == hprev:
                        A2 = e_prev
                    hit, cand, m = lookup(A2, s, t)
                    table[h2] = (bytes(t), s)
                    if hit:
                        mpos, mcand = s, cand
                        if m < 12:
                            matched, mend = True, s + m
                        else:
                            p, c, mode = s + 12, cand + 12, EXTEND
                            tail = p + 16 > n
                    else:
                        advance = True
        else:
            m = _common(A, r, 16)
            if m < 16:
                matched, mend = True, p + m
            else:
                p += 16
                c += 16
                tail = p + 16 > n
        if tail:
            while p < n and block[p] == block[c]:
                p += 1
                c += 1
            matched, mend = True, p
        if matched:
            tokens.append((mpos - next_emit, mend - mpos, mpos - mcand))
            s = mend
            next_emit = mend
            mode = CHAIN
            if s >= s_limit:
                finished = True
        if advance:
            s = s_next
            step = skip >> 5
            s_next = s + step
            skip += step
            mode = PROBE
            if s_next > s_limit:
                finished = True
        if finished:
            if next_emit < n:
                tokens.append((n - next_emit, 0, 0))
            return tokens, rounds


def _put_literal(out, lit):
    n1 = len(lit) - 1
    if n1 <= 59:
        out.append(n1 << 2)
    elif n1 < 256:
        out += bytes([60 << 2, n1])
    else:
        out += bytes([61 << 2, n1 & 255, n1 >> 8])
    out += lit


def _put_copy(out, offset, ln):
    def copy2(k):
        out.extend([((k - 1) << 2) | 2, offset & 255, offset >> 8])
    while ln >= 68:
        copy2(64)
        ln -= 64
    if ln > 64:
        copy2(60)
        ln -= 60
    if ln <= 11 and offset <= 2047:
        out.extend([((offset >> 8) << 5) | ((ln - 4) << 2) | 1, offset & 255])
    else:
        copy2(ln)


def compress_one_block_stream(data, spec):
    """The raw stream of an input of at most 65536 bytes, and the rounds."""
    n = len(data)
    out = bytearray()
    v = n
    while v >= 128:
        out.append((v & 127) | 128)
        v >>= 7
    out.append(v)
    if n == 0:
        return bytes(out), 0
    if n < 17:
        _put_literal(out, data)
        return bytes(out), 0
    tokens, rounds = lane_tokens(data, spec)
    at = 0
    for lit, ln, off in tokens:
        if lit:
            _put_literal(out, data[at:at + lit])
            at += lit
        if ln:
            _put_copy(out, off, ln)
            at += ln
    assert at == n
    return bytes(out), rounds
