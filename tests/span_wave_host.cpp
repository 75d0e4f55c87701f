// TEST INFRASTRUCTURE: k_compress_spans (rust-snappy_amd/csrc/snapmi_compress.hip)
// on the host.  The uniform half of a step - span_walk() of
// rust-snappy_amd/csrc/snapmi_span.hpp, the very text the kernel compiles - runs
// as it is; the 64 lanes around it (hash, the lane-ordered table exchange,
// the 16-byte compare, the lane-ordered store that puts the table right, the
// schedule-ordered step of long miss runs) are emulated one lane after the
// other in ascending order, which is the order gfx950 applies the lanes of one
// DS instruction in (tests/hw/lds_atomic_order.hip, lds_write_order.hip).
// tests/test_span_wave_cpu.py compares the stream with the oracle's.  Never
// linked into the product library.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <string.h>

#include <vector>

#include "snapmi_span.hpp"

namespace {
using namespace snapmi;

struct Delta {
    uint32_t d[448];
    Delta()
    {
        uint32_t skip = 32, p = 0;
        for (int i = 0; i < 448; i++) {
            d[i] = p < 0x100000u ? p : 0x100000u;
            const uint32_t step = skip >> 5;
            p += step;
            skip += step;
        }
    }
};
const Delta kDelta;

uint32_t le32(const uint8_t *p)
{
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
uint32_t common(const uint8_t *a, const uint8_t *b, uint32_t lim)
{
    uint32_t m = 0;
    while (m < lim && a[m] == b[m])
        m++;
    return m;
}

struct Token {
    uint32_t lit, len, off;
};
struct Sink {
    std::vector<Token> t;
    void token(uint32_t lit, uint32_t len, uint32_t off)
    {
        t.push_back({lit, len, off});
    }
};
struct Lanes {
    uint32_t mv[64], ov[64];
    uint32_t m(uint32_t l) const { return mv[l]; }
    uint32_t old(uint32_t l) const { return ov[l]; }
};

struct Out {
    uint8_t *p;
    uint32_t cap, bad;
    const uint8_t *in;
    uint32_t in8(uint32_t k) { return in[k]; }
    uint32_t in32(uint32_t k) { return le32(in + k); }
    void out8(uint32_t k, uint32_t v)
    {
        if (k >= cap)
            bad = 1;
        else
            p[k] = (uint8_t)v;
    }
    void out32(uint32_t k, uint32_t v)
    {
        if (k + 4 > cap)
            bad = 1;
        else
            memcpy(p + k, &v, 4);
    }
};

// The wave of span_par_walk on the host: 64 values side by side, every
// operator elementwise, the cross-lane operations as array gathers.
struct HV {
    uint32_t v[64];
    HV() { memset(v, 0, sizeof v); }
    explicit HV(uint32_t x)
    {
        for (int i = 0; i < 64; i++)
            v[i] = x;
    }
};
struct HB {
    bool v[64];
};
#define HV_OP(op)                                                             \
    HV operator op(const HV &a, const HV &b)                                  \
    {                                                                         \
        HV r;                                                                 \
        for (int i = 0; i < 64; i++)                                          \
            r.v[i] = a.v[i] op b.v[i];                                        \
        return r;                                                             \
    }
HV_OP(+) HV_OP(-) HV_OP(&) HV_OP(|)
#undef HV_OP
struct WaveHost {
    typedef HV u32;
    typedef HB b1;
    mutable uint64_t bperms = 0, cuts = 0;
    void count_cut() const { cuts++; }
    HV lane() const
    {
        HV r;
        for (int i = 0; i < 64; i++)
            r.v[i] = i;
        return r;
    }
    HV sel(const HB &c, const HV &a, const HV &b) const
    {
        HV r;
        for (int i = 0; i < 64; i++)
            r.v[i] = c.v[i] ? a.v[i] : b.v[i];
        return r;
    }
#define HB_CMP(name, op)                                                      \
    HB name(const HV &a, const HV &b) const                                   \
    {                                                                         \
        HB r;                                                                 \
        for (int i = 0; i < 64; i++)                                          \
            r.v[i] = a.v[i] op b.v[i];                                        \
        return r;                                                             \
    }
    HB_CMP(lt, <) HB_CMP(ge, >=) HB_CMP(eq, ==)
#undef HB_CMP
    HB band(const HB &a, const HB &b) const
    {
        HB r;
        for (int i = 0; i < 64; i++)
            r.v[i] = a.v[i] && b.v[i];
        return r;
    }
    HB bor(const HB &a, const HB &b) const
    {
        HB r;
        for (int i = 0; i < 64; i++)
            r.v[i] = a.v[i] || b.v[i];
        return r;
    }
    HB bnot(const HB &a) const
    {
        HB r;
        for (int i = 0; i < 64; i++)
            r.v[i] = !a.v[i];
        return r;
    }
    uint64_t ballot(const HB &a) const
    {
        uint64_t m = 0;
        for (int i = 0; i < 64; i++)
            m |= (uint64_t)a.v[i] << i;
        return m;
    }
    HV bperm(const HV &idx, const HV &val) const
    {
        HV r;
        for (int i = 0; i < 64; i++) {
            if (idx.v[i] > 63)
                abort(); // the kernel's ds_bpermute would wrap silently
            r.v[i] = val.v[idx.v[i]];
        }
        bperms++;
        return r;
    }
    uint32_t readlane(const HV &val, uint32_t l) const
    {
        if (l > 63)
            abort();
        return val.v[l];
    }
    HB bit(uint64_t mask, const HV &i) const
    {
        HB r;
        for (int k = 0; k < 64; k++) {
            if (i.v[k] > 63)
                abort();
            r.v[k] = (mask >> i.v[k]) & 1;
        }
        return r;
    }
    HV next_bit(uint64_t mask, const HV &t) const
    {
        HV r;
        for (int k = 0; k < 64; k++) {
            if (t.v[k] > 63)
                abort();
            const uint64_t x = mask >> t.v[k];
            r.v[k] = x ? t.v[k] + (uint32_t)__builtin_ctzll(x) : 64;
        }
        return r;
    }
    HV shr_lo(uint64_t mask, const HV &i) const
    {
        HV r;
        for (int k = 0; k < 64; k++) {
            if (i.v[k] > 63)
                abort();
            r.v[k] = (uint32_t)(mask >> i.v[k]);
        }
        return r;
    }
    HV prev_bit(uint64_t mask, const HV &t) const
    {
        HV r;
        for (int k = 0; k < 64; k++) {
            if (t.v[k] > 63)
                abort();
            const uint64_t x = mask & ((1ull << t.v[k]) - 1);
            r.v[k] = x ? 63u - (uint32_t)__builtin_clzll(x) : 64;
        }
        return r;
    }
};

// span_par_walk (and span_fast_ok_w) over the host's wave
uint32_t par_walk(SpanState &st, uint64_t hits, uint64_t cbits, const Lanes &ln,
                  uint32_t &emit, uint64_t &vh, HV &lit, uint64_t &T,
                  uint32_t &at, uint64_t &cuts, uint32_t n, bool *ok_w)
{
    WaveHost w;
    HV m, old;
    HB cb;
    for (uint32_t l = 0; l < 64; l++) {
        m.v[l] = ln.mv[l];
        old.v[l] = ln.ov[l];
        cb.v[l] = (cbits >> l) & 1;
    }
    if (ok_w)
        *ok_w = span_fast_ok_w(w, st, hits, n);
    const uint32_t rc =
        span_par_walk(w, st, hits, m, old, cb, emit, vh, lit, T, at);
    cuts = w.cuts;
    return rc;
}
} // namespace

// stats[0] window steps, [1] schedule steps, [2] cuts of fast steps, [3] long
// matches,
// [4] tokens, [5] lanes touched, [6] tokens of window steps, [7] window
// steps that took the fast walk, [8] the table's entries
extern "C" uint32_t span_wave_compress(const uint8_t *src, uint32_t n,
                                       uint8_t *out, uint32_t out_cap,
                                       uint64_t *stats)
{
    if (n == 0 || n > 65536)
        return 0x80000000u;
    Out o{out, out_cap, 0, src};
    uint32_t d = 0;
    for (uint32_t v = n;;) {
        if (v < 128) {
            o.out8(d++, v);
            break;
        }
        o.out8(d++, (v & 127) | 128);
        v >>= 7;
    }
    if (n < 17) {
        d = tiny_put_literal(o, d, 0, n);
        return o.bad ? 0x80000001u : d;
    }
    // the table the kernel instance for this block length has room for
    // (k_match_spans_8k / the 64 KiB kernels; 4 096 for the 20-table variant
    // that was measured and dropped): every index below is
    // checked against what the reference would allocate (tsize) AND against
    // that room
    const uint32_t room = n <= 4096 ? 4096 : (n <= 8192 ? 8192 : 16384);
    uint32_t shift = 24, tsize = 256;
    for (; tsize < room && tsize < n; tsize *= 2)
        shift--;
    stats[8] = tsize;
    std::vector<uint16_t> table_mem(room, 0);
    uint16_t *const table = table_mem.data();
    const uint32_t s_limit = n - 15;
    SpanState st{1, 0, 0, 0};
    Sink sink;
    Lanes ln;
    uint32_t run0 = 1;
    bool done = false;
    while (!done) {
        if (!st.chain && st.q >= kSpanRun) {
            // the schedule-ordered step (k_compress_blocks' batch, q > 0):
            // lane l = probe q + l of the run that began at run0
            stats[1]++;
            uint32_t p[64], h[64], cand[64], m[64];
            bool valid[64];
            bool any_invalid = false;
            int kh = -1;
            for (uint32_t l = 0; l < 64; l++) {
                p[l] = run0 + kDelta.d[st.q + l];
                const uint32_t nextp = run0 + kDelta.d[st.q + l + 1];
                valid[l] = nextp <= s_limit;
                any_invalid |= !valid[l];
                cand[l] = 0;
                m[l] = 0;
                if (valid[l]) {
                    h[l] = tiny_hash(le32(src + p[l]), shift);
                    if (h[l] >= tsize)
                        return 0x80000004u;
                    cand[l] = table[h[l]];
                    table[h[l]] = (uint16_t)p[l];
                    m[l] = common(src + p[l], src + cand[l], 16);
                    if (m[l] >= 4 && kh < 0)
                        kh = (int)l;
                }
            }
            if (kh < 0) {
                if (any_invalid)
                    break;
                st.q += 64;
                continue;
            }
            for (uint32_t l = kh + 1; l < 64; l++)
                if (valid[l] && cand[l] <= p[kh])
                    table[h[l]] = (uint16_t)cand[l];
            uint32_t len = m[kh];
            if (len == 16)
                len += common(src + p[kh] + 16, src + cand[kh] + 16,
                              n - p[kh] - 16);
            sink.token(p[kh] - st.next_emit, len, p[kh] - cand[kh]);
            st.s = p[kh] + len;
            st.next_emit = st.s;
            st.chain = 1;
            st.q = 0;
            if (st.s >= s_limit)
                break;
            continue;
        }
        // the window step
        stats[0]++;
        const uint32_t base = st.s, lo = st.s - st.chain;
        bool act[64];
        uint32_t h[64];
        uint64_t hits = 0, cbits = 0;
        for (uint32_t l = 0; l < 64; l++) {
            const uint32_t P = base - 1 + l;
            act[l] = l ? P + 16 <= n : st.chain != 0;
            ln.mv[l] = 0;
            ln.ov[l] = 0;
            if (!act[l])
                continue;
            h[l] = tiny_hash(le32(src + P), shift);
            if (h[l] >= tsize)
                return 0x80000004u;
            ln.ov[l] = table[h[l]];
            table[h[l]] = (uint16_t)P;
            ln.mv[l] = common(src + P, src + ln.ov[l], 16);
            if (l && ln.mv[l] >= 4)
                hits |= 1ull << l;
            if (ln.ov[l] >= lo)
                cbits |= 1ull << l;
        }
        uint64_t T = 0;
        uint32_t at = 0, rc = kSpanCont;
        const size_t tok0 = sink.t.size();
        // the kernel's order: the fast walk where its conditions hold, the
        // exact walk otherwise
        bool fast = span_fast_ok(st, hits, n);
        if (fast) {
            uint64_t vh, cuts = 0;
            HV lit;
            uint32_t emit = st.next_emit;
            bool ok_w = false;
            rc = par_walk(st, hits, cbits, ln, emit, vh, lit, T, at, cuts, n,
                          &ok_w);
            if (!ok_w)
                return 0x80000005u; // span_fast_ok_w disagrees
            stats[7]++;
            for (uint32_t l = 1; l < 64; l++) {
                if (!((vh >> l) & 1))
                    continue;
                const uint32_t P = base - 1 + l;
                sink.token(lit.v[l], ln.mv[l], P - ln.ov[l]);
            }
            if (st.next_emit != emit)
                return 0x80000003u;
            stats[2] += cuts;
        }
        if (!fast) {
            WaveHost w0;
            if (span_fast_ok_w(w0, st, hits, n))
                return 0x80000005u;
            rc = span_walk(st, hits, cbits, s_limit, ln, sink, T, at);
        }
        stats[5] += (uint64_t)__builtin_popcountll(T);
        stats[6] += sink.t.size() - tok0;
        for (uint32_t l = 0; l < 64; l++) {
            if (!act[l])
                continue;
            const bool t = (T >> l) & 1, c = (cbits >> l) & 1;
            if (t)
                table[h[l]] = (uint16_t)(base - 1 + l);
            else if (!c)
                table[h[l]] = (uint16_t)ln.ov[l];
        }
        if (rc == kSpanLong) {
            stats[3]++;
            const uint32_t P = st.s, cand = ln.ov[at];
            const uint32_t len =
                16 + common(src + P + 16, src + cand + 16, n - P - 16);
            sink.token(P - st.next_emit, len, P - cand);
            st.s = P + len;
            st.next_emit = st.s;
            st.chain = 1;
            st.q = 0;
            if (st.s >= s_limit)
                done = true;
        } else if (rc == kSpanDone) {
            done = true;
        } else if (!st.chain && st.q >= kSpanRun) {
            run0 = st.s - st.q;
        }
    }
    stats[4] += sink.t.size();
    uint32_t atpos = 0;
    for (const Token &t : sink.t) {
        if (t.lit) {
            d = tiny_put_literal(o, d, atpos, t.lit);
            atpos += t.lit;
        }
        d = tiny_put_copy(o, d, t.off, t.len);
        atpos += t.len;
    }
    if (st.next_emit < n)
        d = tiny_put_literal(o, d, st.next_emit, n - st.next_emit);
    return o.bad ? 0x80000001u : d;
}

// Differential check of the two walks on RANDOM per-lane results (not only the
// ones real data produces): wherever the fast walk may run, its inserted
// lanes, its tokens and the state it leaves must be the exact walk's.  Returns
// 0, or the number of the first case that differs.
// seen[0..3]: windows compared, with a cut, ending in a long match, ending
// in a run.
extern "C" uint32_t span_walk_diff(uint32_t seed, uint32_t cases,
                                   uint64_t *seen)
{
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    auto rnd = [&]() {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        return x;
    };
    for (uint32_t c = 1; c <= cases; c++) {
        Lanes ln;
        const uint32_t base = 1000 + (uint32_t)(rnd() % 30000);
        const uint32_t n = 65536, s_limit = n - 15;
        SpanState st{base, (uint32_t)(rnd() % 20), (uint32_t)(rnd() & 1), 0};
        if (st.chain)
            st.q = 0;
        st.next_emit = base - (uint32_t)(rnd() % 50) - (st.chain ? 0 : st.q);
        const uint32_t lo = base - st.chain;
        uint64_t hits = 0, cbits = 0;
        const uint32_t density = 1 + (uint32_t)(rnd() % 6);   // hits per 8
        const uint32_t cdens = (uint32_t)(rnd() % 12);        // C bits per 64
        for (uint32_t l = 0; l < 64; l++) {
            ln.mv[l] = 0;
            ln.ov[l] = (uint32_t)(rnd() % (lo - 1)); // a position in front
            const bool active = l ? true : st.chain != 0;
            if (!active)
                continue;
            if (l && rnd() % 8 < density) {
                const uint32_t r = (uint32_t)(rnd() % 16);
                ln.mv[l] = r < 11 ? 4 + r % 8 : (r < 14 ? 12 + r % 4 : 16);
                hits |= 1ull << l;
            } else {
                ln.mv[l] = (uint32_t)(rnd() % 4);
            }
            const uint32_t first = st.chain ? 0 : 1;
            if (l > first && rnd() % 64 < cdens) { // a pred in the window
                const uint32_t p = first + (uint32_t)(rnd() % (l - first));
                ln.ov[l] = base - 1 + p;
                cbits |= 1ull << l;
            }
        }
        if (!span_fast_ok(st, hits, n))
            continue;
        // exact walk
        SpanState sa = st;
        Sink ka;
        uint64_t Ta = 0;
        uint32_t ata = 0;
        const uint32_t ra =
            span_walk(sa, hits, cbits, s_limit, ln, ka, Ta, ata);
        // the lane-parallel walk (span_par_walk), as the kernel drives it
        uint32_t cutc = 0;
        {
            SpanState sc = st;
            uint64_t vhc, Tc, cuts = 0;
            HV lit;
            uint32_t emitc = st.next_emit, atc = 0;
            bool ok_w = false;
            const uint32_t rcc = par_walk(sc, hits, cbits, ln, emitc, vhc, lit,
                                          Tc, atc, cuts, n, &ok_w);
            if (!ok_w)
                return c;
            if (rcc != ra || Tc != Ta || sc.s != sa.s ||
                sc.next_emit != sa.next_emit || emitc != sa.next_emit)
                return c;
            if ((uint32_t)__builtin_popcountll(vhc) != ka.t.size())
                return c;
            size_t i = 0;
            for (uint32_t l = 1; l < 64; l++) {
                if (!((vhc >> l) & 1))
                    continue;
                if (ka.t[i].lit != lit.v[l] || ka.t[i].len != ln.mv[l] ||
                    ka.t[i].off != base - 1 + l - ln.ov[l])
                    return c;
                i++;
            }
            if (ra == kSpanLong && atc != ata)
                return c;
            if (ra != kSpanLong &&
                (sa.chain != sc.chain || (!sa.chain && sa.q != sc.q)))
                return c;
            cutc = (uint32_t)cuts;
        }
        seen[0]++;
        seen[1] += cutc;
        seen[2] += ra == kSpanLong;
        seen[3] += !cutc && ra == kSpanCont && sa.s == base + 63;
    }
    return 0;
}
