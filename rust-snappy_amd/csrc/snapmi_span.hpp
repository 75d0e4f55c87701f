// snapmi: the walk of k_compress_spans (snapmi_compress.hip) - the uniform
// (scalar) half of a wavefront's step over a WINDOW of 63 consecutive
// positions of its block.
//
// The wavefront-per-block kernel of rounds 1-3 (k_compress_blocks) evaluates
// 64 probes of the reference's schedule per step and stops at the first hit:
// one LDS atomic + one candidate gather + one ballot PER EMITTED COPY, ~2 100
// cycles each, ~7 000 copies per 64 KiB of text.  But everything a step needs
// from memory - "what did the table hold for the hash at position P" and "how
// many bytes match there" - can be fetched for ALL positions of a window at
// once, whether the parse will look them up or not; what is sequential in the
// reference (src/compress.rs:195-317: which positions are looked up, which
// are inserted) is then a walk over per-lane results that are already in
// registers: a handful of scalar instructions per event instead of a memory
// round trip.  Lane L of a step holds position base - 1 + L:
//
//   old(L)  what ONE lane-ordered LDS exchange returned for the slot of that
//           position's hash: the table's entry, or - if a lower lane of this
//           very step has the same hash - that lane's position ("C bit");
//   m(L)    common prefix (0..16) of the 16 bytes at the position and the 16
//           bytes at old(L); hit = m >= 4.
//
// The exchange has written EVERY position of the window into the table; the
// walk decides which of them the reference really inserts (`touched`), and one
// lane-ordered store afterwards puts every slot right: touched lanes write
// their position, untouched lanes without a C bit give back what they
// displaced.  That is exact as long as a lane with a C bit is only touched
// when the lower lane it collided with was touched too (then that lane's
// position IS the reference's candidate, and the bytes compared were the
// right ones); otherwise the walk stops in front of it and the next step
// starts there with a fresh exchange ("cut": 1-2 % of the steps on text).
//
// Runs of misses are walked with one find-first-set over (hits | C bits).  A
// run of more than 32 misses leaves the window regime (the reference's stride
// grows, src/compress.rs:207-211): the caller continues it with the
// schedule-ordered step of k_compress_blocks.  A match of 16 bytes or more is
// finished by the caller (extend_match), then the next step starts behind it.
//
// The same text runs on the host: tests/span_wave_host.cpp emulates the 64
// lanes (exchange and store in ascending lane order) around this walk and
// tests/test_span_wave_cpu.py checks its bytes against the reference's.
#ifndef SNAPMI_SPAN_HPP
#define SNAPMI_SPAN_HPP

#include <stdint.h>

#include "snapmi_tiny.hpp" // SNAPMI_LANE_FN

namespace snapmi {

// probes 0 .. kSpanRun of a run lie at consecutive positions (skip reaches 64
// after 32 probes, src/compress.rs:207-211); the window regime handles probes
// 0 .. kSpanRun - 1, whose limit check is "position + 1 <= s_limit"
constexpr uint32_t kSpanRun = 32;

struct SpanState {
    uint32_t s;         // position of the next lookup
    uint32_t q;         // chain == 0: probes of the current run already done
    uint32_t chain;     // 1: a copy ended at s; s - 1 is not inserted yet
    uint32_t next_emit; // first byte not yet covered by a token
};

enum : uint32_t {
    kSpanCont = 0, // state says where the next step starts
    kSpanLong = 1, // lane `at` hit with >= 16 equal bytes: the caller extends
                   // the match, emits the token and restarts behind it
    kSpanDone = 2, // the block's parse is over (src/compress.rs:212-214,275-277)
};

// LN: uniform accessors of per-lane results, LN::m(lane), LN::old(lane).
// SINK: SINK::token(literal_len, copy_len, offset).
// hits / cbits: bit L = lane L hit / has a C bit (only lanes that took part in
// the exchange).  touched: bit L set = the reference inserted lane L's
// position (bit 0 = the insert of s - 1 when st.chain was set).
template <class LN, class SINK>
SNAPMI_LANE_FN uint32_t span_walk(SpanState &st, const uint64_t hits,
                                  const uint64_t cbits, const uint32_t s_limit,
                                  const LN &ln, SINK &sink, uint64_t &touched,
                                  uint32_t &at)
{
    const uint32_t base = st.s; // lane L = position base - 1 + L
    uint64_t T = st.chain ? 1ull : 0ull;
    uint32_t L = 1, q = st.q;
    bool chain = st.chain != 0;
    const uint64_t stop = hits | cbits;
    uint32_t rc = kSpanCont;
    // (the state is written once, behind the loop, from these: stores into
    // `st` from a dozen branches keep the struct in scratch memory)
    uint32_t ns = base, nq = st.q, nchain = st.chain, nemit = st.next_emit;
    for (;;) {
        uint32_t P = base - 1 + L;
        if (!chain) {
            // a run of probes, one position apart: lanes L, L+1, .. are probes
            // q, q+1, ..; plain misses (no hit, no C bit) up to the first
            // lane that is neither, the window's end, probe kSpanRun or the
            // block's limit (the probe at P needs P + 1 <= s_limit)
            uint32_t room = 64 - L;
            if (kSpanRun - q < room)
                room = kSpanRun - q;
            const uint32_t lim = s_limit > P ? s_limit - P : 0;
            if (lim < room)
                room = lim;
            const uint64_t ahead = L < 64 ? stop >> L : 0;
            uint32_t k = ahead ? (uint32_t)__builtin_ctzll(ahead) : 64;
            if (k > room)
                k = room;
            if (k) {
                T |= (k == 64 ? ~0ull : ((1ull << k) - 1)) << L;
                q += k;
                L += k;
                P += k;
            }
            if (k == room) { // a limit, whatever lane L holds
                ns = P;
                nq = q;
                nchain = 0;
                rc = P + 1 > s_limit ? kSpanDone : kSpanCont;
                break;
            }
        }
        // lane L (< 64): a chain check, or a probe that hits or has a C bit
        const uint64_t bit = 1ull << L;
        if (cbits & bit) {
            const uint32_t pred = ln.old(L) - (base - 1);
            if (!((T >> pred) & 1)) { // cut: the next step starts here
                ns = P;
                nq = q;
                nchain = chain ? 1 : 0;
                break;
            }
        }
        T |= bit;
        if (hits & bit) {
            const uint32_t m = ln.m(L);
            if (m >= 16) {
                at = L;
                ns = P;
                rc = kSpanLong;
                break;
            }
            sink.token(P - nemit, m, P - ln.old(L));
            const uint32_t e = P + m;
            nemit = e;
            ns = e;
            nq = 0;
            nchain = 1;
            if (e >= s_limit) {
                rc = kSpanDone;
                break;
            }
            // src/compress.rs:290-297: insert e - 1, then the check at e
            const uint32_t Li = L + m - 1;
            if (Li >= 63)
                break; // both belong to the next step
            const uint64_t ibit = 1ull << Li;
            if (cbits & ibit) {
                const uint32_t pred = ln.old(Li) - (base - 1);
                if (!((T >> pred) & 1))
                    break;
            }
            T |= ibit;
            L = Li + 1;
            chain = true;
        } else {
            // a miss: behind a chain check the run starts anew
            // (src/compress.rs:310-312)
            q = chain ? 0 : q + 1;
            chain = false;
            L++;
        }
    }
    st.s = ns;
    st.q = nq;
    st.chain = nchain;
    st.next_emit = nemit;
    touched = T;
    return rc;
}

// ---------------------------------------------------------------------
// The fast walk.  span_walk above costs the scalar unit ~70 instructions and
// ten branches per copy (measured in round 4: 5 700 of a step's 10 000 cycles
// on text, nine copies per step).  But where the parse GOES depends on nothing
// but the hit mask and the match lengths, so it can be derived for all lanes
// at once (span_par_walk below).  Conditions the derivation does not cover
// send the step to span_walk instead, which is exact everywhere:
//   * the window reaches the block's limit region (s_limit checks, done());
//   * 32 lanes in a row without a hit (the run-length rule, kSpanRun).
// ---------------------------------------------------------------------
// may this step take the fast walk?  n = block length, st.s = window base
// (written without branches: at five wavefronts per CU a taken scalar branch
// costs as much as a dozen ALU instructions)
SNAPMI_LANE_FN bool span_fast_ok(const SpanState &st, const uint64_t hits,
                                 const uint32_t n)
{
    // lanes 1..63 without a hit, as 63 bits: 32 in a row anywhere?
    const uint64_t z = (~hits) >> 1;
    uint64_t r = z & (z >> 1);
    r &= r >> 2;
    r &= r >> 4;
    r &= r >> 8;
    r &= r >> 16;
    // ... and the run the window may start in (bit 63 of ~z is set)
    const uint32_t lead = (uint32_t)__builtin_ctzll(~z);
    const bool run_ok = (st.chain != 0) | (st.q + lead < kSpanRun);
    // st.s + 93 <= n: every lane active, no limit check can fail, no copy
    // of fewer than 16 bytes can end the block
    return (st.s + 93 <= n) & (r == 0) & run_ok;
}

// ---------------------------------------------------------------------
// The lane-parallel walk (round 5).  Round 4's fast walk followed the chain
// of copies on the scalar unit and collected the lanes inside them as it went
// - a v_readlane, ~14 scalar instructions and a loop branch per copy, and the
// masks and the state behind it another ~150 scalar instructions: half of a
// step's time at five wavefronts per CU, where every instruction a lone
// wavefront issues costs 4-5 cycles (alice29.txt tiled to 1 GiB: 37.1 ms with
// it, 28.1 with the first version of this one).  But the chain is pointer
// chasing over per-lane successors: behind a copy at hit lane X (m bytes,
// ending inside the window) the next copy is at the first hit lane at or
// behind X + m - a find-first-set every hit lane does for itself - so the
// set of copies the reference emits is the orbit of the window's first hit
// under that map: one v_readlane per copy, five instructions a hop, at most
// 16 in a window (a copy is 4 bytes or more).  Everything else is per-lane
// arithmetic on that orbit: the copy below a lane and where it ends (one
// ds_bpermute) say whether the lane lies inside a copy, is its last byte (the
// insert of e - 1, src/compress.rs:290-297), or is looked up; a token's
// literal starts where the copy below ended.  span_walk's cut (an inserted
// lane with a C bit whose lower lane is not inserted: one step in five on
// text) is a mask: the step ends in front of the lowest such lane.  Per step
// 170 scalar + 185 vector instructions and 31 branches where round 4 had 404 +
// 190 and 58 (profiles/r5_span_kernel_counters.txt).
//
// W is the wave: per-lane values W::u32 / flags W::b1 with the operators of
// uint32_t, and
//   lane()               0..63
//   sel(b, x, y)  lt / ge / eq(x, y)  band / bor / bnot
//   ballot(b)            uniform 64-bit mask
//   bperm(idx, v)        v of lane idx (0..63, any lane)
//   readlane(v, l)       v of the uniform lane l
//   bit(mask, i)         bit i (per lane, 0..63) of a uniform mask
//   next_bit(mask, t)    lowest set bit >= t (per lane, 0..63) or 64
//   prev_bit(mask, t)    highest set bit < t (per lane, 0..63) or 64
//   shr_lo(mask, i)      the low 32 bits of mask >> i (per lane, 0..63)
//   count_cut()          a hook for the host test's statistics
// snapmi_compress.hip instantiates it over the hardware, tests/
// span_wave_host.cpp over arrays of 64 - the same text - and
// test_span_wave_cpu.py compares it with span_walk on random windows and the
// streams it gives with the reference restatement's.
// ---------------------------------------------------------------------
#if defined(__HIPCC__)
#define SNAPMI_WAVE_FN __device__ __forceinline__
#else
#define SNAPMI_WAVE_FN inline
#endif

// span_fast_ok with its one expensive test - 32 lanes in a row without a hit,
// ten 64-bit scalar shifts and ANDs - done by the lanes: lane L of 1..32 looks
// at the 32 bits from its own (one 64-bit shift per lane, one ballot).
template <class W>
SNAPMI_WAVE_FN bool span_fast_ok_w(const W &w, const SpanState &st,
                                   const uint64_t hits, const uint32_t n)
{
    typedef typename W::u32 u32;
    const u32 lane = w.lane();
    const uint64_t run32 = w.ballot(
        w.band(w.band(w.ge(lane, u32(1)), w.lt(lane, u32(33))),
               w.eq(w.shr_lo(hits, lane), u32(0))));
    const uint64_t h1 = hits >> 1;
    const uint32_t lead = h1 ? (uint32_t)__builtin_ctzll(h1) : 64;
    const bool run_ok = (st.chain != 0) | (st.q + lead < kSpanRun);
    return (st.s + 93 <= n) & (run32 == 0) & run_ok;
}

// hits (lanes 1..63, at least one: span_fast_ok) as for span_walk;
// m, old, cbit: this lane's match length, exchanged entry and C bit.
// emit: in, the first byte not covered by a token; out, the same behind the
// step.  Out: vh = lanes whose copy is a token of this step, lit = a vh
// lane's literal length, touched = lanes the reference inserts, at = the lane
// of a long match (kSpanLong).
template <class W>
SNAPMI_WAVE_FN uint32_t
span_par_walk(const W &w, SpanState &st, const uint64_t hits,
              const typename W::u32 m,
              const typename W::u32 old, const typename W::b1 cbit,
              uint32_t &emit, uint64_t &vh, typename W::u32 &lit,
              uint64_t &touched, uint32_t &at)
{
    typedef typename W::u32 u32;
    typedef typename W::b1 b1;
    const uint32_t base = st.s, emit0 = emit;
    const u32 lane = w.lane();
    const b1 hit = w.bit(hits, lane);
    // a hit lane's copy ends in front of lane `end`; it is the last of the
    // step if it is long (the caller extends it) or reaches lane 63's insert
    const u32 end = lane + m;
    const b1 last = w.bor(w.ge(m, u32(16)), w.ge(end, u32(64)));
    const u32 nx = w.next_bit(hits, w.sel(w.lt(end, u32(63)), end, u32(63)));
    const b1 go = w.band(hit, w.band(w.bnot(last), w.lt(nx, u32(64))));
    // The copies of the step: the orbit of the first hit under J.  The scalar
    // unit hops through it - a v_readlane per copy whose wait states hold the
    // mask update and the exit test, ~20 cycles a hop, 9 hops on text - which
    // equals four rounds of doubling over ds_bpermute (R |= R[frontier]:
    // exact too, built first) on text and beats them by 5 % where the windows
    // hold fewer copies (HTML, URLs); for the small-block kernel, whose ten
    // wavefronts per CU keep the scalar unit 70 % busy, the doubling rounds
    // measured 2 % better at 8 KiB and nothing at 2-4 KiB: one method kept.
    // At most 16 copies of 4 bytes or more fit a window.
    const u32 J = w.sel(go, nx, u32(64));
    uint64_t V = 0;
    uint32_t cur = 1 + (uint32_t)__builtin_ctzll(hits >> 1);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < 16; i++) {
        V |= 1ull << cur;
        cur = w.readlane(J, cur);
        if (cur >= 64)
            break;
    }
    // the last copy of the step
    const uint32_t Xl = 63u - (uint32_t)__builtin_clzll(V);
    const uint32_t ml = w.readlane(m, Xl), el = Xl + ml;
    const bool is_long = ml >= 16, is_out = !is_long & (el >= 64);
    const uint32_t stop = is_long ? Xl + 1 : 64; // lanes 1 .. stop-1 walked
    // the copy below this lane and where it ends
    const u32 Xp = w.prev_bit(V, lane);
    const b1 has = w.lt(Xp, u32(64));
    const u32 eP = w.bperm(w.sel(has, Xp, lane), end);
    const b1 in_range =
        w.band(w.ge(lane, u32(1)), w.lt(lane, u32(stop)));
    const b1 inside = w.band(has, w.lt(lane, eP));
    const b1 looked = w.band(in_range, w.bnot(inside));
    const b1 ins = w.band(w.band(in_range, inside),
                          w.band(w.eq(lane + u32(1), eP),
                                 w.lt(lane, u32(63))));
    const b1 tb = w.bor(w.bor(looked, ins),
                        w.band(w.eq(lane, u32(0)), w.ge(u32(st.chain), u32(1))));
    const uint64_t T = w.ballot(tb);
    // an inserted lane with a C bit needs the lane it collided with inserted
    const u32 pred = (old - u32(base - 1)) & u32(63);
    const uint64_t bad =
        w.ballot(w.band(w.band(tb, cbit), w.bnot(w.bit(T, pred))));
    lit = w.sel(has, lane - eP, lane + u32(base - 1 - emit0));
    const uint64_t lbit = is_long ? 1ull << Xl : 0;
    const bool prev = (V & ((1ull << Xl) - 1)) != 0; // a copy below the last
    const uint32_t ePl = w.readlane(eP, Xl);
    if (bad == 0) {
        vh = V & ~lbit;
        touched = T;
        at = Xl;
        if (is_long) {
            emit = prev ? base - 1 + ePl : emit0;
            st.s = base - 1 + Xl;
            st.next_emit = emit;
            return kSpanLong;
        }
        const uint32_t pe = base - 1 + el; // where the last copy ends
        st.s = is_out ? pe : base + 63;
        st.q = is_out ? 0 : 63 - el;
        st.chain = is_out ? 1 : 0;
        st.next_emit = pe;
        emit = pe;
        return kSpanCont;
    }
    // The step ends in front of lane `cut` (span_walk's cut): what lies below
    // stands.  Lane `cut` is the insert behind a copy (the copy stands, insert
    // and check are the next step's), a chain check, or a probe of a run.
    const uint32_t cut = (uint32_t)__builtin_ctzll(bad); // >= 1
    w.count_cut(); // (a test's statistics; nothing on the device)
    const uint64_t below = (1ull << cut) - 1;
    vh = V & below;
    touched = T & below;
    at = 0;
    const bool has_c = vh != 0; // a copy below the cut, ending at lane e_c
    const uint32_t e_c = w.readlane(eP, cut);
    const bool c_ins = has_c & (cut < e_c);
    const bool c_chk =
        !c_ins & ((has_c & (e_c == cut)) | ((cut == 1) & (st.chain != 0)));
    const bool c_run = !c_ins & !c_chk;
    const uint32_t q_run =
        has_c ? cut - e_c - 1 : (st.chain ? cut - 2 : st.q + cut - 1);
    const uint32_t nemit =
        c_ins ? base + cut : (has_c ? base - 1 + e_c : emit0);
    st.s = c_ins ? base + cut : base - 1 + cut;
    st.q = c_run ? q_run : 0;
    st.chain = c_run ? 0 : 1;
    st.next_emit = nemit;
    emit = nemit;
    return kSpanCont;
}

} // namespace snapmi
#endif
