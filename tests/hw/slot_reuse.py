#!/usr/bin/env python3
"""How often would a small per-lane buffer of pending table inserts save an
HBM transaction in the lane-per-block match finder?  

Replays the reference's table access sequence (src/compress.rs:207-245,
290-313) per 64 KiB block of the 12 bench inputs and measures, for every
table access, how many accesses ago the same slot was touched.  An access
whose slot was touched within the last K accesses could be served from a
K-entry buffer (no read; the buffered write is replaced, not issued).  CPU
only; prints one line per input and the corpus total."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as O

KS = (1, 2, 4, 8, 16, 64)


def block_accesses(src):
    n = len(src)
    if n < 17:
        return []
    shift, size = 24, 256
    while size < 16384 and size < n:
        shift -= 1
        size *= 2
    le32 = lambda p: int.from_bytes(src[p:p + 4], "little")
    h = lambda x: ((x * 0x1E35A7BD) & 0xFFFFFFFF) >> shift
    table = {}
    acc = []            # (slot, kind) kind 0 = probe, 1 = insert s-1, 2 = chain lookup
    s, s_limit = 1, n - 15
    while True:
        skip, s_next = 32, s
        hit = False
        while True:
            s = s_next
            step = skip >> 5
            s_next = s + step
            skip += step
            if s_next > s_limit:
                return acc
            slot = h(le32(s))
            cand = table.get(slot, 0)
            table[slot] = s
            acc.append((slot, 0))
            if le32(s) == le32(cand):
                break
        while True:
            base, c = s, cand + 4
            s += 4
            while s < n and src[s] == src[c]:
                s += 1
                c += 1
            if s >= s_limit:
                return acc
            slot1 = h(le32(s - 1))
            table[slot1] = s - 1
            acc.append((slot1, 1))
            slot = h(le32(s))
            cand = table.get(slot, 0)
            table[slot] = s
            acc.append((slot, 2))
            if le32(s) != le32(cand):
                s_next = s + 1
                s = s_next
                break
        # fall back into the probe loop at s (= old s + 1)
        s_next = s


def main():
    tot = [0] * len(KS)
    tot_n = 0
    for name, data in O.corpus_round():
        hits = [0] * len(KS)
        n_acc = 0
        for b in range(0, min(len(data), 4 * 65536), 65536):
            acc = block_accesses(data[b:b + 65536])
            last = {}
            for t, (slot, kind) in enumerate(acc):
                if slot in last:
                    dist = t - last[slot]
                    for i, k in enumerate(KS):
                        if dist <= k:
                            hits[i] += 1
                last[slot] = t
            n_acc += len(acc)
        if n_acc == 0:
            continue
        print(f"{name:18s} accesses {n_acc:7d}  re-touched within K accesses: "
              + "  ".join(f"K={k}: {100 * x / n_acc:5.2f}%" for k, x in zip(KS, hits)))
        tot = [a + b for a, b in zip(tot, hits)]
        tot_n += n_acc
    print(f"{'corpus':18s} accesses {tot_n:7d}  re-touched within K accesses: "
          + "  ".join(f"K={k}: {100 * x / tot_n:5.2f}%" for k, x in zip(KS, tot)))


if __name__ == "__main__":
    main()
