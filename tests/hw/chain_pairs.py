#!/usr/bin/env python3
"""Could the lane-per-block match finder save DRAM transactions on its CHAIN
rounds?    A chain round (after a copy: insert s-1, look
up s, insert s; reference src/compress.rs:290-313) issues one table read and
two table writes.  Two ideas, counted on the reference's own access sequence
(tests/hw/slot_reuse.py replays it; CPU only):

 (a) hold the s-1 insert in a register and retire it together with the next
     write when both entries lie in one 32-byte sector (two 16-byte entries:
     slots that differ in their lowest bit only) - how often is that?
 (b) a victim buffer of the last two written slots in VGPRs: how often does a
     table READ hit one of them (so that the read, a transaction of its own,
     is not issued)?"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import oracle_lib as O
from slot_reuse import block_accesses


def main():
    T = dict(acc=0, writes=0, chain=0, same_sector=0, same_slot=0,
             reads=0, victim2=0, next_sector=0)
    for name, data in O.corpus_round():
        t = dict.fromkeys(T, 0)
        for b in range(0, min(len(data), 4 * 65536), 65536):
            acc = block_accesses(data[b:b + 65536])
            written = []          # the last slots written, newest last
            prev = None
            for slot, kind in acc:
                t["acc"] += 1
                if kind != 1:     # probe or chain lookup: a read (+ a write)
                    t["reads"] += 1
                    if slot in written[-2:]:
                        t["victim2"] += 1
                t["writes"] += 1
                if kind == 2:     # (prev is the s-1 insert of this round)
                    t["chain"] += 1
                    if prev[0] == slot:
                        t["same_slot"] += 1
                    elif prev[0] >> 1 == slot >> 1:
                        t["same_sector"] += 1
                if prev is not None and prev[0] >> 1 == slot >> 1 \
                        and prev[0] != slot:
                    t["next_sector"] += 1
                written.append(slot)
                prev = (slot, kind)
        if t["acc"]:
            print(f"{name:18s} writes {t['writes']:7d}  chain rounds {t['chain']:6d}: "
                  f"s-1 and s in one slot {100*t['same_slot']/max(t['chain'],1):5.2f}%, "
                  f"in one 32-byte sector {100*t['same_sector']/max(t['chain'],1):5.2f}%  | "
                  f"reads {t['reads']:7d}: hit one of the last 2 written slots "
                  f"{100*t['victim2']/t['reads']:5.2f}%")
        for k in T:
            T[k] += t[k]
    saved_a = T["same_sector"] + T["same_slot"]
    print(f"{'corpus':18s} writes {T['writes']}, chain rounds {T['chain']}: "
          f"(a) saves {saved_a} writes = {100*saved_a/T['writes']:.2f}% of all "
          f"table writes; (b) saves {T['victim2']} reads = "
          f"{100*T['victim2']/T['reads']:.2f}% of all table reads; consecutive "
          f"writes (any kind) in one sector: {100*T['next_sector']/T['writes']:.2f}%")


if __name__ == "__main__":
    main()
