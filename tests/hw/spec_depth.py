"""(CPU) Rounds of the lane-per-block match finder per 64 KiB block if up to K
probes of a run of misses are resolved in one round (K = 1: k_match_blocks,
K = 2: k_match_blocks_spec; positions at most `reach` bytes behind the round's
first probe, because their bytes must be in the registers the window was read
into).  Chain rounds that hit and match extensions cost what they cost today.
Checked (to a round or two: the block's last rounds are counted roughly) against
tests/model_match_lane.py for K = 1, 2."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import model_match_lane as M  # noqa: E402
import oracle_lib as O  # noqa: E402


def rounds(block, K, reach):
    n = len(block)
    shift, tsize = 24, 256
    while tsize < 16384 and tsize < n:
        shift -= 1
        tsize *= 2
    s_limit = n - 15

    def h(i):
        return ((int.from_bytes(block[i:i + 4], "little") * 0x1E35A7BD)
                & 0xFFFFFFFF) >> shift

    def grouped(probes):
        r = i = 0
        while i < len(probes):
            j = i + 1
            while j < len(probes) and j - i < K and \
                    probes[j] - probes[i] <= reach:
                j += 1
            r += 1
            i = j
        return r

    table, R = {}, 0
    s, probes, skip, s_next = 1, [], 32, 1
    while True:
        while True:
            s = s_next
            step = skip >> 5
            s_next = s + step
            skip += step
            if s_next > s_limit:
                return R + grouped(probes)
            cand = table.get(h(s), 0)
            table[h(s)] = s
            probes.append(s)
            if block[s:s + 4] == block[cand:cand + 4]:
                break
        R += grouped(probes)
        probes = []
        while True:
            base, m = s, 4
            while base + m < n and block[base + m] == block[cand + m]:
                m += 1
            if m >= 12:
                R += (m - 12) // 16 + 1
            s = base + m
            if s >= s_limit:
                return R
            table[h(s - 1)] = s - 1
            cand = table.get(h(s), 0)
            table[h(s)] = s
            if block[s:s + 4] != block[cand:cand + 4]:
                probes, skip, s_next = [s], 32, s + 1
                break
            R += 1


if __name__ == "__main__":
    print("file: rounds per 64 KiB block at (K, reach) = (1,0) (2,3) (3,3) "
          "(4,3) (8,15)")
    for name in ("alice29.txt", "asyoulik.txt", "lcet10.txt", "plrabn12.txt",
                 "urls.10K", "html", "geo.protodata", "kppkn.gtb",
                 "paper-100k.pdf", "fireworks.jpeg"):
        d = (O.CORPUS / name).read_bytes()[:65536]
        row = [rounds(d, K, reach) for K, reach in
               ((1, 0), (2, 3), (3, 3), (4, 3), (8, 15))]
        _, m1 = M.compress_one_block_stream(d, False)
        _, m2 = M.compress_one_block_stream(d, True)
        assert abs(m1 - row[0]) <= 2 and abs(m2 - row[1]) <= 2, (name, m1, m2)
        print(f"{name:16s}", *row,
              " ratio K=2:", round(row[1] / row[0], 2),
              " K=4:", round(row[3] / row[0], 2))
