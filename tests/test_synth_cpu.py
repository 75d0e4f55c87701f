"""The cfg3 workload generator is reproducible: the vectorised
generator bench_configs.synth_text equals the one-token-at-a-time reference,
and the bytes are pinned by a checksum - independent of the torch version
(no torch RNG: counter-based splitmix64 in integer arithmetic)."""
import hashlib

import torch

import bench_configs as B


def test_synth_text_equals_the_sequential_reference():
    n = 1 << 20
    ref = B.synth_text_reference(n)
    got = B.synth_text(torch.device("cpu"), n).numpy().tobytes()
    assert got == ref
    assert len(ref) == n
    # the rule: a newline directly behind the first token that passes column 72
    for line in ref[:100000].split(b"\n")[:-1]:
        assert len(line) > 72, line
        assert len(line) - len(line.split(b" ")[-1]) - 1 <= 72 or \
            b" " not in line, line
    # a prefix is a prefix: periods of different length share their start
    assert B.synth_text(torch.device("cpu"), 1 << 18).numpy().tobytes() == \
        ref[:1 << 18]
    assert hashlib.sha256(ref).hexdigest() == SHA_1MIB


SHA_1MIB = "4246a5e5b9084fb3a47f8328a991d60e83cd83305a7ed0bc996ee0009d0c6f5c"
