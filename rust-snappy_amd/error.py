"""Mirror of snap::Error (reference src/error.rs:72-180)."""

KINDS = {
    0: ("Ok", ()),
    1: ("TooBig", ("given", "max")),
    2: ("BufferTooSmall", ("given", "min")),
    3: ("Empty", ()),
    4: ("Header", ()),
    5: ("HeaderMismatch", ("expected_len", "got_len")),
    6: ("Literal", ("len", "src_len", "dst_len")),
    7: ("CopyRead", ("len", "src_len")),
    8: ("CopyWrite", ("len", "dst_len")),
    9: ("Offset", ("offset", "dst_pos")),
    10: ("StreamHeader", ("byte",)),
    11: ("StreamHeaderMismatch", ("bytes",)),
    12: ("UnsupportedChunkType", ("byte",)),
    13: ("UnsupportedChunkLength", ("len", "header")),
    14: ("Checksum", ("expected", "got")),
    64: ("UnexpectedEof", ()),
    100: ("Device", ()),
    101: ("Argument", ()),
}


def _escape(raw):
    """std::ascii::escape_default over bytes."""
    out = []
    for b in raw:
        if b == 9:
            out.append("\\t")
        elif b == 13:
            out.append("\\r")
        elif b == 10:
            out.append("\\n")
        elif b in (39, 34, 92):
            out.append("\\" + chr(b))
        elif 0x20 <= b <= 0x7E:
            out.append(chr(b))
        else:
            out.append(f"\\x{b:02x}")
    return "".join(out)


class Error(Exception):
    """snap::Error: compares equal by variant and field values, like the
    reference's PartialEq (src/error.rs:190-245)."""

    def __init__(self, kind, a=0, b=0, c=0, message=None):
        self.kind = int(kind)
        self.variant, names = KINDS.get(self.kind, (f"Kind{kind}", ()))
        vals = (int(a), int(b), int(c))
        self.abc = vals  # the raw (a, b, c) of snapmi_error
        self.fields = dict(zip(names, vals))
        self.message = message
        super().__init__(self.display())

    def debug(self):
        """The reference's Debug form: `Variant { field: value, .. }`."""
        text = self.variant
        if self.fields:
            text += " { " + ", ".join(f"{k}: {v}"
                                      for k, v in self.fields.items()) + " }"
        return text

    def __repr__(self):
        return self.debug()

    def display(self):
        """impl fmt::Display for Error, reference src/error.rs:249-335: the
        text a user of the reference (szip: szip/main.rs:75-82) sees."""
        f = self.fields
        v = self.variant
        if v == "TooBig":
            t = (f"snappy: input buffer (size = {f['given']}) is larger than "
                 f"allowed (size = {f['max']})")
        elif v == "BufferTooSmall":
            t = (f"snappy: output buffer (size = {f['given']}) is smaller "
                 f"than required (size = {f['min']})")
        elif v == "Empty":
            t = "snappy: corrupt input (empty)"
        elif v == "Header":
            t = "snappy: corrupt input (invalid header)"
        elif v == "HeaderMismatch":
            t = ("snappy: corrupt input (header mismatch; expected "
                 f"{f['expected_len']} decompressed bytes but got "
                 f"{f['got_len']})")
        elif v == "Literal":
            t = ("snappy: corrupt input (expected literal read of length "
                 f"{f['len']}; remaining src: {f['src_len']}; remaining dst: "
                 f"{f['dst_len']})")
        elif v == "CopyRead":
            t = ("snappy: corrupt input (expected copy read of length "
                 f"{f['len']}; remaining src: {f['src_len']})")
        elif v == "CopyWrite":
            t = ("snappy: corrupt input (expected copy write of length "
                 f"{f['len']}; remaining dst: {f['dst_len']})")
        elif v == "Offset":
            t = ("snappy: corrupt input (expected valid offset but got "
                 f"offset {f['offset']}; dst position: {f['dst_pos']})")
        elif v == "StreamHeader":
            t = ("snappy: corrupt input (expected stream header but got "
                 f"unexpected chunk type byte {f['byte']})")
        elif v == "StreamHeaderMismatch":
            # (the ABI carries the six bytes little-endian in one field;
            # std::ascii::escape_default, src/error.rs:337-340)
            raw = int(f["bytes"]).to_bytes(8, "little")[:6]
            t = ("snappy: corrupt input (expected sNaPpY stream header but "
                 f"got {_escape(raw)})")
        elif v == "UnsupportedChunkType":
            t = ("snappy: corrupt input (unsupported chunk type: "
                 f"{f['byte']})")
        elif v == "UnsupportedChunkLength":
            t = ("snappy: corrupt input (invalid stream header length: "
                 f"{f['len']})" if f["header"] else
                 f"snappy: corrupt input (unsupported chunk length: "
                 f"{f['len']})")
        elif v == "Checksum":
            t = ("snappy: corrupt input (bad checksum; expected: "
                 f"{f['expected']}, got: {f['got']})")
        else:  # not a snap::Error: UnexpectedEof (io), Device, Argument
            t = self.debug()
        if self.message:
            t += f": {self.message}"
        return t

    def key(self):
        return (self.variant,) + tuple(self.fields.values())

    def __eq__(self, other):
        return isinstance(other, Error) and self.key() == other.key()

    def __hash__(self):
        return hash(self.key())


class DeviceError(Error):
    """No usable GPU / HIP failure.  Never swallowed, never worked around."""
