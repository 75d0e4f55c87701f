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
        text = self.variant
        if self.fields:
            text += " { " + ", ".join(f"{k}: {v}"
                                      for k, v in self.fields.items()) + " }"
        if message:
            text += f": {message}"
        super().__init__(text)

    def key(self):
        return (self.variant,) + tuple(self.fields.values())

    def __eq__(self, other):
        return isinstance(other, Error) and self.key() == other.key()

    def __hash__(self):
        return hash(self.key())


class DeviceError(Error):
    """No usable GPU / HIP failure.  Never swallowed, never worked around."""
