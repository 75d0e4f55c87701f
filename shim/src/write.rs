//! `snap::write::FrameEncoder` (reference src/write.rs).
//!
//! The reference compresses one <= 64 KiB chunk per call of `Inner::write`.
//! Here the same state machine decides where chunks begin and end - so the
//! stream is byte for byte the reference's - but the chunks are queued and
//! handed to the GPU `BATCH` bytes at a time (`snapmi_frame_encode_host`:
//! CRC32C, raw compression and framing of all queued chunks in one launch
//! sequence).  `flush()` compresses what is queued.
use std::fmt;
use std::io::{self, Write};

pub use crate::error::IntoInnerError;
use crate::gpu::{self, Context, Failure, SnapmiError};
use crate::MAX_BLOCK_SIZE;

/// Bytes queued before the device is called (bounded memory).
const BATCH: usize = 64 << 20;

/// Compresses what is written to it into the Snappy frame format and writes
/// that to `W` (reference src/write.rs:19-50).  Flushed on drop, errors of
/// that flush ignored, like the reference.
pub struct FrameEncoder<W: io::Write> {
    inner: Option<Inner<W>>,
    /// The reference's `src`: at most one block of not yet emitted bytes.
    src: Vec<u8>,
}

struct Inner<W> {
    w: W,
    ctx: Context,
    /// Chunks cut but not yet compressed, back to back, and their lengths.
    queue: Vec<u8>,
    lens: Vec<u32>,
    /// Framed output of one batch.
    dst: Vec<u8>,
    wrote_stream_ident: bool,
}

impl<W: io::Write> FrameEncoder<W> {
    /// A new streaming compressor writing to `wtr`.
    pub fn new(wtr: W) -> FrameEncoder<W> {
        FrameEncoder {
            inner: Some(Inner {
                w: wtr,
                ctx: Context::new(),
                queue: Vec::new(),
                lens: Vec::new(),
                dst: Vec::new(),
                wrote_stream_ident: false,
            }),
            src: Vec::with_capacity(MAX_BLOCK_SIZE),
        }
    }

    /// Flushes and returns the writer (reference :87-97).
    pub fn into_inner(mut self) -> Result<W, IntoInnerError<FrameEncoder<W>>> {
        match self.flush() {
            Ok(()) => Ok(self.inner.take().unwrap().w),
            Err(err) => Err(IntoInnerError::new(self, err)),
        }
    }

    /// The underlying writer.
    pub fn get_ref(&self) -> &W {
        &self.inner.as_ref().unwrap().w
    }

    /// The underlying writer, mutably (writing to it corrupts the stream).
    pub fn get_mut(&mut self) -> &mut W {
        &mut self.inner.as_mut().unwrap().w
    }
}

impl<W: io::Write> Drop for FrameEncoder<W> {
    fn drop(&mut self) {
        if self.inner.is_some() {
            let _ = self.flush(); // reference :112-120
        }
    }
}

impl<W: io::Write> io::Write for FrameEncoder<W> {
    // reference :123-152: fill `src`; a write that does not fit flushes the
    // buffer, or - when the buffer is empty - goes out directly, partial
    // tail chunk included
    fn write(&mut self, mut buf: &[u8]) -> io::Result<usize> {
        let mut total = 0;
        loop {
            let free = MAX_BLOCK_SIZE - self.src.len();
            let n = if buf.len() <= free {
                break;
            } else if self.src.is_empty() {
                self.inner.as_mut().unwrap().cut(buf)?
            } else {
                self.src.extend_from_slice(&buf[..free]);
                self.flush_src()?;
                free
            };
            buf = &buf[n..];
            total += n;
        }
        self.src.extend_from_slice(buf);
        Ok(total + buf.len())
    }

    // reference :154-161, plus: everything queued is compressed and written
    fn flush(&mut self) -> io::Result<()> {
        self.flush_src()?;
        self.inner.as_mut().unwrap().emit()
    }
}

impl<W: io::Write> FrameEncoder<W> {
    fn flush_src(&mut self) -> io::Result<()> {
        if !self.src.is_empty() {
            self.inner.as_mut().unwrap().cut(&self.src)?;
            self.src.clear();
        }
        Ok(())
    }
}

impl<W: io::Write> Inner<W> {
    /// reference `Inner::write` (:171-190): `buf` becomes chunks of at most
    /// 65536 bytes; they are compressed when a batch is full.
    fn cut(&mut self, buf: &[u8]) -> io::Result<usize> {
        for c in buf.chunks(MAX_BLOCK_SIZE) {
            self.queue.extend_from_slice(c);
            self.lens.push(c.len() as u32);
        }
        if self.queue.len() >= BATCH {
            self.emit()?;
        }
        Ok(buf.len())
    }

    fn emit(&mut self) -> io::Result<()> {
        if self.lens.is_empty() {
            return Ok(());
        }
        let cap = unsafe { gpu::snapmi_frame_encode_bound(self.queue.len(), self.lens.len()) };
        self.dst.resize(cap, 0);
        let flags = if self.wrote_stream_ident { gpu::SNAPMI_FRAME_NO_IDENT } else { 0 };
        let mut written = 0usize;
        let rc = unsafe {
            gpu::snapmi_frame_encode_host(
                self.ctx.as_ptr(), self.queue.as_ptr(), self.lens.as_ptr(), self.lens.len(),
                flags, self.dst.as_mut_ptr(), cap, &mut written,
            )
        };
        self.queue.clear();
        self.lens.clear();
        if rc != 0 {
            let e = SnapmiError::default();
            return Err(match gpu::to_failure(rc, &e, Some(&self.ctx)) {
                Failure::Snap(e) => io::Error::from(e),
                Failure::UnexpectedEof => io::ErrorKind::UnexpectedEof.into(),
                Failure::Device(msg) => io::Error::new(io::ErrorKind::Other, msg),
            });
        }
        self.wrote_stream_ident = true; // identifier once per stream (:167-170)
        self.w.write_all(&self.dst[..written])
    }
}

impl<W: fmt::Debug + io::Write> fmt::Debug for FrameEncoder<W> {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        f.debug_struct("FrameEncoder")
            .field("w", &self.inner.as_ref().map(|i| &i.w))
            .field("src", &"[...]")
            .finish()
    }
}
