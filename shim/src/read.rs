//! `snap::read::FrameDecoder` and `snap::read::FrameEncoder` (reference
//! src/read.rs).
//!
//! `FrameDecoder` takes what the reader has - up to `BATCH` bytes, but never
//! waiting for more once a read came back short - lets
//! `snapmi_frame_decode_host` decode every complete chunk in it in one device
//! call (header checks, raw decode, CRC, all in the reference's order) and
//! keeps the cut-off tail for the next round.  Bytes of the chunks in front of
//! a bad chunk are returned before the error (reference :111-118).
//!
//! Memory, as in rust-snappy_amd/frame.py (the tested mirror of this file):
//! input and output of a batch are staged in page-locked memory
//! (`gpu::PinnedBuf`) so that the copies of the device call are asynchronous;
//! the output room is sized by a host scan of the chunk headers
//! (`snapmi_frame_scan_host`: 65536 bytes per data chunk of the batch), and a
//! caller's buffer that has that much room gets the decoded bytes straight
//! from the device call.
use std::cmp;
use std::fmt;
use std::io::{self, Read};

use crate::gpu::{self, Context, Failure, PinnedBuf, SnapmiError};
use crate::MAX_BLOCK_SIZE;

/// Most that one device call is handed.  Buffers start small and grow towards
/// it only while the reader keeps them full.
const BATCH: usize = 64 << 20;
const FIRST: usize = 1 << 20;

/// Decompresses a Snappy frame stream while it is read (reference :37-101).
pub struct FrameDecoder<R: io::Read> {
    r: R,
    ctx: Context,
    /// Compressed bytes read but not decoded yet (src[..srce]), pinned.
    src: PinnedBuf,
    srce: usize,
    /// Decoded bytes not yet handed out: dst[dsts..dste], pinned.
    dst: PinnedBuf,
    dsts: usize,
    dste: usize,
    /// An error that follows the bytes in `dst`.
    pending: Option<io::Error>,
    eof: bool,
    read_stream_ident: bool,
    /// First 10 bytes of the reference reader's scratch buffer (the
    /// truncated-varint rule of reference :216; see include/snapmi.h).
    stale: [u8; 10],
}

impl<R: io::Read> FrameDecoder<R> {
    /// A new streaming decompressor reading from `rdr`.
    pub fn new(rdr: R) -> FrameDecoder<R> {
        FrameDecoder {
            r: rdr,
            ctx: Context::new(),
            src: PinnedBuf::with_len(FIRST),
            srce: 0,
            dst: PinnedBuf::new(),
            dsts: 0,
            dste: 0,
            pending: None,
            eof: false,
            read_stream_ident: false,
            stale: [0; 10],
        }
    }

    /// The underlying reader.
    pub fn get_ref(&self) -> &R {
        &self.r
    }

    /// The underlying reader, mutably.
    pub fn get_mut(&mut self) -> &mut R {
        &mut self.r
    }

    /// The underlying reader; undecoded input is dropped.
    pub fn into_inner(self) -> R {
        self.r
    }

    /// One batch: read, decode the whole chunks, keep the tail.
    ///
    /// The reference reads one chunk per call (:105-172).  Batching must not
    /// turn into waiting: one inner read at least, more only while the reader
    /// fills what it is offered - a short read means it has no more right now
    /// (a pipe, a socket, a peer that waits for our answer before it sends
    /// on).  A read error leaves `src[..srce]` as it is: nothing is lost, the
    /// next call carries on.
    ///
    /// `direct`: the caller's buffer.  When it has room for every data chunk
    /// of the batch the bytes are decoded straight into it and their number
    /// is returned; otherwise they wait in `dst` (returns 0).
    fn fill(&mut self, direct: &mut [u8]) -> io::Result<usize> {
        loop {
            let mut full = true;
            while self.srce < self.src.len() && !self.eof {
                let ask = self.src.len() - self.srce;
                let n = match self.r.read(&mut self.src[self.srce..]) {
                    Ok(n) => n,
                    Err(ref e) if e.kind() == io::ErrorKind::Interrupted => continue,
                    Err(e) => return Err(e),
                };
                if n == 0 {
                    self.eof = true;
                }
                self.srce += n;
                if n < ask {
                    full = false;
                    break;
                }
            }
            if full && !self.eof && self.src.len() < BATCH {
                // the reader keeps up: a larger batch next time
                let len = cmp::min(2 * self.src.len(), BATCH);
                self.src.resize(len);
            }
            if self.srce == 0 {
                return Ok(0); // clean end of the stream
            }
            let mut flags = 0;
            if self.read_stream_ident {
                flags |= gpu::SNAPMI_FRAME_CONTINUATION;
            }
            if self.eof {
                flags |= gpu::SNAPMI_FRAME_FINAL;
            }
            // room for every data chunk of the batch: the host scan counts
            // them (a chunk yields at most 65536 bytes)
            let (mut nd, mut scanned) = (0u64, 0u64);
            unsafe {
                gpu::snapmi_frame_scan_host(
                    self.src.as_ptr() as *const _, self.srce as u64,
                    flags & gpu::SNAPMI_FRAME_CONTINUATION, std::ptr::null_mut(),
                    std::ptr::null_mut(), 0, &mut nd, &mut scanned,
                );
            }
            let room = cmp::max(nd as usize, 1) * MAX_BLOCK_SIZE;
            let mine = direct.len() < room;
            if mine {
                self.dst.resize(room);
            }
            let (out, out_cap) = if mine {
                (self.dst.as_mut_ptr(), room)
            } else {
                (direct.as_mut_ptr(), room)
            };
            let (mut written, mut consumed) = (0usize, 0usize);
            let mut e = SnapmiError::default();
            let rc = unsafe {
                gpu::snapmi_frame_decode_host(
                    self.ctx.as_ptr(), self.src.as_ptr(), self.srce, flags,
                    self.stale.as_mut_ptr(), out, out_cap,
                    &mut written, &mut consumed, &mut e,
                )
            };
            self.dsts = 0;
            self.dste = if mine { written } else { 0 };
            let handed = if mine { 0 } else { written };
            if rc != 0 {
                // the good bytes first, the error by the read that reaches it
                self.pending = Some(match gpu::to_failure(rc, &e, Some(&self.ctx)) {
                    Failure::Snap(e) => io::Error::from(e),
                    Failure::UnexpectedEof => io::ErrorKind::UnexpectedEof.into(),
                    Failure::Device(msg) => io::Error::new(io::ErrorKind::Other, msg),
                });
                self.srce = 0;
                return Ok(handed);
            }
            if consumed == 0 {
                // not one whole chunk in the buffer (a chunk is < 76 KiB, so
                // only at the very start): read more
                if self.eof {
                    return Err(io::Error::new(io::ErrorKind::Other, "snapmi: no progress"));
                }
                if self.srce == self.src.len() {
                    let len = 2 * self.src.len();
                    self.src.resize(len);
                }
                continue;
            }
            self.read_stream_ident = true;
            self.src.copy_within(consumed..self.srce, 0);
            self.srce -= consumed;
            return Ok(handed);
        }
    }
}

impl<R: io::Read> io::Read for FrameDecoder<R> {
    fn read(&mut self, buf: &mut [u8]) -> io::Result<usize> {
        loop {
            if self.dsts < self.dste {
                let len = cmp::min(self.dste - self.dsts, buf.len());
                buf[..len].copy_from_slice(&self.dst[self.dsts..self.dsts + len]);
                self.dsts += len;
                return Ok(len);
            }
            if let Some(err) = self.pending.take() {
                return Err(err);
            }
            if self.eof && self.srce == 0 {
                return Ok(0);
            }
            let n = self.fill(buf)?;
            if n > 0 {
                return Ok(n); // decoded straight into `buf`
            }
            if self.dsts == self.dste && self.pending.is_none() && self.eof && self.srce == 0 {
                return Ok(0);
            }
        }
    }
}

impl<R: fmt::Debug + io::Read> fmt::Debug for FrameDecoder<R> {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        f.debug_struct("FrameDecoder")
            .field("r", &self.r)
            .field("src", &"[...]")
            .field("dst", &"[...]")
            .field("dsts", &self.dsts)
            .field("dste", &self.dste)
            .field("read_stream_ident", &self.read_stream_ident)
            .finish()
    }
}

/// Compresses what is read from `R` into the Snappy frame format (reference
/// :254-409).  As in the reference every chunk is what ONE read of up to
/// 65536 bytes from `R` returned (:378); up to `BATCH` bytes of such reads
/// are compressed by one device call.
pub struct FrameEncoder<R: io::Read> {
    r: R,
    ctx: Context,
    src: PinnedBuf,
    lens: Vec<u32>,
    dst: PinnedBuf,
    dsts: usize,
    dste: usize,
    eof: bool,
    wrote_stream_ident: bool,
    /// A reader's error that follows the bytes in `dst`.
    pending: Option<io::Error>,
}

impl<R: io::Read> FrameEncoder<R> {
    /// A new streaming compressor reading from `rdr`.
    pub fn new(rdr: R) -> FrameEncoder<R> {
        FrameEncoder {
            r: rdr,
            ctx: Context::new(),
            src: PinnedBuf::new(),
            lens: Vec::new(),
            dst: PinnedBuf::new(),
            dsts: 0,
            dste: 0,
            eof: false,
            wrote_stream_ident: false,
            pending: None,
        }
    }

    /// The underlying reader.
    pub fn get_ref(&self) -> &R {
        &self.r
    }

    /// The underlying reader, mutably.
    pub fn get_mut(&mut self) -> &mut R {
        &mut self.r
    }

    /// Up to `BATCH` bytes of inner reads -> one device call.  A read that
    /// fails does not lose the chunks gathered before it: they are compressed
    /// and handed out, the error is returned by the read that follows them
    /// (the reference does one inner read per outer read, :378, so an error
    /// there loses nothing either).  A short read ends the batch.
    fn fill(&mut self) -> io::Result<()> {
        self.src.clear();
        self.lens.clear();
        while self.src.len() < BATCH {
            let at = self.src.len();
            self.src.resize(at + MAX_BLOCK_SIZE);
            let n = match self.r.read(&mut self.src[at..]) {
                Ok(n) => n,
                Err(ref e) if e.kind() == io::ErrorKind::Interrupted => {
                    self.src.resize(at);
                    continue;
                }
                Err(e) => {
                    self.src.resize(at);
                    if self.lens.is_empty() {
                        return Err(e);
                    }
                    self.pending = Some(e);
                    break;
                }
            };
            self.src.resize(at + n);
            if n == 0 {
                self.eof = true;
                break;
            }
            self.lens.push(n as u32);
            if n < MAX_BLOCK_SIZE {
                break;
            }
        }
        self.dsts = 0;
        self.dste = 0;
        if self.lens.is_empty() {
            return Ok(());
        }
        let cap = unsafe { gpu::snapmi_frame_encode_bound(self.src.len(), self.lens.len()) };
        self.dst.resize(cap);
        let flags = if self.wrote_stream_ident { gpu::SNAPMI_FRAME_NO_IDENT } else { 0 };
        let mut written = 0usize;
        let rc = unsafe {
            gpu::snapmi_frame_encode_host(
                self.ctx.as_ptr(), self.src.as_ptr(), self.lens.as_ptr(), self.lens.len(), flags,
                self.dst.as_mut_ptr(), cap, &mut written,
            )
        };
        if rc != 0 {
            let e = SnapmiError::default();
            return Err(match gpu::to_failure(rc, &e, Some(&self.ctx)) {
                Failure::Snap(e) => io::Error::from(e),
                Failure::UnexpectedEof => io::ErrorKind::UnexpectedEof.into(),
                Failure::Device(msg) => io::Error::new(io::ErrorKind::Other, msg),
            });
        }
        self.wrote_stream_ident = true;
        self.dste = written;
        Ok(())
    }
}

impl<R: io::Read> io::Read for FrameEncoder<R> {
    fn read(&mut self, buf: &mut [u8]) -> io::Result<usize> {
        loop {
            if self.dsts < self.dste {
                let len = cmp::min(self.dste - self.dsts, buf.len());
                buf[..len].copy_from_slice(&self.dst[self.dsts..self.dsts + len]);
                self.dsts += len;
                return Ok(len);
            }
            if let Some(err) = self.pending.take() {
                return Err(err);
            }
            if self.eof {
                return Ok(0);
            }
            self.fill()?;
        }
    }
}

impl<R: fmt::Debug + io::Read> fmt::Debug for FrameEncoder<R> {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        f.debug_struct("FrameEncoder")
            .field("r", &self.r)
            .field("dst", &"[...]")
            .field("dsts", &self.dsts)
            .field("dste", &self.dste)
            .finish()
    }
}
