//! `snap::raw`: the raw block codec (reference src/compress.rs,
//! src/decompress.rs) over the scalar entry points of libsnapmi.so.
//!
//! Each call stages its buffers through the context (H2D, kernels, D2H).  The
//! throughput numbers of the GPU codec are reached with batches of streams
//! (`snapmi_compress_batch` / the frame types of this crate), not with one
//! 100 KB call at a time - see INTEGRATION.md section 1 for the latency of
//! these calls.
use std::io;

use crate::error::{Error, Result};
use crate::gpu::{self, Context, Failure, SnapmiError};
use crate::MAX_INPUT_SIZE;

/// Upper bound of the compressed size of `input_len` bytes, 0 if the input
/// is too large (reference src/compress.rs:42-53).
pub fn max_compress_len(input_len: usize) -> usize {
    unsafe { gpu::snapmi_max_compress_len(input_len) }
}

/// Decompressed length announced by a raw stream (reference
/// src/decompress.rs:30-35): `Ok(0)` for empty input.
pub fn decompress_len(input: &[u8]) -> Result<usize> {
    let mut n = 0usize;
    let mut e = SnapmiError::default();
    let rc = unsafe { gpu::snapmi_decompress_len(input.as_ptr(), input.len(), &mut n, &mut e) };
    match rc {
        0 => Ok(n),
        k => Err(expect_snap(gpu::to_failure(k, &e, None))),
    }
}

fn expect_snap(f: Failure) -> Error {
    match f {
        Failure::Snap(e) => e,
        // The raw API has no error channel for "the GPU went away"; the
        // reference's functions cannot fail that way.  Loud, never silent.
        Failure::UnexpectedEof => unreachable!("raw codec reported a reader EOF"),
        Failure::Device(msg) => panic!("snap (MI355X): {}", msg),
    }
}

/// Raw encoder (reference src/compress.rs:55-170).  Owns its device scratch
/// and stream like the reference's encoder owns its hash tables: reuse it.
pub struct Encoder {
    ctx: Context,
}

impl Encoder {
    /// A new encoder on this thread's GPU.
    pub fn new() -> Encoder {
        Encoder { ctx: Context::new() }
    }

    /// Not in the reference: allocates, now, the device memory that batches
    /// of up to `blocks` 64 KiB blocks will need (`snapmi_ctx_prepare`) - what
    /// `Encoder::new` does for the reference's 34 KiB of tables
    /// (src/compress.rs:80-82) is gigabytes here, so it is a call of its own.
    /// `top_of_memory`: see `SNAPMI_PREPARE_TOP_OF_MEMORY` in include/snapmi.h
    /// (for a process that owns the GPU; seizes it for a moment).
    pub fn prepare(&mut self, blocks: u64, top_of_memory: bool) -> io::Result<()> {
        let flags = if top_of_memory { gpu::SNAPMI_PREPARE_TOP_OF_MEMORY } else { 0 };
        match unsafe { gpu::snapmi_ctx_prepare(self.ctx.as_ptr(), blocks, flags) } {
            0 => Ok(()),
            k => Err(io::Error::new(
                io::ErrorKind::Other,
                format!("snapmi_ctx_prepare: {} ({})", k, self.ctx.last_error()),
            )),
        }
    }

    /// Not in the reference: what the context holds and what its last batch
    /// did, by name (`snapmi_ctx_get_info`: "scratch_bytes",
    /// "token_blocks_spilled", ... - include/snapmi.h).
    pub fn info(&mut self, name: &str) -> io::Result<i64> {
        let c = std::ffi::CString::new(name)
            .map_err(|e| io::Error::new(io::ErrorKind::InvalidInput, e))?;
        let mut v = 0i64;
        match unsafe { gpu::snapmi_ctx_get_info(self.ctx.as_ptr(), c.as_ptr(), &mut v) } {
            0 => Ok(v),
            k => Err(io::Error::new(
                io::ErrorKind::Other,
                format!("snapmi_ctx_get_info: {} ({})", k, self.ctx.last_error()),
            )),
        }
    }

    /// Compresses `input` into `output` (which must hold
    /// `max_compress_len(input.len())` bytes); returns the bytes written.
    pub fn compress(&mut self, input: &[u8], output: &mut [u8]) -> Result<usize> {
        let mut n = 0usize;
        let mut e = SnapmiError::default();
        let rc = unsafe {
            gpu::snapmi_raw_compress(
                self.ctx.as_ptr(), input.as_ptr(), input.len(), output.as_mut_ptr(),
                output.len(), &mut n, &mut e,
            )
        };
        match rc {
            0 => Ok(n),
            k => Err(expect_snap(gpu::to_failure(k, &e, Some(&self.ctx)))),
        }
    }

    /// Compresses into a fresh vector (reference :164-169).
    pub fn compress_vec(&mut self, input: &[u8]) -> Result<Vec<u8>> {
        let cap = max_compress_len(input.len());
        if cap == 0 && !input.is_empty() {
            return Err(Error::TooBig { given: input.len() as u64, max: MAX_INPUT_SIZE });
        }
        let mut buf = vec![0; cap.max(1)];
        let n = self.compress(input, &mut buf)?;
        buf.truncate(n);
        Ok(buf)
    }
}

impl Default for Encoder {
    fn default() -> Encoder {
        Encoder::new()
    }
}

impl Clone for Encoder {
    fn clone(&self) -> Encoder {
        Encoder::new() // scratch is not shared, as in the reference
    }
}

impl std::fmt::Debug for Encoder {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("Encoder").finish()
    }
}

/// Raw decoder (reference src/decompress.rs:37-111): stateless.
#[derive(Clone, Debug, Default)]
pub struct Decoder {
    _dummy: (),
}

impl Decoder {
    /// A new decoder.
    pub fn new() -> Decoder {
        Decoder { _dummy: () }
    }

    /// Decompresses `input` into `output` (at least `decompress_len(input)`
    /// bytes); returns the bytes written.
    pub fn decompress(&mut self, input: &[u8], output: &mut [u8]) -> Result<usize> {
        gpu::with_shared(|ctx| {
            let mut n = 0usize;
            let mut e = SnapmiError::default();
            let rc = unsafe {
                gpu::snapmi_raw_decompress(
                    ctx.as_ptr(), input.as_ptr(), input.len(), output.as_mut_ptr(),
                    output.len(), &mut n, &mut e,
                )
            };
            match rc {
                0 => Ok(n),
                k => Err(expect_snap(gpu::to_failure(k, &e, Some(ctx)))),
            }
        })
    }

    /// Decompresses into a fresh vector (reference :105-110).
    pub fn decompress_vec(&mut self, input: &[u8]) -> Result<Vec<u8>> {
        let mut buf = vec![0; decompress_len(input)?];
        let n = self.decompress(input, &mut buf)?;
        buf.truncate(n);
        Ok(buf)
    }
}
