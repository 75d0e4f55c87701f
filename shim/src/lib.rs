//! `snap` on an MI355X: the public surface of BurntSushi/rust-snappy's hot
//! path (reference `src/lib.rs:56-109`) over `libsnapmi.so`.
//!
//! ```text
//! snap::raw::{Encoder, Decoder, max_compress_len, decompress_len}
//! snap::read::{FrameDecoder, FrameEncoder}
//! snap::write::FrameEncoder
//! snap::{Error, Result}, snap::write::IntoInnerError
//! ```
//!
//! Every compress / decompress call runs HIP kernels; there is no CPU codec
//! in this crate.  A call that cannot reach the GPU fails with an
//! `io::Error` of kind `Other` (stream types) or panics in the infallible
//! constructors, never with a silently computed result.
#![deny(missing_docs)]

mod error;
mod gpu;

/// Raw (unframed) Snappy, reference `src/raw.rs`.
pub mod raw;
/// Streaming decompression / compression on read, reference `src/read.rs`.
pub mod read;
/// Streaming compression on write, reference `src/write.rs`.
pub mod write;

pub use crate::error::{Error, Result};

/// Largest input a raw stream can describe, reference `src/lib.rs:93`.
pub(crate) const MAX_INPUT_SIZE: u64 = std::u32::MAX as u64;
/// Block / chunk size of the format, reference `src/lib.rs:97`.
pub(crate) const MAX_BLOCK_SIZE: usize = 1 << 16;
