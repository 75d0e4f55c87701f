//! FFI to `libsnapmi.so` (include/snapmi.h) and the context wrapper.
//!
//! One `Context` = one HIP device + stream + device scratch.  The reference's
//! `Encoder` owns its hash tables behind `&mut self` (src/compress.rs:67-70);
//! here it owns a `Context` the same way.  `Decoder` is stateless in the
//! reference; it uses a thread-local context.
use std::cell::RefCell;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};
use std::ptr;

use crate::error::Error;

/// `snapmi_error`: variant + fields of `snap::Error`.
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct SnapmiError {
    pub kind: i32,
    pub reserved: u32,
    pub a: u64,
    pub b: u64,
    pub c: u64,
}

#[repr(C)]
pub struct SnapmiCtx {
    _private: [u8; 0],
}

pub const SNAPMI_FRAME_NO_IDENT: u32 = 1;
pub const SNAPMI_FRAME_CONTINUATION: u32 = 1;
pub const SNAPMI_FRAME_FINAL: u32 = 2;
pub const SNAPMI_E_UNEXPECTED_EOF: i32 = 64;
pub const SNAPMI_E_DEVICE: i32 = 100;
/// `snapmi_ctx_prepare`: place the tables at the far end of the device's
/// memory (holds ALL free device memory for a moment: include/snapmi.h)
pub const SNAPMI_PREPARE_TOP_OF_MEMORY: u32 = 1;

extern "C" {
    pub fn snapmi_ctx_create(device: c_int, hip_stream: *mut c_void, out: *mut *mut SnapmiCtx) -> c_int;
    pub fn snapmi_ctx_destroy(ctx: *mut SnapmiCtx);
    pub fn snapmi_last_error(ctx: *const SnapmiCtx) -> *const c_char;
    pub fn snapmi_max_compress_len(input_len: usize) -> usize;
    pub fn snapmi_decompress_len(
        input: *const u8, input_len: usize, result: *mut usize, err: *mut SnapmiError,
    ) -> c_int;
    pub fn snapmi_raw_compress(
        ctx: *mut SnapmiCtx, input: *const u8, input_len: usize, output: *mut u8,
        output_cap: usize, written: *mut usize, err: *mut SnapmiError,
    ) -> c_int;
    pub fn snapmi_raw_decompress(
        ctx: *mut SnapmiCtx, input: *const u8, input_len: usize, output: *mut u8,
        output_cap: usize, written: *mut usize, err: *mut SnapmiError,
    ) -> c_int;
    pub fn snapmi_ctx_prepare(ctx: *mut SnapmiCtx, blocks: u64, flags: u32) -> c_int;
    pub fn snapmi_ctx_get_info(ctx: *mut SnapmiCtx, name: *const c_char, value: *mut i64) -> c_int;
    pub fn snapmi_host_alloc(bytes: usize) -> *mut c_void;
    pub fn snapmi_host_free(p: *mut c_void);
    pub fn snapmi_frame_scan_host(
        h_in: *const c_void, in_len: u64, flags: u32, stale10: *mut u8, h_offsets: *mut u64,
        cap: u64, n_chunks: *mut u64, consumed: *mut u64,
    ) -> c_int;
    pub fn snapmi_frame_encode_bound(total_bytes: usize, n_chunks: usize) -> usize;
    pub fn snapmi_frame_encode_host(
        ctx: *mut SnapmiCtx, h_in: *const u8, h_chunk_lens: *const u32, n: usize, flags: u32,
        h_out: *mut u8, out_cap: usize, written: *mut usize,
    ) -> c_int;
    pub fn snapmi_frame_decode_host(
        ctx: *mut SnapmiCtx, h_in: *const u8, in_len: usize, flags: u32, stale10: *mut u8,
        h_out: *mut u8, out_cap: usize, written: *mut usize, consumed: *mut usize,
        err: *mut SnapmiError,
    ) -> c_int;
}

/// Page-locked, device-mapped host memory (`snapmi_host_alloc`), used like a
/// `Vec<u8>`.  The host-buffer calls (`snapmi_frame_encode_host`,
/// `snapmi_frame_decode_host`) copy asynchronously - slice i+1 on its way to
/// the device, the kernels of slice i, the result of slice i-1 on its way
/// home - only from and to memory like this; from a pageable `Vec` the three
/// legs run one after the other.  The adapters
/// stage their batches here, as rust-snappy_amd/frame.py does (`HostBuffer`).
pub struct PinnedBuf {
    ptr: *mut u8,
    cap: usize,
    len: usize,
}

unsafe impl Send for PinnedBuf {}

impl PinnedBuf {
    pub fn new() -> PinnedBuf {
        PinnedBuf { ptr: ptr::null_mut(), cap: 0, len: 0 }
    }

    pub fn with_len(len: usize) -> PinnedBuf {
        let mut b = PinnedBuf::new();
        b.resize(len);
        b
    }

    /// Room for at least `cap` bytes; what is in `[..len]` is kept.  Grows by
    /// an eighth beyond the request, like the library's own scratch.
    pub fn reserve(&mut self, cap: usize) {
        if cap <= self.cap {
            return;
        }
        let want = cap + cap / 8;
        let p = unsafe { snapmi_host_alloc(want) } as *mut u8;
        if p.is_null() {
            panic!("snap (MI355X): snapmi_host_alloc({}) failed", want);
        }
        if self.len > 0 {
            unsafe { ptr::copy_nonoverlapping(self.ptr, p, self.len) };
        }
        if !self.ptr.is_null() {
            unsafe { snapmi_host_free(self.ptr as *mut c_void) };
        }
        self.ptr = p;
        self.cap = want;
    }

    /// `Vec::resize(len, 0)`.
    pub fn resize(&mut self, len: usize) {
        self.reserve(len);
        if len > self.len {
            unsafe { ptr::write_bytes(self.ptr.add(self.len), 0, len - self.len) };
        }
        self.len = len;
    }

    pub fn clear(&mut self) {
        self.len = 0;
    }

    pub fn extend_from_slice(&mut self, s: &[u8]) {
        self.reserve(self.len + s.len());
        unsafe { ptr::copy_nonoverlapping(s.as_ptr(), self.ptr.add(self.len), s.len()) };
        self.len += s.len();
    }
}

impl std::ops::Deref for PinnedBuf {
    type Target = [u8];
    fn deref(&self) -> &[u8] {
        if self.ptr.is_null() {
            &[]
        } else {
            unsafe { std::slice::from_raw_parts(self.ptr, self.len) }
        }
    }
}

impl std::ops::DerefMut for PinnedBuf {
    fn deref_mut(&mut self) -> &mut [u8] {
        if self.ptr.is_null() {
            &mut []
        } else {
            unsafe { std::slice::from_raw_parts_mut(self.ptr, self.len) }
        }
    }
}

impl Drop for PinnedBuf {
    fn drop(&mut self) {
        if !self.ptr.is_null() {
            unsafe { snapmi_host_free(self.ptr as *mut c_void) };
        }
    }
}

/// Owner of a `snapmi_ctx`.
pub struct Context {
    raw: *mut SnapmiCtx,
}

// A context is used by one thread at a time (`&mut` everywhere below).
unsafe impl Send for Context {}

impl Context {
    /// A context on device `SNAPMI_DEVICE` (default 0).  Panics without a
    /// usable GPU: the codec has no CPU path to fall back to.
    pub fn new() -> Context {
        let dev = std::env::var("SNAPMI_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut raw = ptr::null_mut();
        let rc = unsafe { snapmi_ctx_create(dev, ptr::null_mut(), &mut raw) };
        if rc != 0 || raw.is_null() {
            panic!("snap (MI355X): no usable HIP device {} (snapmi_ctx_create = {})", dev, rc);
        }
        Context { raw }
    }

    pub fn as_ptr(&mut self) -> *mut SnapmiCtx {
        self.raw
    }

    /// Text of the last device / argument failure on this context.
    pub fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(snapmi_last_error(self.raw)).to_string_lossy().into_owned() }
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { snapmi_ctx_destroy(self.raw) }
    }
}

thread_local! {
    static SHARED: RefCell<Option<Context>> = RefCell::new(None);
}

/// Runs `f` with this thread's shared context (the stateless `Decoder`).
pub fn with_shared<T>(f: impl FnOnce(&mut Context) -> T) -> T {
    SHARED.with(|c| {
        let mut slot = c.borrow_mut();
        f(slot.get_or_insert_with(Context::new))
    })
}

/// What a non-zero return of the C ABI means on the Rust side.
pub enum Failure {
    /// A `snap::Error` (reference `src/error.rs:72-180`).
    Snap(Error),
    /// `io::ErrorKind::UnexpectedEof` of the frame reader (src/read.rs:151...).
    UnexpectedEof,
    /// HIP / device / argument failure: not a property of the data.
    Device(String),
}

/// Rebuilds the reference's error from `(kind, a, b, c)`: every variant of
/// `src/error.rs:72-180`, the reader's EOF, and the device class.
pub fn to_failure(kind: i32, e: &SnapmiError, ctx: Option<&Context>) -> Failure {
    let (a, b, c) = (e.a, e.b, e.c);
    Failure::Snap(match kind {
        1 => Error::TooBig { given: a, max: b },
        2 => Error::BufferTooSmall { given: a, min: b },
        3 => Error::Empty,
        4 => Error::Header,
        5 => Error::HeaderMismatch { expected_len: a, got_len: b },
        6 => Error::Literal { len: a, src_len: b, dst_len: c },
        7 => Error::CopyRead { len: a, src_len: b },
        8 => Error::CopyWrite { len: a, dst_len: b },
        9 => Error::Offset { offset: a, dst_pos: b },
        10 => Error::StreamHeader { byte: a as u8 },
        // the ABI packs the six body bytes little-endian into `a`
        11 => Error::StreamHeaderMismatch { bytes: a.to_le_bytes()[..6].to_vec() },
        12 => Error::UnsupportedChunkType { byte: a as u8 },
        13 => Error::UnsupportedChunkLength { len: a, header: b != 0 },
        14 => Error::Checksum { expected: a as u32, got: b as u32 },
        SNAPMI_E_UNEXPECTED_EOF => return Failure::UnexpectedEof,
        _ => {
            let text = ctx.map(|c| c.last_error()).unwrap_or_default();
            return Failure::Device(format!("snapmi error {}: {}", kind, text));
        }
    })
}
