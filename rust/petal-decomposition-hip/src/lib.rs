//! petal-decomposition's public surface (`Pca`, `RandomizedPca`, `FastIca` and their builders, `DecompositionError`)
//! computed on an MI355X through `libpetal_hip.so`.  UNVERIFIED SOURCE (never compiled: no Rust toolchain in the build
//! image) -- see Cargo.toml.  Model state lives in these structs exactly as in the reference
//! (`src/pca.rs:41-51, 317-329`, `src/ica.rs:41-50`), so serde persistence keeps its field names.
mod ffi;
mod ica;
mod pca;

pub use ica::{FastIca, FastIcaBuilder};
pub use pca::{Pca, PcaBuilder, RandomizedPca, RandomizedPcaBuilder};

use ndarray::{ArrayBase, Data, Ix2};
use std::ffi::CStr;
use std::os::raw::{c_int, c_void};
use std::sync::{Mutex, OnceLock};

/// Same variants and messages as the reference (`src/lib.rs:22-28`).
#[derive(Debug, thiserror::Error)]
pub enum DecompositionError {
    #[error("invalid matrix: {0}")]
    InvalidInput(String),
    #[error("linear algebra operation failed: {0}")]
    LinalgError(String),
}

/// Element types the device path computes in (the reference is generic over `lair::Scalar`; complex scalars are out of
/// scope here).
pub trait HipScalar: Copy + Default + 'static + sealed::Sealed {
    const DTYPE: c_int;
    fn from_f64(v: f64) -> Self;
    fn to_f64(self) -> f64;
}
mod sealed {
    pub trait Sealed {}
    impl Sealed for f32 {}
    impl Sealed for f64 {}
}
impl HipScalar for f32 {
    const DTYPE: c_int = 0;
    fn from_f64(v: f64) -> Self { v as f32 }
    fn to_f64(self) -> f64 { f64::from(self) }
}
impl HipScalar for f64 {
    const DTYPE: c_int = 1;
    fn from_f64(v: f64) -> Self { v }
    fn to_f64(self) -> f64 { self }
}

/// One context per process (GPU 0 unless `PETAL_HIP_DEVICE` says otherwise); the C ABI wants one caller at a time.
struct Ctx(*mut ffi::PetalCtx);
unsafe impl Send for Ctx {}
impl Drop for Ctx {
    fn drop(&mut self) { unsafe { ffi::petal_ctx_destroy(self.0) } }
}
static CTX: OnceLock<Result<Mutex<Ctx>, String>> = OnceLock::new();

pub(crate) fn with_ctx<T>(f: impl FnOnce(*mut ffi::PetalCtx) -> c_int, ok: impl FnOnce() -> T) -> Result<T, DecompositionError> {
    let slot = CTX.get_or_init(|| {
        let device = std::env::var("PETAL_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { ffi::petal_ctx_create(device, std::ptr::null_mut(), &mut raw) };
        if rc == ffi::PETAL_OK && !raw.is_null() { Ok(Mutex::new(Ctx(raw))) } else { Err(format!("petal_ctx_create failed ({rc}): no usable MI355X")) }
    });
    let guard = match slot {
        Ok(m) => m.lock().unwrap_or_else(|p| p.into_inner()),
        Err(msg) => return Err(DecompositionError::LinalgError(msg.clone())),
    };
    let rc = f(guard.0);
    if rc == ffi::PETAL_OK {
        return Ok(ok());
    }
    let msg = unsafe { CStr::from_ptr(ffi::petal_last_error(guard.0)) }.to_string_lossy().into_owned();
    Err(if rc == ffi::PETAL_INVALID_INPUT { DecompositionError::InvalidInput(msg) } else { DecompositionError::LinalgError(msg) })
}

/// A borrowed ndarray view as the ABI's strided matrix (any layout ndarray accepts: strides are already in elements).
pub(crate) fn view<A: HipScalar, S: Data<Elem = A>>(a: &ArrayBase<S, Ix2>) -> ffi::PetalMatrix {
    let (rows, cols) = a.dim();
    let st = a.strides();
    ffi::PetalMatrix {
        data: a.as_ptr() as *mut c_void,
        rows: rows as i64,
        cols: cols as i64,
        row_stride: st[0] as i64,
        col_stride: st[1] as i64,
        dtype: A::DTYPE,
        space: 0,
    }
}
