//! Raw bindings of include/petal_hip.h (only the entry points the facade uses).
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct PetalCtx {
    _private: [u8; 0],
}

/// `petal_matrix`: a strided view, strides in ELEMENTS (ndarray's convention).
#[repr(C)]
pub struct PetalMatrix {
    pub data: *mut c_void,
    pub rows: i64,
    pub cols: i64,
    pub row_stride: i64,
    pub col_stride: i64,
    pub dtype: c_int, // 0 = f32, 1 = f64
    pub space: c_int, // 0 = host, 1 = device
}

pub const PETAL_OK: c_int = 0;
pub const PETAL_INVALID_INPUT: c_int = 1;
pub const PETAL_LINALG_ERROR: c_int = 2;

extern "C" {
    pub fn petal_ctx_create(device: c_int, stream: *mut c_void, out: *mut *mut PetalCtx) -> c_int;
    pub fn petal_ctx_destroy(ctx: *mut PetalCtx);
    pub fn petal_last_error(ctx: *const PetalCtx) -> *const c_char;
    /// PETAL_OPT_* (include/petal_hip.h): the numerics / kernel-form switches of a ctx; defaults from the environment at creation
    pub fn petal_ctx_set_option(ctx: *mut PetalCtx, option: c_int, value: f64) -> c_int;
    pub fn petal_ctx_get_option(ctx: *const PetalCtx, option: c_int, value: *mut f64) -> c_int;
    pub fn petal_pca_fit(
        ctx: *mut PetalCtx, x: *const PetalMatrix, k: i64, centering: c_int, components: *mut c_void,
        means: *mut c_void, singular: *mut c_void, total_variance: *mut c_void, y_out: *const PetalMatrix,
    ) -> c_int;
    pub fn petal_rpca_fit(
        ctx: *mut PetalCtx, x: *const PetalMatrix, k: i64, n_oversample: i64, n_iter: i64, centering: c_int,
        omega: *const c_void, components: *mut c_void, means: *mut c_void, singular: *mut c_void,
        total_variance: *mut c_void, y_out: *const PetalMatrix,
    ) -> c_int;
    pub fn petal_transform(
        ctx: *mut PetalCtx, x: *const PetalMatrix, components: *const c_void, means: *const c_void, k: i64,
        d: i64, centering: c_int, y_out: *const PetalMatrix,
    ) -> c_int;
    pub fn petal_inverse_transform(
        ctx: *mut PetalCtx, y: *const PetalMatrix, components: *const c_void, means: *const c_void, k: i64,
        d: i64, centering: c_int, x_out: *const PetalMatrix,
    ) -> c_int;
    pub fn petal_fastica_fit(
        ctx: *mut PetalCtx, x: *const PetalMatrix, n_components: i64, tol: f64, max_iter: i64, mode: c_int,
        w_init: *const c_void, components: *mut c_void, means: *mut c_void, n_iter: *mut i64,
        y_out: *const PetalMatrix,
    ) -> c_int;
}
