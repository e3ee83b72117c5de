//! `FastIca` (`src/ica.rs:41-308` of the reference) over the C ABI.
use crate::{ffi, pca, view, with_ctx, DecompositionError, HipScalar};
use ndarray::{Array1, Array2, ArrayBase, Data, Ix2};
use rand::Rng;
use rand_distr::StandardNormal;
use rand_pcg::Mcg128Xsl64 as Pcg;
use std::os::raw::c_void;

const TOL: f64 = 1e-4; // src/ica.rs:217
const MAX_ITER: i64 = 200; // src/ica.rs:216
/// 0 = textbook (W W^T)^(-1/2) W, 1 = the reference's literal arithmetic.
const MODE: i32 = 0;

#[cfg_attr(feature = "serde", derive(serde::Serialize, serde::Deserialize))]
#[derive(Debug, Clone)]
pub struct FastIca<A: HipScalar, R = Pcg> {
    rng: R,
    components: Array2<A>,
    means: Array1<A>,
    n_iter: usize,
}

impl<A: HipScalar> FastIca<A, Pcg> {
    pub fn new() -> Self { FastIcaBuilder::new().build() }
    pub fn with_seed(seed: u128) -> Self { FastIcaBuilder::new().seed(seed).build() }
}
impl<A: HipScalar> Default for FastIca<A, Pcg> {
    fn default() -> Self { Self::new() }
}

impl<A: HipScalar, R: Rng> FastIca<A, R> {
    pub fn with_rng(rng: R) -> Self { FastIcaBuilder::with_rng(rng).build() }

    pub fn fit<S: Data<Elem = A>>(&mut self, input: &ArrayBase<S, Ix2>) -> Result<(), DecompositionError> {
        self.inner_fit(input, None)
    }
    pub fn fit_transform<S: Data<Elem = A>>(&mut self, input: &ArrayBase<S, Ix2>) -> Result<Array2<A>, DecompositionError> {
        let nc = input.nrows().min(input.ncols());
        let mut y = Array2::<A>::default((input.nrows(), nc));
        self.inner_fit(input, Some(&mut y))?;
        Ok(y)
    }
    /// (input - mean) . components^T  (src/ica.rs:120-131)
    pub fn transform<S: Data<Elem = A>>(&self, input: &ArrayBase<S, Ix2>) -> Result<Array2<A>, DecompositionError> {
        if input.ncols() != self.means.len() {
            return Err(DecompositionError::InvalidInput("too many columns".to_string()));
        }
        pca::transform(input, &self.components, &self.means, true)
    }

    fn inner_fit<S: Data<Elem = A>>(&mut self, input: &ArrayBase<S, Ix2>, y: Option<&mut Array2<A>>) -> Result<(), DecompositionError> {
        let (n, d) = input.dim();
        let nc = n.min(d); // src/ica.rs:173
        if n == 0 {
            return Ok(()); // src/ica.rs:174-176
        }
        // w_init exactly as the reference draws it (src/ica.rs:210-214): nc x nc, row-major, f64 StandardNormal -> A
        let w_init = Array2::<A>::from_shape_fn((nc, nc), |_| A::from_f64(self.rng.sample::<f64, _>(StandardNormal)));
        let mut comps = Array2::<A>::default((nc, d));
        let mut means = Array1::<A>::default(d);
        let mut n_iter: i64 = 0;
        let x = view(input);
        let yv = y.as_ref().map(|y| view(&**y));
        with_ctx(
            |ctx| unsafe {
                ffi::petal_fastica_fit(ctx, &x, 0, TOL, MAX_ITER, MODE, w_init.as_ptr() as *const c_void,
                    comps.as_mut_ptr() as *mut c_void, means.as_mut_ptr() as *mut c_void, &mut n_iter,
                    yv.as_ref().map_or(std::ptr::null(), |v| v as *const _))
            },
            || (),
        )?;
        self.components = comps;
        self.means = means;
        self.n_iter = n_iter as usize;
        Ok(())
    }
}

pub struct FastIcaBuilder<R> {
    rng: R,
}
impl FastIcaBuilder<Pcg> {
    /// Randomly seeded PCG, like the reference (src/ica.rs:255-260).
    pub fn new() -> Self {
        use rand::SeedableRng;
        let seed: u128 = rand::rng().random();
        Self { rng: Pcg::from_seed(seed.to_be_bytes()) }
    }
    pub fn seed(mut self, seed: u128) -> Self {
        use rand::SeedableRng;
        self.rng = Pcg::from_seed(seed.to_be_bytes());
        self
    }
}
impl Default for FastIcaBuilder<Pcg> {
    fn default() -> Self { Self::new() }
}
impl<R: Rng> FastIcaBuilder<R> {
    pub fn with_rng(rng: R) -> Self { Self { rng } }
    pub fn build<A: HipScalar>(self) -> FastIca<A, R> {
        FastIca { rng: self.rng, components: Array2::default((0, 0)), means: Array1::default(0), n_iter: 0 }
    }
}
