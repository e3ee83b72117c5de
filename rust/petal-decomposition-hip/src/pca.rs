//! `Pca` (`src/pca.rs:41-231` of the reference) and `RandomizedPca` (`src/pca.rs:317-663`) over the C ABI.
use crate::{ffi, view, with_ctx, DecompositionError, HipScalar};
use ndarray::{Array1, Array2, ArrayBase, Data, Ix2};
use rand::Rng;
use rand_distr::StandardNormal;
use rand_pcg::Mcg128Xsl64 as Pcg;
use std::os::raw::c_void;

const N_OVERSAMPLE: usize = 10; // src/pca.rs:679
const N_ITER: i64 = 7; // src/pca.rs:680

/// Exact PCA.  Field names follow the reference so serialized models interchange.
#[cfg_attr(feature = "serde", derive(serde::Serialize, serde::Deserialize))]
#[derive(Debug, Clone)]
pub struct Pca<A: HipScalar> {
    components: Array2<A>,
    n_samples: usize,
    means: Array1<A>,
    total_variance: A,
    singular: Array1<A>,
    centering: bool,
}

impl<A: HipScalar> Pca<A> {
    pub fn new(n_components: usize) -> Self { PcaBuilder::new(n_components).build() }
    pub fn components(&self) -> &Array2<A> { &self.components }
    pub fn mean(&self) -> &Array1<A> { &self.means }
    pub fn n_components(&self) -> usize { self.components.nrows() }
    pub fn singular_values(&self) -> &Array1<A> { &self.singular }
    /// sigma^2 / total_variance (src/pca.rs:101-105).
    pub fn explained_variance_ratio(&self) -> Array1<A> {
        let tv = self.total_variance.to_f64();
        self.singular.mapv(|s| A::from_f64(s.to_f64() * s.to_f64() / tv))
    }

    pub fn fit<S: Data<Elem = A>>(&mut self, input: &ArrayBase<S, Ix2>) -> Result<(), DecompositionError> {
        self.inner_fit(input, None)
    }
    pub fn fit_transform<S: Data<Elem = A>>(&mut self, input: &ArrayBase<S, Ix2>) -> Result<Array2<A>, DecompositionError> {
        let mut y = Array2::<A>::default((input.nrows(), self.n_components()));
        self.inner_fit(input, Some(&mut y))?;
        Ok(y)
    }
    pub fn transform<S: Data<Elem = A>>(&self, input: &ArrayBase<S, Ix2>) -> Result<Array2<A>, DecompositionError> {
        transform(input, &self.components, &self.means, self.centering)
    }
    pub fn inverse_transform<S: Data<Elem = A>>(&self, input: &ArrayBase<S, Ix2>) -> Result<Array2<A>, DecompositionError> {
        inverse_transform(input, &self.components, &self.means, self.centering)
    }

    fn inner_fit<S: Data<Elem = A>>(&mut self, input: &ArrayBase<S, Ix2>, y: Option<&mut Array2<A>>) -> Result<(), DecompositionError> {
        let (k, d) = (self.n_components(), input.ncols());
        let mut comps = Array2::<A>::default((k, d));
        let mut means = Array1::<A>::default(d);
        let mut sing = Array1::<A>::default(k);
        let mut tv = A::default();
        let x = view(input);
        let yv = y.as_ref().map(|y| view(&**y));
        with_ctx(
            |ctx| unsafe {
                ffi::petal_pca_fit(ctx, &x, k as i64, self.centering as i32, comps.as_mut_ptr() as *mut c_void,
                    means.as_mut_ptr() as *mut c_void, sing.as_mut_ptr() as *mut c_void,
                    &mut tv as *mut A as *mut c_void, yv.as_ref().map_or(std::ptr::null(), |v| v as *const _))
            },
            || (),
        )?;
        if input.nrows() == 0 && self.centering {
            return Ok(()); // mean_axis -> None: Ok, model untouched (src/pca.rs:207-211)
        }
        self.components = comps;
        self.means = means;
        self.singular = sing;
        self.total_variance = tv;
        self.n_samples = input.nrows();
        Ok(())
    }
}

pub struct PcaBuilder {
    n_components: usize,
    centering: bool,
}
impl PcaBuilder {
    pub fn new(n_components: usize) -> Self { Self { n_components, centering: true } }
    pub fn centering(mut self, centering: bool) -> Self { self.centering = centering; self }
    pub fn build<A: HipScalar>(self) -> Pca<A> {
        Pca {
            components: Array2::default((self.n_components, 0)),
            n_samples: 0,
            means: Array1::default(0),
            total_variance: A::default(),
            singular: Array1::default(0),
            centering: self.centering,
        }
    }
}

/// Randomized truncated SVD (Halko range finder with power iterations, src/pca.rs:668-718).
#[cfg_attr(feature = "serde", derive(serde::Serialize, serde::Deserialize))]
#[derive(Debug, Clone)]
pub struct RandomizedPca<A: HipScalar, R = Pcg> {
    rng: R,
    components: Array2<A>,
    n_samples: usize,
    means: Array1<A>,
    total_variance: A,
    singular: Array1<A>,
    centering: bool,
}

impl<A: HipScalar> RandomizedPca<A, Pcg> {
    pub fn new(n_components: usize) -> Self { RandomizedPcaBuilder::new(n_components).build() }
    pub fn with_seed(n_components: usize, seed: u128) -> Self { RandomizedPcaBuilder::new(n_components).seed(seed).build() }
}
impl<A: HipScalar, R: Rng> RandomizedPca<A, R> {
    pub fn with_rng(n_components: usize, rng: R) -> Self { RandomizedPcaBuilder::with_rng(rng, n_components).build() }
    pub fn components(&self) -> &Array2<A> { &self.components }
    pub fn mean(&self) -> &Array1<A> { &self.means }
    pub fn n_components(&self) -> usize { self.components.nrows() }
    pub fn singular_values(&self) -> &Array1<A> { &self.singular }
    pub fn explained_variance_ratio(&self) -> Array1<A> {
        let tv = self.total_variance.to_f64();
        self.singular.mapv(|s| A::from_f64(s.to_f64() * s.to_f64() / tv))
    }

    pub fn fit<S: Data<Elem = A>>(&mut self, input: &ArrayBase<S, Ix2>) -> Result<(), DecompositionError> {
        self.inner_fit(input, None)
    }
    pub fn fit_transform<S: Data<Elem = A>>(&mut self, input: &ArrayBase<S, Ix2>) -> Result<Array2<A>, DecompositionError> {
        let mut y = Array2::<A>::default((input.nrows(), self.n_components()));
        self.inner_fit(input, Some(&mut y))?;
        Ok(y)
    }
    pub fn transform<S: Data<Elem = A>>(&self, input: &ArrayBase<S, Ix2>) -> Result<Array2<A>, DecompositionError> {
        transform(input, &self.components, &self.means, self.centering)
    }
    pub fn inverse_transform<S: Data<Elem = A>>(&self, input: &ArrayBase<S, Ix2>) -> Result<Array2<A>, DecompositionError> {
        inverse_transform(input, &self.components, &self.means, self.centering)
    }

    fn inner_fit<S: Data<Elem = A>>(&mut self, input: &ArrayBase<S, Ix2>, y: Option<&mut Array2<A>>) -> Result<(), DecompositionError> {
        let (k, d) = (self.n_components(), input.ncols());
        if input.nrows() < k || d < k {
            return Err(DecompositionError::InvalidInput(format!("every dimension should be at least {k}")));
        }
        if input.nrows() == 0 && self.centering {
            return Ok(());
        }
        // Omega exactly as the reference draws it (src/pca.rs:701-705): d x (k + 10), row-major fill order, one f64
        // StandardNormal draw per entry cast to the element type; the model's RNG advances once per fit.
        let l = k + N_OVERSAMPLE;
        let omega = Array2::<A>::from_shape_fn((d, l), |_| A::from_f64(self.rng.sample::<f64, _>(StandardNormal)));
        let mut comps = Array2::<A>::default((k, d));
        let mut means = Array1::<A>::default(d);
        let mut sing = Array1::<A>::default(k);
        let mut tv = A::default();
        let x = view(input);
        let yv = y.as_ref().map(|y| view(&**y));
        with_ctx(
            |ctx| unsafe {
                ffi::petal_rpca_fit(ctx, &x, k as i64, N_OVERSAMPLE as i64, N_ITER, self.centering as i32,
                    omega.as_ptr() as *const c_void, comps.as_mut_ptr() as *mut c_void, means.as_mut_ptr() as *mut c_void,
                    sing.as_mut_ptr() as *mut c_void, &mut tv as *mut A as *mut c_void,
                    yv.as_ref().map_or(std::ptr::null(), |v| v as *const _))
            },
            || (),
        )?;
        self.components = comps;
        self.means = means;
        self.singular = sing;
        self.total_variance = tv;
        self.n_samples = input.nrows();
        Ok(())
    }
}

pub struct RandomizedPcaBuilder<R> {
    rng: R,
    n_components: usize,
    centering: bool,
}
impl RandomizedPcaBuilder<Pcg> {
    /// Randomly seeded PCG, like the reference (src/pca.rs:578-585).
    pub fn new(n_components: usize) -> Self {
        use rand::SeedableRng;
        let seed: u128 = rand::rng().random();
        Self { rng: Pcg::from_seed(seed.to_be_bytes()), n_components, centering: true }
    }
    /// `Pcg::from_seed(seed.to_be_bytes())` as in src/pca.rs:599-602.
    pub fn seed(mut self, seed: u128) -> Self {
        use rand::SeedableRng;
        self.rng = Pcg::from_seed(seed.to_be_bytes());
        self
    }
}
impl<R: Rng> RandomizedPcaBuilder<R> {
    pub fn with_rng(rng: R, n_components: usize) -> Self { Self { rng, n_components, centering: true } }
    pub fn centering(mut self, centering: bool) -> Self { self.centering = centering; self }
    pub fn build<A: HipScalar>(self) -> RandomizedPca<A, R> {
        RandomizedPca {
            rng: self.rng,
            components: Array2::default((self.n_components, 0)),
            n_samples: 0,
            means: Array1::default(0),
            total_variance: A::default(),
            singular: Array1::default(0),
            centering: self.centering,
        }
    }
}

/// (input - mean) . components^T  (src/pca.rs:726-750)
pub(crate) fn transform<A: HipScalar, S: Data<Elem = A>>(
    input: &ArrayBase<S, Ix2>, components: &Array2<A>, means: &Array1<A>, centering: bool,
) -> Result<Array2<A>, DecompositionError> {
    let (k, d) = components.dim();
    let mut y = Array2::<A>::default((input.nrows(), k));
    let (x, yv) = (view(input), view(&y));
    with_ctx(
        |ctx| unsafe {
            ffi::petal_transform(ctx, &x, components.as_ptr() as *const c_void, means.as_ptr() as *const c_void, k as i64,
                d as i64, centering as i32, &yv)
        },
        || (),
    )?;
    let _ = &mut y; // written through yv
    Ok(y)
}

/// input . components + mean  (src/pca.rs:788-811)
pub(crate) fn inverse_transform<A: HipScalar, S: Data<Elem = A>>(
    input: &ArrayBase<S, Ix2>, components: &Array2<A>, means: &Array1<A>, centering: bool,
) -> Result<Array2<A>, DecompositionError> {
    let (k, d) = components.dim();
    let mut x_out = Array2::<A>::default((input.nrows(), d));
    let (yv, xv) = (view(input), view(&x_out));
    with_ctx(
        |ctx| unsafe {
            ffi::petal_inverse_transform(ctx, &yv, components.as_ptr() as *const c_void, means.as_ptr() as *const c_void,
                k as i64, d as i64, centering as i32, &xv)
        },
        || (),
    )?;
    let _ = &mut x_out;
    Ok(x_out)
}
