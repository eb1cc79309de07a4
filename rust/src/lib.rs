//! `Layer<Index, ID>` backed by the MI355X kernels behind `include/bp_layer.h`.
//!
//! Same public surface as the reference crate's hot path (`src/lib.rs:80-82`):
//! `LayerBuilder::new().with_min_depth(..).build::<Index64_3D, u32>()`, then
//! `clear / extend / merge / sort / par_sort / scan / scan_filtered / par_scan / par_scan_filtered`.
//! NOT compiled in this pipeline (no Rust toolchain); kept in sync with the tested C++ facade
//! (`include/broadphase/layer.hpp`).
use std::marker::PhantomData;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct BpLayer { _private: [u8; 0] }
pub type BpFilterFn = extern "C" fn(id: u32, id_: u32, user: *mut c_void) -> c_int;

extern "C" {
    fn bp_status_string(status: c_int) -> *const c_char;
    fn bp_layer_last_error(layer: *const BpLayer) -> *const c_char;
    fn bp_layer_create(out: *mut *mut BpLayer, index_kind: c_int, min_depth: u32, index_capacity: usize,
                       collision_capacity: usize, device: c_int) -> c_int;
    fn bp_layer_destroy(layer: *mut BpLayer) -> c_int;
    fn bp_layer_clone(layer: *const BpLayer, out: *mut *mut BpLayer) -> c_int;
    fn bp_layer_equals(a: *const BpLayer, b: *const BpLayer, equal: *mut c_int) -> c_int;
    fn bp_layer_clear(layer: *mut BpLayer) -> c_int;
    fn bp_layer_extend_host(layer: *mut BpLayer, sys_min: *const f32, sys_max: *const f32, bounds: *const f32,
                            ids: *const u32, n: usize) -> c_int;
    fn bp_layer_merge(layer: *mut BpLayer, other: *const BpLayer) -> c_int;
    fn bp_layer_sort(layer: *mut BpLayer) -> c_int;
    fn bp_layer_par_sort(layer: *mut BpLayer) -> c_int;
    fn bp_layer_scan(layer: *mut BpLayer, pairs: *mut *const u32, n: *mut usize) -> c_int;
    fn bp_layer_par_scan(layer: *mut BpLayer, pairs: *mut *const u32, n: *mut usize) -> c_int;
    fn bp_layer_scan_filtered(layer: *mut BpLayer, f: BpFilterFn, user: *mut c_void, pairs: *mut *const u32,
                              n: *mut usize) -> c_int;
    fn bp_layer_par_scan_filtered(layer: *mut BpLayer, f: BpFilterFn, user: *mut c_void, pairs: *mut *const u32,
                                  n: *mut usize) -> c_int;
    fn bp_layer_tree(layer: *mut BpLayer, keys: *mut *const u64, ids: *mut *const u32, m: *mut usize,
                     sorted: *mut c_int) -> c_int;
    fn bp_layer_test_box_batch(layer: *mut BpLayer, sys_min: *const f32, sys_max: *const f32, boxes: *const f32,
                               n_queries: usize, has_max_depth: c_int, max_depth: u32, rows: *mut *const u32,
                               n_rows: *mut usize) -> c_int;
}

/// Index kinds of the reference (`src/index.rs:293-295`); DIM f32 per corner on the wire.
pub trait SpatialIndexKind { const KIND: c_int; const DIM: usize; }
pub struct Index32_2D; pub struct Index64_2D; pub struct Index64_3D;
impl SpatialIndexKind for Index32_2D { const KIND: c_int = 0; const DIM: usize = 2; }
impl SpatialIndexKind for Index64_2D { const KIND: c_int = 1; const DIM: usize = 2; }
impl SpatialIndexKind for Index64_3D { const KIND: c_int = 2; const DIM: usize = 3; }

/// `Bounds<Point>` (`src/geom.rs:84-87`) flattened: min[DIM] then max[DIM].
#[derive(Copy, Clone, Debug, PartialEq)]
pub struct Bounds<const D: usize> { pub min: [f32; D], pub max: [f32; D] }

fn check(rc: c_int, layer: *const BpLayer) {
    if rc != 0 {
        // the reference has no Result on this path; a device failure is a panic
        let (s, m) = unsafe {
            (std::ffi::CStr::from_ptr(bp_status_string(rc)).to_string_lossy().into_owned(),
             std::ffi::CStr::from_ptr(bp_layer_last_error(layer)).to_string_lossy().into_owned())
        };
        panic!("broadphase_amd: {}: {}", s, m);
    }
}

pub struct Layer<Index: SpatialIndexKind, ID = u32> {
    handle: *mut BpLayer,
    collisions: Vec<(u32, u32)>,
    _marker: PhantomData<(Index, ID)>,
}

unsafe impl<Index: SpatialIndexKind, ID> Send for Layer<Index, ID> {}

impl<Index: SpatialIndexKind> Layer<Index, u32> {
    /// `Layer::clear` (`src/layer.rs:84-88`)
    pub fn clear(&mut self) { check(unsafe { bp_layer_clear(self.handle) }, self.handle) }

    /// `Layer::extend` (`src/layer.rs:94-121`)
    pub fn extend<const D: usize, Iter>(&mut self, system_bounds: Bounds<D>, objects: Iter)
    where Iter: Iterator<Item = (Bounds<D>, u32)> {
        assert_eq!(D, Index::DIM);
        let (lo, _) = objects.size_hint();
        let mut flat: Vec<f32> = Vec::with_capacity(lo * 2 * D);
        let mut ids: Vec<u32> = Vec::with_capacity(lo);
        for (b, id) in objects { flat.extend_from_slice(&b.min); flat.extend_from_slice(&b.max); ids.push(id); }
        check(unsafe { bp_layer_extend_host(self.handle, system_bounds.min.as_ptr(), system_bounds.max.as_ptr(),
                                            flat.as_ptr(), ids.as_ptr(), ids.len()) }, self.handle)
    }

    /// `Layer::merge` (`src/layer.rs:127-138`)
    pub fn merge(&mut self, other: &Layer<Index, u32>) { check(unsafe { bp_layer_merge(self.handle, other.handle) }, self.handle) }
    /// `Layer::sort` / `par_sort` (`src/layer.rs:146-165`)
    pub fn sort(&mut self) { check(unsafe { bp_layer_sort(self.handle) }, self.handle) }
    pub fn par_sort(&mut self) { check(unsafe { bp_layer_par_sort(self.handle) }, self.handle) }

    fn collect(&mut self, p: *const u32, n: usize) -> &Vec<(u32, u32)> {
        // the ABI hands back interleaved u32 (Rust tuple layout is unspecified): rebuild the Vec
        self.collisions.clear();
        let s = unsafe { std::slice::from_raw_parts(p, 2 * n) };
        self.collisions.extend(s.chunks_exact(2).map(|c| (c[0], c[1])));
        &self.collisions
    }
    /// `Layer::test_box` (`src/layer.rs:294-313`): ids whose cells overlap `test_bounds`, sorted, duplicate-free
    pub fn test_box<const D: usize>(&mut self, system_bounds: Bounds<D>, test_bounds: Bounds<D>, max_depth: Option<u32>) -> Vec<u32> {
        self.test_box_batch(system_bounds, &[test_bounds], max_depth).into_iter().map(|(_, id)| id).collect()
    }
    /// many boxes in one device pass: rows `(query index, id)`, sorted by query then id
    pub fn test_box_batch<const D: usize>(&mut self, system_bounds: Bounds<D>, boxes: &[Bounds<D>], max_depth: Option<u32>) -> Vec<(u32, u32)> {
        assert_eq!(D, Index::DIM);
        let mut flat: Vec<f32> = Vec::with_capacity(boxes.len() * 2 * D);
        for b in boxes { flat.extend_from_slice(&b.min); flat.extend_from_slice(&b.max); }
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { bp_layer_test_box_batch(self.handle, system_bounds.min.as_ptr(), system_bounds.max.as_ptr(), flat.as_ptr(),
                                               boxes.len(), max_depth.is_some() as c_int, max_depth.unwrap_or(0), &mut p, &mut n) },
              self.handle);
        let s = unsafe { std::slice::from_raw_parts(p, 2 * n) };
        s.chunks_exact(2).map(|c| (c[0], c[1])).collect()
    }
    /// `Layer::scan` (`src/layer.rs:449-453`)
    pub fn scan<'a>(&'a mut self) -> &'a Vec<(u32, u32)> {
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { bp_layer_scan(self.handle, &mut p, &mut n) }, self.handle);
        self.collect(p, n)
    }
    /// `Layer::par_scan` (`src/layer.rs:482-488`)
    pub fn par_scan<'a>(&'a mut self) -> &'a Vec<(u32, u32)> {
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { bp_layer_par_scan(self.handle, &mut p, &mut n) }, self.handle);
        self.collect(p, n)
    }
    /// `Layer::scan_filtered` (`src/layer.rs:460-477`): the filter sees raw pairs before dedup
    pub fn scan_filtered<'a, F: FnMut(u32, u32) -> bool>(&'a mut self, mut filter: F) -> &'a Vec<(u32, u32)> {
        extern "C" fn thunk<F: FnMut(u32, u32) -> bool>(a: u32, b: u32, u: *mut c_void) -> c_int {
            let f = unsafe { &mut *(u as *mut F) };
            if f(a, b) { 1 } else { 0 }
        }
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { bp_layer_scan_filtered(self.handle, thunk::<F>, &mut filter as *mut F as *mut c_void, &mut p, &mut n) }, self.handle);
        self.collect(p, n)
    }
    /// `Layer::par_scan_filtered` (`src/layer.rs:493-520`)
    pub fn par_scan_filtered<'a, F: Copy + Send + Sync + FnMut(u32, u32) -> bool>(&'a mut self, mut filter: F) -> &'a Vec<(u32, u32)> {
        extern "C" fn thunk<F: FnMut(u32, u32) -> bool>(a: u32, b: u32, u: *mut c_void) -> c_int {
            let f = unsafe { &mut *(u as *mut F) };
            if f(a, b) { 1 } else { 0 }
        }
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { bp_layer_par_scan_filtered(self.handle, thunk::<F>, &mut filter as *mut F as *mut c_void, &mut p, &mut n) }, self.handle);
        self.collect(p, n)
    }
    /// `Layer::iter` (`src/layer.rs:79-81`): (index value, id) in current order
    pub fn iter(&mut self) -> Vec<(u64, u32)> {
        let (mut k, mut i, mut m, mut s) = (std::ptr::null(), std::ptr::null(), 0usize, 0 as c_int);
        check(unsafe { bp_layer_tree(self.handle, &mut k, &mut i, &mut m, &mut s) }, self.handle);
        let (ks, is) = unsafe { (std::slice::from_raw_parts(k, m), std::slice::from_raw_parts(i, m)) };
        ks.iter().cloned().zip(is.iter().cloned()).collect()
    }
}

impl<Index: SpatialIndexKind, ID> Drop for Layer<Index, ID> {
    fn drop(&mut self) { unsafe { bp_layer_destroy(self.handle); } }
}
impl<Index: SpatialIndexKind> Clone for Layer<Index, u32> {
    fn clone(&self) -> Self {
        let mut h = std::ptr::null_mut();
        check(unsafe { bp_layer_clone(self.handle, &mut h) }, self.handle);
        Layer { handle: h, collisions: Vec::new(), _marker: PhantomData }
    }
}
impl<Index: SpatialIndexKind> PartialEq for Layer<Index, u32> {
    fn eq(&self, other: &Self) -> bool {
        let mut eq = 0;
        check(unsafe { bp_layer_equals(self.handle, other.handle, &mut eq) }, self.handle);
        eq != 0
    }
}

/// `LayerBuilder` (`src/layer.rs:619-696`)
#[derive(Default)]
pub struct LayerBuilder { min_depth: u32, index_capacity: Option<usize>, collision_capacity: Option<usize>, test_capacity: Option<usize> }
impl LayerBuilder {
    pub fn new() -> Self { Self::default() }
    pub fn with_min_depth(&mut self, depth: u32) -> &mut Self { self.min_depth = depth; self }
    pub fn with_index_capacity(&mut self, c: usize) -> &mut Self { self.index_capacity = Some(c); self }
    pub fn with_collision_capacity(&mut self, c: usize) -> &mut Self { self.collision_capacity = Some(c); self }
    pub fn with_test_capacity(&mut self, c: usize) -> &mut Self { self.test_capacity = Some(c); self }
    pub fn build<Index: SpatialIndexKind, ID>(&self) -> Layer<Index, ID> {
        let mut h = std::ptr::null_mut();
        check(unsafe { bp_layer_create(&mut h, Index::KIND, self.min_depth, self.index_capacity.unwrap_or(0),
                                       self.collision_capacity.unwrap_or(0), -1) }, std::ptr::null());
        Layer { handle: h, collisions: Vec::new(), _marker: PhantomData }
    }
}
