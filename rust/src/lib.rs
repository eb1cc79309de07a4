//! `Layer<Index, ID>` backed by the MI355X kernels behind `include/bp_layer.h` (+ `include/bp_sharded.h`).
//!
//! Type-compatible with the hot-path surface the reference crate re-exports (`src/lib.rs:80-82`):
//!   * `Index32_2D`, `Index64_2D`, `Index64_3D` are value-carrying newtypes implementing `SpatialIndex`
//!     (`src/index.rs:38-63, 293-295`) -- `depth`, `origin`, `set_depth`, `set_origin`, `overlaps`,
//!     `same_cell_at_depth`, `subdivide`, `Ord` on the raw value;
//!   * `Bounds<Point>` is generic over cgmath points (`src/geom.rs:84-133`);
//!   * `Layer::extend<Iter, Point_>(Bounds<Point_>, Iter)` takes cgmath `Bounds<Point2<f32>>` /
//!     `Bounds<Point3<f32>>` exactly like `src/layer.rs:94-101`, `iter()` yields `&(Index, ID)`
//!     (`src/layer.rs:79-81`), `scan` / `par_scan` / `scan_filtered` / `par_scan_filtered` return
//!     `&Vec<(ID, ID)>` (`src/layer.rs:449-520`), `LayerBuilder` has the reference's knobs (`:619-696`).
//! So `examples/main.rs:364-374, 410-459` (`clear -> extend -> par_sort -> par_scan` over
//! `Bounds<Point2<f32>>`, `Index32_2D`, `with_min_depth(4)`) compiles against this crate by changing the `use`.
//! With the `serde` feature `Layer`, the index types and `Bounds` serialize exactly like the reference's derives
//! (`src/layer.rs:41-68`, `src/index.rs:68`, `src/geom.rs:83`), so `data/src/lib.rs:28-49` and
//! `tests/test_layer.rs:19-124` keep compiling by changing the `use`.
//! `ID` is `u32`.  Zero-copy access to the result rows is offered next to the reference's
//! signature: `par_scan_rows() -> &[Pair]` (`#[repr(C)]`, the ABI's own layout).
//!
//! NOT compiled in this pipeline (the image has no Rust toolchain).  The C++ facade
//! (`include/broadphase/layer.hpp`) has the same shape and IS built and run on the GPU:
//! `tests/cpp/test_facade.cpp` ends with the example's frame loop written against it call for call.
extern crate cgmath;
#[cfg(feature = "serde")]
#[macro_use]
extern crate serde;

use cgmath::{Array, EuclideanSpace, Point2, Point3, Vector2, Vector3};
use std::cell::{Cell, UnsafeCell};
use std::fmt::Debug;
use std::hash::Hash;
use std::os::raw::{c_char, c_int, c_void};

// ---------------------------------------------------------------------------------------------------
// FFI (include/bp_layer.h, include/bp_sharded.h)
// ---------------------------------------------------------------------------------------------------
#[repr(C)]
pub struct BpLayer { _private: [u8; 0] }
#[repr(C)]
pub struct BpGroup { _private: [u8; 0] }
#[repr(C)]
pub struct BpSharded { _private: [u8; 0] }
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
    fn bp_layer_set_tree(layer: *mut BpLayer, min_depth: u32, keys: *const u64, ids: *const u32, m: usize, sorted: c_int) -> c_int;
    fn bp_layer_min_depth(layer: *const BpLayer, min_depth: *mut u32) -> c_int;
    fn bp_layer_test_box_batch(layer: *mut BpLayer, sys_min: *const f32, sys_max: *const f32, boxes: *const f32,
                               n_queries: usize, has_max_depth: c_int, max_depth: u32, rows: *mut *const u32,
                               n_rows: *mut usize) -> c_int;

    fn bp_group_unique_id(out_128_bytes: *mut c_void) -> c_int;
    fn bp_group_create_rccl(out: *mut *mut BpGroup, rank: c_int, world: c_int, unique_id: *const c_void, device: c_int) -> c_int;
    fn bp_group_create_local(out: *mut *mut BpGroup, world: c_int) -> c_int;
    fn bp_group_destroy(group: *mut BpGroup) -> c_int;
    fn bp_sharded_create(out: *mut *mut BpSharded, group: *mut BpGroup, rank: c_int, index_kind: c_int, min_depth: u32,
                         split_depth: u32, device: c_int, id_count: u64) -> c_int;
    fn bp_sharded_destroy(sh: *mut BpSharded) -> c_int;
    fn bp_sharded_last_error(sh: *const BpSharded) -> *const c_char;
    fn bp_sharded_clear(sh: *mut BpSharded) -> c_int;
    fn bp_sharded_extend(sh: *mut BpSharded, sys_min: *const f32, sys_max: *const f32, bounds: *const f32,
                         ids: *const u32, n: usize) -> c_int;
    fn bp_sharded_par_scan(sh: *mut BpSharded, pair_order: c_int, d_pairs: *mut *const u32, n_pairs: *mut usize) -> c_int;
    fn bp_sharded_copy_pairs(sh: *mut BpSharded, host_dst: *mut u32, n_pairs: usize) -> c_int;
}

fn cstr(p: *const c_char) -> String {
    if p.is_null() { String::new() } else { unsafe { std::ffi::CStr::from_ptr(p) }.to_string_lossy().into_owned() }
}
/// The reference has no `Result` on this path (its only failure mode is a panic): a device failure panics.
fn check(rc: c_int, layer: *const BpLayer) {
    if rc != 0 {
        panic!("broadphase_amd: {}: {}", cstr(unsafe { bp_status_string(rc) }), cstr(unsafe { bp_layer_last_error(layer) }));
    }
}

// ---------------------------------------------------------------------------------------------------
// ObjectID (src/traits.rs:6-16)
// ---------------------------------------------------------------------------------------------------
pub trait ObjectID: Copy + Clone + Hash + Ord + Send + Sync + Debug {}
impl<T: Copy + Clone + Hash + Ord + Send + Sync + Debug> ObjectID for T {}

// ---------------------------------------------------------------------------------------------------
// SpatialIndex and the three index types (src/index.rs:38-63, 293-295)
// ---------------------------------------------------------------------------------------------------
pub trait SpatialIndex: Clone + Copy + Default + Ord + Send + Debug {
    type Diff: cgmath::VectorSpace<Scalar = u32>;
    type Point: Copy + EuclideanSpace<Diff = Self::Diff, Scalar = u32>;
    type SubdivideResult: AsRef<[Self]>;

    fn clamp_depth(depth: u32) -> u32;
    fn origin(self) -> Self::Point;
    fn depth(self) -> u32;
    fn set_origin(self, origin: Self::Point) -> Self;
    fn set_depth(self, depth: u32) -> Self;
    /// cells of `depth + 1` in sorted order; `None` at the depth limit
    fn subdivide(self) -> Option<Self::SubdivideResult>;
    fn overlaps(self, other: Self) -> bool;
    fn same_cell_at_depth(lhs: Self, rhs: Self, depth: u32) -> bool;

    /// index kind of the C ABI (`bp_index_kind`) and the value as it crosses it (zero-extended to 64 bits)
    #[doc(hidden)]
    const KIND: c_int;
    #[doc(hidden)]
    const DIM: usize;
    #[doc(hidden)]
    fn from_raw(value: u64) -> Self;
    #[doc(hidden)]
    fn raw(self) -> u64;
}

/// bit k of the top `axis_bits` of `v` -> bit `dim * k` (one Morton lane)
fn spread_axis(v: u32, axis_bits: u32, dim: u32) -> u64 {
    let a = (v >> (32 - axis_bits)) as u64;
    let mut out = 0u64;
    for k in 0..axis_bits {
        out |= ((a >> k) & 1) << (dim * k);
    }
    out
}
/// inverse of `spread_axis`
fn gather_axis(m: u64, axis_bits: u32, dim: u32) -> u32 {
    let mut a = 0u64;
    for k in 0..axis_bits {
        a |= ((m >> (dim * k)) & 1) << k;
    }
    (a as u32) << (32 - axis_bits)
}

macro_rules! index_common {
    ($name:ident, $prim:ty, $dim:expr, $depth_bits:expr, $axis_bits:expr) => {
        #[derive(Clone, Copy, Default, Eq, Ord, PartialEq, PartialOrd, Hash)]
        #[cfg_attr(feature = "serde", derive(Deserialize, Serialize))]  // src/index.rs:68: a newtype over the raw value
        #[repr(transparent)]
        pub struct $name(pub(crate) $prim);

        impl $name {
            const DIM_: u32 = $dim;
            const DEPTH_BITS: u32 = $depth_bits;
            const AXIS_BITS: u32 = $axis_bits;
            const ORIGIN_BITS: u32 = $dim * $axis_bits;
            const TOP: u32 = $dim * $axis_bits + $depth_bits;
            fn depth_mask() -> u64 { (1u64 << Self::DEPTH_BITS) - 1 }
            fn origin_mask() -> u64 { ((1u64 << Self::ORIGIN_BITS) - 1) << Self::DEPTH_BITS }
            fn level_mask(depth: u32) -> u64 {
                if depth == 0 { 0 } else { ((1u64 << (Self::DIM_ * depth)) - 1) << (Self::TOP - Self::DIM_ * depth) }
            }
            fn morton(self) -> u64 { ((self.0 as u64) & Self::origin_mask()) >> Self::DEPTH_BITS }
            fn with_morton(self, m: u64) -> Self {
                $name((((self.0 as u64) & !Self::origin_mask()) | ((m << Self::DEPTH_BITS) & Self::origin_mask())) as $prim)
            }
            fn child(self, c: u64, depth: u32) -> Self {
                let shift = Self::TOP - Self::DIM_ * (depth + 1);
                <Self as SpatialIndex>::set_depth($name(((self.0 as u64) | (c << shift)) as $prim), depth + 1)
            }
        }

        impl Debug for $name {
            fn fmt(&self, f: &mut std::fmt::Formatter) -> std::fmt::Result {
                write!(f, "{}(depth {}, morton {:#x})", stringify!($name), <Self as SpatialIndex>::depth(*self), self.morton())
            }
        }
    };
}
macro_rules! index_trait_common {
    ($name:ident, $prim:ty, $kind:expr) => {
        fn clamp_depth(depth: u32) -> u32 { std::cmp::min(depth, Self::AXIS_BITS) }
        fn depth(self) -> u32 { ((self.0 as u64) & Self::depth_mask()) as u32 }
        fn set_depth(self, depth: u32) -> Self {
            $name((((self.0 as u64) & !Self::depth_mask()) | (Self::clamp_depth(depth) as u64 & Self::depth_mask())) as $prim)
        }
        fn overlaps(self, other: Self) -> bool {
            Self::same_cell_at_depth(self, other, std::cmp::min(self.depth(), other.depth()))
        }
        fn same_cell_at_depth(lhs: Self, rhs: Self, depth: u32) -> bool {
            ((lhs.0 as u64) ^ (rhs.0 as u64)) & Self::level_mask(depth) == 0
        }
        const KIND: c_int = $kind;
        const DIM: usize = Self::DIM_ as usize;
        fn from_raw(value: u64) -> Self { $name(value as $prim) }
        fn raw(self) -> u64 { self.0 as u64 }
    };
}

index_common!(Index32_2D, u32, 2, 4, 14);
index_common!(Index64_2D, u64, 2, 5, 29);
index_common!(Index64_3D, u64, 3, 5, 19);

macro_rules! index_2d {
    ($name:ident, $prim:ty, $kind:expr) => {
        impl SpatialIndex for $name {
            type Diff = Vector2<u32>;
            type Point = Point2<u32>;
            type SubdivideResult = [Self; 4];
            index_trait_common!($name, $prim, $kind);
            fn origin(self) -> Point2<u32> {
                let m = self.morton();
                Point2::new(gather_axis(m, Self::AXIS_BITS, 2), gather_axis(m >> 1, Self::AXIS_BITS, 2))
            }
            fn set_origin(self, origin: Point2<u32>) -> Self {
                self.with_morton(spread_axis(origin.x, Self::AXIS_BITS, 2) | (spread_axis(origin.y, Self::AXIS_BITS, 2) << 1))
            }
            fn subdivide(self) -> Option<[Self; 4]> {
                let d = self.depth();
                if d < Self::AXIS_BITS { Some([self.child(0, d), self.child(1, d), self.child(2, d), self.child(3, d)]) } else { None }
            }
        }
    };
}
index_2d!(Index32_2D, u32, 0);
index_2d!(Index64_2D, u64, 1);

impl SpatialIndex for Index64_3D {
    type Diff = Vector3<u32>;
    type Point = Point3<u32>;
    type SubdivideResult = [Self; 8];
    index_trait_common!(Index64_3D, u64, 2);
    fn origin(self) -> Point3<u32> {
        let m = self.morton();
        Point3::new(gather_axis(m, Self::AXIS_BITS, 3), gather_axis(m >> 1, Self::AXIS_BITS, 3), gather_axis(m >> 2, Self::AXIS_BITS, 3))
    }
    fn set_origin(self, origin: Point3<u32>) -> Self {
        self.with_morton(spread_axis(origin.x, Self::AXIS_BITS, 3)
            | (spread_axis(origin.y, Self::AXIS_BITS, 3) << 1)
            | (spread_axis(origin.z, Self::AXIS_BITS, 3) << 2))
    }
    fn subdivide(self) -> Option<[Self; 8]> {
        let d = self.depth();
        if d < Self::AXIS_BITS {
            Some([self.child(0, d), self.child(1, d), self.child(2, d), self.child(3, d),
                  self.child(4, d), self.child(5, d), self.child(6, d), self.child(7, d)])
        } else {
            None
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Bounds<Point> (src/geom.rs:84-133): inclusive min / max over any cgmath point type
// ---------------------------------------------------------------------------------------------------
#[derive(Copy, Clone, Debug, PartialEq)]
#[cfg_attr(feature = "serde", derive(Deserialize, Serialize))]  // src/geom.rs:83
pub struct Bounds<Point> {
    pub min: Point,
    pub max: Point,
}

impl<Point> Bounds<Point>
where
    Point: EuclideanSpace + Array<Element = <Point as EuclideanSpace>::Scalar> + Copy,
{
    pub fn new(min: Point, max: Point) -> Self { Bounds { min, max } }

    pub fn overlaps(self, other: Bounds<Point>) -> bool {
        (0..Point::len()).all(|i| !(self.min[i] > other.max[i] || self.max[i] < other.min[i]))
    }

    pub fn contains(self, other: Bounds<Point>) -> bool {
        (0..Point::len()).all(|i| !(self.min[i] > other.min[i] || self.max[i] < other.max[i]))
    }

    pub fn center(self) -> Point { self.min.midpoint(self.max) }
}

// ---------------------------------------------------------------------------------------------------
// Layer<Index, ID> (src/layer.rs:42-68)
// ---------------------------------------------------------------------------------------------------
/// One row of a result list as the ABI lays it out (`(id, id_)`, two consecutive `u32`)
#[repr(C)]
#[derive(Copy, Clone, Debug, PartialEq, Eq, PartialOrd, Ord, Hash)]
pub struct Pair {
    pub id: u32,
    pub id_: u32,
}

/// Rust leaves tuple layout unspecified; when `(u32, u32)` happens to be the ABI's row layout the rows are copied
/// with one `memcpy` instead of element by element.
fn tuple_rows_are_abi_rows() -> bool {
    let probe: (u32, u32) = (1, 2);
    let base = &probe as *const (u32, u32) as usize;
    std::mem::size_of::<(u32, u32)>() == 8
        && (&probe.0 as *const u32 as usize) - base == 0
        && (&probe.1 as *const u32 as usize) - base == 4
}

pub struct Layer<Index, ID = u32>
where
    Index: SpatialIndex,
    ID: ObjectID,
{
    handle: *mut BpLayer,
    collisions: Vec<(ID, ID)>,
    /// host mirror of the tree for `iter()`; refreshed lazily (the tree itself lives in HBM)
    tree_mirror: UnsafeCell<Vec<(Index, ID)>>,
    mirror_valid: Cell<bool>,
}

unsafe impl<Index: SpatialIndex, ID: ObjectID> Send for Layer<Index, ID> {}

impl<Index: SpatialIndex> Layer<Index, u32> {
    fn touched(&mut self) { self.mirror_valid.set(false); }

    /// `Layer::iter` (`src/layer.rs:79-81`): `(index, id)` in current order
    pub fn iter(&self) -> std::slice::Iter<'_, (Index, u32)> {
        if !self.mirror_valid.get() {
            let (mut k, mut i, mut m, mut s) = (std::ptr::null(), std::ptr::null(), 0usize, 0 as c_int);
            check(unsafe { bp_layer_tree(self.handle, &mut k, &mut i, &mut m, &mut s) }, self.handle);
            // SAFETY: `&self` methods never hand out `&mut` to the mirror, and every `&mut self` method invalidates it
            let mirror = unsafe { &mut *self.tree_mirror.get() };
            mirror.clear();
            if m > 0 {
                let (ks, is) = unsafe { (std::slice::from_raw_parts(k, m), std::slice::from_raw_parts(i, m)) };
                mirror.extend(ks.iter().zip(is.iter()).map(|(&key, &id)| (Index::from_raw(key), id)));
            }
            self.mirror_valid.set(true);
        }
        unsafe { (*self.tree_mirror.get()).iter() }
    }

    /// `Layer::clear` (`src/layer.rs:84-88`)
    pub fn clear(&mut self) {
        self.touched();
        check(unsafe { bp_layer_clear(self.handle) }, self.handle)
    }

    /// `Layer::extend` (`src/layer.rs:94-121`): objects outside `system_bounds` are skipped silently
    pub fn extend<Iter, Point_>(&mut self, system_bounds: Bounds<Point_>, objects: Iter)
    where
        Iter: std::iter::Iterator<Item = (Bounds<Point_>, u32)>,
        Point_: EuclideanSpace<Scalar = f32> + Array<Element = f32> + Copy,
    {
        self.touched();
        let dim = Point_::len();
        assert_eq!(dim, Index::DIM, "point dimension differs from the index type's");
        let reserve = match objects.size_hint() { (_, Some(upper)) => upper, (lower, None) => lower };
        let mut flat: Vec<f32> = Vec::with_capacity(reserve * 2 * dim);
        let mut ids: Vec<u32> = Vec::with_capacity(reserve);
        for (bounds, id) in objects {
            for a in 0..dim { flat.push(bounds.min[a]); }
            for a in 0..dim { flat.push(bounds.max[a]); }
            ids.push(id);
        }
        let (mut smin, mut smax) = ([0f32; 3], [0f32; 3]);
        for a in 0..dim { smin[a] = system_bounds.min[a]; smax[a] = system_bounds.max[a]; }
        check(unsafe { bp_layer_extend_host(self.handle, smin.as_ptr(), smax.as_ptr(), flat.as_ptr(), ids.as_ptr(), ids.len()) },
              self.handle)
    }

    /// `Layer::merge` (`src/layer.rs:127-138`)
    pub fn merge(&mut self, other: &Layer<Index, u32>) {
        self.touched();
        check(unsafe { bp_layer_merge(self.handle, other.handle) }, self.handle)
    }

    /// `Layer::sort` (`src/layer.rs:159-165`) / `Layer::par_sort` (`:146-152`): one device implementation
    pub fn sort(&mut self) { self.touched(); check(unsafe { bp_layer_sort(self.handle) }, self.handle) }
    pub fn par_sort(&mut self) { self.touched(); check(unsafe { bp_layer_par_sort(self.handle) }, self.handle) }

    fn collect(&mut self, p: *const u32, n: usize) -> &Vec<(u32, u32)> {
        self.collisions.clear();
        if n > 0 {
            if tuple_rows_are_abi_rows() {
                self.collisions.reserve(n);
                unsafe {
                    std::ptr::copy_nonoverlapping(p as *const (u32, u32), self.collisions.as_mut_ptr(), n);
                    self.collisions.set_len(n);
                }
            } else {
                let rows = unsafe { std::slice::from_raw_parts(p, 2 * n) };
                self.collisions.extend(rows.chunks_exact(2).map(|c| (c[0], c[1])));
            }
        }
        &self.collisions
    }

    /// `Layer::scan` (`src/layer.rs:449-453`)
    pub fn scan<'a>(&'a mut self) -> &'a Vec<(u32, u32)> {
        self.touched();
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { bp_layer_scan(self.handle, &mut p, &mut n) }, self.handle);
        self.collect(p, n)
    }

    /// `Layer::par_scan` (`src/layer.rs:482-488`)
    pub fn par_scan<'a>(&'a mut self) -> &'a Vec<(u32, u32)> {
        self.touched();
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { bp_layer_par_scan(self.handle, &mut p, &mut n) }, self.handle);
        self.collect(p, n)
    }

    /// the same list without a copy: the pinned host rows the library owns, valid until the next `&mut self` call
    pub fn par_scan_rows<'a>(&'a mut self) -> &'a [Pair] {
        self.touched();
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { bp_layer_par_scan(self.handle, &mut p, &mut n) }, self.handle);
        if n == 0 { &[] } else { unsafe { std::slice::from_raw_parts(p as *const Pair, n) } }
    }

    /// `Layer::scan_filtered` (`src/layer.rs:460-477`): the filter sees every raw pair before dedup, in the reference's order
    pub fn scan_filtered<'a, F>(&'a mut self, mut filter: F) -> &'a Vec<(u32, u32)>
    where
        F: FnMut(u32, u32) -> bool,
    {
        extern "C" fn thunk<F: FnMut(u32, u32) -> bool>(a: u32, b: u32, user: *mut c_void) -> c_int {
            let f = unsafe { &mut *(user as *mut F) };
            if f(a, b) { 1 } else { 0 }
        }
        self.touched();
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { bp_layer_scan_filtered(self.handle, thunk::<F>, &mut filter as *mut F as *mut c_void, &mut p, &mut n) },
              self.handle);
        self.collect(p, n)
    }

    /// `Layer::par_scan_filtered` (`src/layer.rs:493-520`)
    pub fn par_scan_filtered<'a, F>(&'a mut self, mut filter: F) -> &'a Vec<(u32, u32)>
    where
        F: Copy + Send + Sync + FnMut(u32, u32) -> bool,
    {
        extern "C" fn thunk<F: FnMut(u32, u32) -> bool>(a: u32, b: u32, user: *mut c_void) -> c_int {
            let f = unsafe { &mut *(user as *mut F) };
            if f(a, b) { 1 } else { 0 }
        }
        self.touched();
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { bp_layer_par_scan_filtered(self.handle, thunk::<F>, &mut filter as *mut F as *mut c_void, &mut p, &mut n) },
              self.handle);
        self.collect(p, n)
    }

    /// `Layer::test_box` (`src/layer.rs:294-313`): ids whose cells overlap `test_bounds`, sorted, duplicate-free
    pub fn test_box<Point_>(&mut self, system_bounds: Bounds<Point_>, test_bounds: Bounds<Point_>, max_depth: Option<u32>) -> Vec<u32>
    where
        Point_: EuclideanSpace<Scalar = f32> + Array<Element = f32> + Copy,
    {
        self.touched();
        let dim = Point_::len();
        assert_eq!(dim, Index::DIM);
        let (mut smin, mut smax, mut row) = ([0f32; 3], [0f32; 3], [0f32; 6]);
        for a in 0..dim {
            smin[a] = system_bounds.min[a]; smax[a] = system_bounds.max[a];
            row[a] = test_bounds.min[a]; row[dim + a] = test_bounds.max[a];
        }
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { bp_layer_test_box_batch(self.handle, smin.as_ptr(), smax.as_ptr(), row.as_ptr(), 1,
                                               max_depth.is_some() as c_int, max_depth.unwrap_or(0), &mut p, &mut n) }, self.handle);
        let rows = unsafe { std::slice::from_raw_parts(p, 2 * n) };
        rows.chunks_exact(2).map(|c| c[1]).collect()
    }
}

impl<Index: SpatialIndex, ID: ObjectID> Drop for Layer<Index, ID> {
    fn drop(&mut self) { unsafe { bp_layer_destroy(self.handle); } }
}

/// `impl Clone` (`src/layer.rs:595-616`): persistent state only
impl<Index: SpatialIndex> Clone for Layer<Index, u32> {
    fn clone(&self) -> Self {
        let mut h = std::ptr::null_mut();
        check(unsafe { bp_layer_clone(self.handle, &mut h) }, self.handle);
        Layer { handle: h, collisions: Vec::new(), tree_mirror: UnsafeCell::new(Vec::new()), mirror_valid: Cell::new(false) }
    }
}

/// `impl PartialEq` (`src/layer.rs:576-586`): `min_depth`, entries in current order, `sorted`
impl<Index: SpatialIndex> PartialEq for Layer<Index, u32> {
    fn eq(&self, other: &Self) -> bool {
        let mut eq = 0;
        check(unsafe { bp_layer_equals(self.handle, other.handle, &mut eq) }, self.handle);
        eq != 0
    }
}
impl<Index: SpatialIndex> Eq for Layer<Index, u32> {}

// ---------------------------------------------------------------------------------------------------
// serde (feature "serde"): the reference derives Serialize / Deserialize on Layer with every temporary skipped
// (`src/layer.rs:41-68`), so the serialized form is exactly `min_depth: u32` + `tree: (Vec<(Index, ID)>, bool)` -- what
// `data/src/lib.rs:28-49` stores in a BR_SCENE and what `tests/test_layer.rs:19-40` loads and compares.  Here the tree
// lives in HBM: it is read through `bp_layer_tree` and written back through `bp_layer_set_tree`.
// ---------------------------------------------------------------------------------------------------
#[cfg(feature = "serde")]
mod serde_impl {
    use super::*;
    use serde::de::{self, Deserialize, Deserializer, MapAccess, SeqAccess, Visitor};
    use serde::ser::{Serialize, SerializeStruct, Serializer};

    impl<Index> Serialize for Layer<Index, u32>
    where
        Index: SpatialIndex + Serialize,
    {
        fn serialize<S: Serializer>(&self, serializer: S) -> Result<S::Ok, S::Error> {
            let mut min_depth = 0u32;
            check(unsafe { bp_layer_min_depth(self.handle, &mut min_depth) }, self.handle);
            let (mut k, mut i, mut m, mut sorted) = (std::ptr::null(), std::ptr::null(), 0usize, 0 as c_int);
            check(unsafe { bp_layer_tree(self.handle, &mut k, &mut i, &mut m, &mut sorted) }, self.handle);
            let (keys, ids) = unsafe { (std::slice::from_raw_parts(k, m), std::slice::from_raw_parts(i, m)) };
            let tree: (Vec<(Index, u32)>, bool) =
                (keys.iter().zip(ids.iter()).map(|(&key, &id)| (Index::from_raw(key), id)).collect(), sorted != 0);
            let mut st = serializer.serialize_struct("Layer", 2)?;
            st.serialize_field("min_depth", &min_depth)?;
            st.serialize_field("tree", &tree)?;
            st.end()
        }
    }

    impl<'de, Index> Deserialize<'de> for Layer<Index, u32>
    where
        Index: SpatialIndex + Deserialize<'de>,
    {
        fn deserialize<D: Deserializer<'de>>(deserializer: D) -> Result<Self, D::Error> {
            struct LayerVisitor<Index>(std::marker::PhantomData<Index>);
            fn build<Index: SpatialIndex>(min_depth: u32, tree: (Vec<(Index, u32)>, bool)) -> Layer<Index, u32> {
                let mut layer: Layer<Index, u32> = LayerBuilder::new().with_min_depth(min_depth).build();
                let keys: Vec<u64> = tree.0.iter().map(|e| e.0.raw()).collect();
                let ids: Vec<u32> = tree.0.iter().map(|e| e.1).collect();
                check(unsafe { bp_layer_set_tree(layer.handle, min_depth, keys.as_ptr(), ids.as_ptr(), keys.len(), tree.1 as c_int) },
                      layer.handle);
                layer.touched();
                layer
            }
            impl<'de, Index: SpatialIndex + Deserialize<'de>> Visitor<'de> for LayerVisitor<Index> {
                type Value = Layer<Index, u32>;
                fn expecting(&self, f: &mut std::fmt::Formatter) -> std::fmt::Result { f.write_str("struct Layer") }
                fn visit_seq<A: SeqAccess<'de>>(self, mut seq: A) -> Result<Self::Value, A::Error> {  // bincode: positional
                    let min_depth: u32 = seq.next_element()?.ok_or_else(|| de::Error::invalid_length(0, &self))?;
                    let tree: (Vec<(Index, u32)>, bool) = seq.next_element()?.ok_or_else(|| de::Error::invalid_length(1, &self))?;
                    Ok(build(min_depth, tree))
                }
                fn visit_map<A: MapAccess<'de>>(self, mut map: A) -> Result<Self::Value, A::Error> {  // self-describing formats
                    let (mut min_depth, mut tree): (Option<u32>, Option<(Vec<(Index, u32)>, bool)>) = (None, None);
                    while let Some(key) = map.next_key::<String>()? {
                        match key.as_str() {
                            "min_depth" => min_depth = Some(map.next_value()?),
                            "tree" => tree = Some(map.next_value()?),
                            _ => { let _: de::IgnoredAny = map.next_value()?; }
                        }
                    }
                    Ok(build(min_depth.ok_or_else(|| de::Error::missing_field("min_depth"))?,
                             tree.ok_or_else(|| de::Error::missing_field("tree"))?))
                }
            }
            deserializer.deserialize_struct("Layer", &["min_depth", "tree"], LayerVisitor::<Index>(std::marker::PhantomData))
        }
    }
}

/// `LayerBuilder` (`src/layer.rs:619-696`)
#[derive(Default)]
pub struct LayerBuilder {
    min_depth: u32,
    index_capacity: Option<usize>,
    collision_capacity: Option<usize>,
    test_capacity: Option<usize>,
    device: Option<i32>,
}

impl LayerBuilder {
    pub fn new() -> Self { Self::default() }
    pub fn with_min_depth(&mut self, depth: u32) -> &mut Self { self.min_depth = depth; self }
    pub fn with_index_capacity(&mut self, capacity: usize) -> &mut Self { self.index_capacity = Some(capacity); self }
    pub fn with_collision_capacity(&mut self, capacity: usize) -> &mut Self { self.collision_capacity = Some(capacity); self }
    /// accepted for source compatibility; the batched queries size their own scratch
    pub fn with_test_capacity(&mut self, capacity: usize) -> &mut Self { self.test_capacity = Some(capacity); self }
    /// not in the reference: the HIP device ordinal (default: the current device)
    pub fn with_device(&mut self, device: i32) -> &mut Self { self.device = Some(device); self }

    pub fn build<Index, ID>(&self) -> Layer<Index, ID>
    where
        Index: SpatialIndex,
        ID: ObjectID,
    {
        let mut h = std::ptr::null_mut();
        check(unsafe { bp_layer_create(&mut h, Index::KIND, self.min_depth, self.index_capacity.unwrap_or(0),
                                       self.collision_capacity.unwrap_or(0), self.device.unwrap_or(-1)) }, std::ptr::null());
        Layer { handle: h, collisions: Vec::new(), tree_mirror: UnsafeCell::new(Vec::new()), mirror_valid: Cell::new(false) }
    }
}

// ---------------------------------------------------------------------------------------------------
// Multi-GPU (include/bp_sharded.h): one rank of a Layer whose objects shard across the GPUs of a node
// ---------------------------------------------------------------------------------------------------
pub struct Group { handle: *mut BpGroup }
unsafe impl Send for Group {}
unsafe impl Sync for Group {}

impl Group {
    /// rank 0 makes the id, the host hands the 128 bytes to every rank (its own channel: MPI, a file, a socket)
    pub fn unique_id() -> [u8; 128] {
        let mut id = [0u8; 128];
        let rc = unsafe { bp_group_unique_id(id.as_mut_ptr() as *mut c_void) };
        assert!(rc == 0, "broadphase_amd: {}", cstr(unsafe { bp_status_string(rc) }));
        id
    }
    /// one process (or thread) per GPU over RCCL
    pub fn rccl(rank: i32, world: i32, unique_id: &[u8; 128], device: i32) -> Group {
        let mut h = std::ptr::null_mut();
        let rc = unsafe { bp_group_create_rccl(&mut h, rank, world, unique_id.as_ptr() as *const c_void, device) };
        assert!(rc == 0, "broadphase_amd: {}", cstr(unsafe { bp_status_string(rc) }));
        Group { handle: h }
    }
    /// `world` ranks inside this process, one host thread each
    pub fn local(world: i32) -> Group {
        let mut h = std::ptr::null_mut();
        let rc = unsafe { bp_group_create_local(&mut h, world) };
        assert!(rc == 0, "broadphase_amd: {}", cstr(unsafe { bp_status_string(rc) }));
        Group { handle: h }
    }
}
impl Drop for Group {
    fn drop(&mut self) { unsafe { bp_group_destroy(self.handle); } }
}

/// This rank's share of the multi-GPU `Layer`.  Every method is collective over the group.
pub struct ShardedLayer<'g, Index: SpatialIndex> {
    handle: *mut BpSharded,
    rows: Vec<Pair>,
    _group: &'g Group,
    _index: std::marker::PhantomData<Index>,
}

impl<'g, Index: SpatialIndex> ShardedLayer<'g, Index> {
    /// `id_count`: ids lie in `[0, id_count)` (0: unknown)
    pub fn new(group: &'g Group, rank: i32, min_depth: u32, device: i32, id_count: u64) -> Self {
        let mut h = std::ptr::null_mut();
        let rc = unsafe { bp_sharded_create(&mut h, group.handle, rank, Index::KIND, min_depth, 0, device, id_count) };
        assert!(rc == 0, "broadphase_amd: {}", cstr(unsafe { bp_status_string(rc) }));
        ShardedLayer { handle: h, rows: Vec::new(), _group: group, _index: std::marker::PhantomData }
    }
    fn check(&self, rc: c_int) {
        if rc != 0 {
            panic!("broadphase_amd: {}: {}", cstr(unsafe { bp_status_string(rc) }), cstr(unsafe { bp_sharded_last_error(self.handle) }));
        }
    }
    pub fn clear(&mut self) { let rc = unsafe { bp_sharded_clear(self.handle) }; self.check(rc) }
    /// this rank's own objects (the union over ranks is the scene)
    pub fn extend<Iter, Point_>(&mut self, system_bounds: Bounds<Point_>, objects: Iter)
    where
        Iter: std::iter::Iterator<Item = (Bounds<Point_>, u32)>,
        Point_: EuclideanSpace<Scalar = f32> + Array<Element = f32> + Copy,
    {
        let dim = Point_::len();
        assert_eq!(dim, Index::DIM);
        let (mut flat, mut ids): (Vec<f32>, Vec<u32>) = (Vec::new(), Vec::new());
        for (bounds, id) in objects {
            for a in 0..dim { flat.push(bounds.min[a]); }
            for a in 0..dim { flat.push(bounds.max[a]); }
            ids.push(id);
        }
        let (mut smin, mut smax) = ([0f32; 3], [0f32; 3]);
        for a in 0..dim { smin[a] = system_bounds.min[a]; smax[a] = system_bounds.max[a]; }
        let rc = unsafe { bp_sharded_extend(self.handle, smin.as_ptr(), smax.as_ptr(), flat.as_ptr(), ids.as_ptr(), ids.len()) };
        self.check(rc)
    }
    /// this rank's first-id range of `Layer::par_scan`'s list over the union scene (rank order = list order)
    pub fn par_scan<'a>(&'a mut self) -> &'a [Pair] {
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        let rc = unsafe { bp_sharded_par_scan(self.handle, 0, &mut p, &mut n) };
        self.check(rc);
        self.rows.clear();
        self.rows.resize(n, Pair { id: 0, id_: 0 });
        let rc = unsafe { bp_sharded_copy_pairs(self.handle, self.rows.as_mut_ptr() as *mut u32, n) };
        self.check(rc);
        &self.rows
    }
}
impl<'g, Index: SpatialIndex> Drop for ShardedLayer<'g, Index> {
    fn drop(&mut self) { unsafe { bp_sharded_destroy(self.handle); } }
}
