// broadphase/layer.hpp -- C++ host-side mirror of the reference's `Layer<Index, ID>` and
// `LayerBuilder` (zvxryb/broadphase-rs src/layer.rs:42-68, 619-696) over the C ABI of
// include/bp_layer.h.  Header-only; link with -lbroadphase_amd.
//
// The reference is Rust and no Rust toolchain exists in this pipeline, so this facade is the
// tested drop-in host layer (tests/cpp/test_facade.cpp); the Rust shim with the same shape is
// in INTEGRATION.md / rust/src/lib.rs.  Method names, argument meaning and error behaviour follow
// the reference: objects outside the system bounds are skipped silently (layer.rs:108-111),
// results are borrowed until the next mutating call (layer.rs:449-451), par_* are the same
// implementation as the serial names, a failing bp_* call becomes an exception (the reference has
// no Result on this path; its only failure mode is a panic).
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../bp_layer.h"

namespace broadphase {

// SpatialIndex implementations (src/index.rs:293-295)
struct Index32_2D { static constexpr int kind = BP_INDEX32_2D; static constexpr int dim = 2; };
struct Index64_2D { static constexpr int kind = BP_INDEX64_2D; static constexpr int dim = 2; };
struct Index64_3D { static constexpr int kind = BP_INDEX64_3D; static constexpr int dim = 3; };

// Bounds<Point> (src/geom.rs:84-87): inclusive min / max, DIM f32 each
template <int DIM> struct Bounds {
    float min[DIM];
    float max[DIM];
};

class Error : public std::runtime_error {
   public:
    Error(int status, const std::string& what) : std::runtime_error(std::string(bp_status_string(status)) + ": " + what), status_(status) {}
    int status() const { return status_; }

   private:
    int status_;
};

// borrowed view of the pair list: (id, id_) interleaved, valid until the next mutating call
struct PairView {
    const uint32_t* data = nullptr;
    size_t len = 0;
    std::pair<uint32_t, uint32_t> operator[](size_t i) const { return {data[2 * i], data[2 * i + 1]}; }
    size_t size() const { return len; }
};

template <class Index, class ID = uint32_t> class Layer {
    static_assert(sizeof(ID) == 4, "the device path carries 32-bit object IDs");

   public:
    Layer(uint32_t min_depth = 0, size_t index_capacity = 0, size_t collision_capacity = 0, int device = -1) {
        int rc = bp_layer_create(&h_, Index::kind, min_depth, index_capacity, collision_capacity, device);
        if (rc != BP_OK) throw Error(rc, bp_layer_last_error(nullptr));
    }
    ~Layer() { bp_layer_destroy(h_); }
    Layer(const Layer& o) { check(bp_layer_clone(o.h_, &h_), o.h_); }  // impl Clone (layer.rs:595-616)
    Layer& operator=(const Layer& o) {
        if (this != &o) {
            bp_layer_t* n = nullptr;
            check(bp_layer_clone(o.h_, &n), o.h_);
            bp_layer_destroy(h_);
            h_ = n;
        }
        return *this;
    }
    Layer(Layer&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    Layer& operator=(Layer&& o) noexcept {
        if (this != &o) { bp_layer_destroy(h_); h_ = o.h_; o.h_ = nullptr; }
        return *this;
    }

    // Layer::clear (layer.rs:84-88)
    void clear() { check(bp_layer_clear(h_)); }

    // Layer::extend (layer.rs:94-121): objects = (bounds, id) pairs; host memory
    void extend(const Bounds<Index::dim>& system_bounds, const std::vector<std::pair<Bounds<Index::dim>, ID>>& objects) {
        std::vector<float> b(objects.size() * 2 * Index::dim);
        std::vector<uint32_t> ids(objects.size());
        for (size_t i = 0; i < objects.size(); ++i) {
            for (int a = 0; a < Index::dim; ++a) {
                b[i * 2 * Index::dim + a] = objects[i].first.min[a];
                b[i * 2 * Index::dim + Index::dim + a] = objects[i].first.max[a];
            }
            ids[i] = (uint32_t)objects[i].second;
        }
        extend(system_bounds, b.data(), ids.data(), objects.size());
    }
    // same, from SoA arrays (host or device pointers; detected)
    void extend(const Bounds<Index::dim>& system_bounds, const float* bounds, const uint32_t* ids, size_t n) {
        check(bp_layer_extend(h_, system_bounds.min, system_bounds.max, bounds, ids, n));
    }

    // Layer::merge (layer.rs:127-138)
    void merge(const Layer& other) { check(bp_layer_merge(h_, other.h_)); }

    // Layer::sort / par_sort (layer.rs:146-165)
    void sort() { check(bp_layer_sort(h_)); }
    void par_sort() { check(bp_layer_par_sort(h_)); }

    // Layer::scan / par_scan (layer.rs:449-453, 482-488)
    PairView scan() { PairView v; check(bp_layer_scan(h_, &v.data, &v.len)); return v; }
    PairView par_scan() { PairView v; check(bp_layer_par_scan(h_, &v.data, &v.len)); return v; }
    // result left in HBM
    PairView scan_device() { PairView v; check(bp_layer_scan_device(h_, &v.data, &v.len)); return v; }

    // Layer::scan_filtered / par_scan_filtered (layer.rs:460-477, 493-520)
    template <class F> PairView scan_filtered(F&& filter) { return filtered(&bp_layer_scan_filtered, std::forward<F>(filter)); }
    template <class F> PairView par_scan_filtered(F&& filter) { return filtered(&bp_layer_par_scan_filtered, std::forward<F>(filter)); }

    // Layer::test_box (layer.rs:294-313): sorted duplicate-free ids whose cells overlap test_bounds;
    // max_depth < 0 stands for None.  (One query per call here; bp_layer_test_box_batch takes many.)
    std::vector<ID> test_box(const Bounds<Index::dim>& system_bounds, const Bounds<Index::dim>& test_bounds, int max_depth = -1) {
        float box[2 * Index::dim];
        for (int a = 0; a < Index::dim; ++a) { box[a] = test_bounds.min[a]; box[Index::dim + a] = test_bounds.max[a]; }
        const uint32_t* rows = nullptr;
        size_t n = 0;
        check(bp_layer_test_box_batch(h_, system_bounds.min, system_bounds.max, box, 1, max_depth >= 0 ? 1 : 0,
                                      max_depth >= 0 ? (uint32_t)max_depth : 0u, &rows, &n));
        std::vector<ID> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = (ID)rows[2 * i + 1];
        return out;
    }

    // Layer::test_ray (layer.rs:327-351): ids of the cells the ray origin + t * direction, t in [range_min, range_max],
    // passes through (the range is clamped to the system bounds); max_depth < 0 stands for None
    std::vector<ID> test_ray(const Bounds<Index::dim>& system_bounds, const float (&origin)[Index::dim],
                             const float (&direction)[Index::dim], float range_min, float range_max, int max_depth = -1) {
        float row[2 * Index::dim + 2];
        for (int a = 0; a < Index::dim; ++a) { row[a] = origin[a]; row[Index::dim + a] = direction[a]; }
        row[2 * Index::dim] = range_min;
        row[2 * Index::dim + 1] = range_max;
        const uint32_t* rows = nullptr;
        size_t n = 0;
        check(bp_layer_test_ray_batch(h_, system_bounds.min, system_bounds.max, row, 1, max_depth >= 0 ? 1 : 0,
                                      max_depth >= 0 ? (uint32_t)max_depth : 0u, &rows, &n));
        std::vector<ID> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = (ID)rows[2 * i + 1];
        return out;
    }

    // Layer::pick_ray (layer.rs:418-446) with the built-in ray/AABB distance over bounds_by_id (rows of 2*dim f32):
    // true + (dist, id) of the nearest box hit within max_dist
    bool pick_ray_aabb(const Bounds<Index::dim>& system_bounds, const float (&origin)[Index::dim],
                       const float (&direction)[Index::dim], float max_dist, const float* bounds_by_id, size_t n_rows,
                       float* dist, ID* id, int max_depth = -1) {
        float row[2 * Index::dim + 1];
        for (int a = 0; a < Index::dim; ++a) { row[a] = origin[a]; row[Index::dim + a] = direction[a]; }
        row[2 * Index::dim] = max_dist;
        float d = 0.f;
        uint32_t hit = 0xFFFFFFFFu;
        check(bp_layer_pick_ray_aabb_batch(h_, system_bounds.min, system_bounds.max, row, 1, max_depth >= 0 ? 1 : 0,
                                           max_depth >= 0 ? (uint32_t)max_depth : 0u, bounds_by_id, n_rows, &d, &hit));
        if (hit == 0xFFFFFFFFu) return false;
        *dist = d;
        *id = (ID)hit;
        return true;
    }

    // Layer::iter (layer.rs:79-81): entries in current order
    std::vector<std::pair<uint64_t, ID>> iter() {
        const uint64_t* k; const uint32_t* i; size_t m; int s;
        check(bp_layer_tree(h_, &k, &i, &m, &s));
        std::vector<std::pair<uint64_t, ID>> out(m);
        for (size_t e = 0; e < m; ++e) out[e] = {k[e], (ID)i[e]};
        return out;
    }

    bool operator==(const Layer& o) const { int eq = 0; check(bp_layer_equals(h_, o.h_, &eq)); return eq != 0; }  // layer.rs:576-586
    bool operator!=(const Layer& o) const { return !(*this == o); }

    size_t len() const { size_t m = 0; check(bp_layer_len(h_, &m)); return m; }
    bool sorted() const { int s = 0; check(bp_layer_is_sorted(h_, &s)); return s != 0; }
    uint32_t min_depth() const { uint32_t d = 0; check(bp_layer_min_depth(h_, &d)); return d; }
    bp_layer_t* handle() const { return h_; }

   private:
    template <class F> PairView filtered(int (*fn)(bp_layer_t*, bp_filter_fn, void*, const uint32_t**, size_t*), F&& filter) {
        auto thunk = [](uint32_t a, uint32_t b, void* u) -> int { return (*static_cast<typename std::remove_reference<F>::type*>(u))((ID)a, (ID)b) ? 1 : 0; };
        PairView v;
        check(fn(h_, thunk, (void*)&filter, &v.data, &v.len));
        return v;
    }
    void check(int rc, const bp_layer_t* who = nullptr) const {
        if (rc != BP_OK) throw Error(rc, bp_layer_last_error(who ? who : h_));
    }
    bp_layer_t* h_ = nullptr;
};

// LayerBuilder (layer.rs:619-696)
class LayerBuilder {
   public:
    static LayerBuilder create() { return LayerBuilder(); }  // LayerBuilder::new()
    LayerBuilder& with_min_depth(uint32_t d) { min_depth_ = d; return *this; }
    LayerBuilder& with_index_capacity(size_t c) { index_capacity_ = c; return *this; }
    LayerBuilder& with_collision_capacity(size_t c) { collision_capacity_ = c; return *this; }
    LayerBuilder& with_test_capacity(size_t c) { test_capacity_ = c; return *this; }  // query path: out of scope, accepted
    LayerBuilder& with_device(int d) { device_ = d; return *this; }                   // not in the reference
    template <class Index, class ID = uint32_t> Layer<Index, ID> build() const {
        return Layer<Index, ID>(min_depth_, index_capacity_, collision_capacity_, device_);
    }

   private:
    uint32_t min_depth_ = 0;
    size_t index_capacity_ = 0, collision_capacity_ = 0, test_capacity_ = 0;
    int device_ = -1;
};

}  // namespace broadphase
