// C++ facade test (GPU): mirrors the reference's usage snippet (src/lib.rs:24-47) and the
// assertions of tests/test_layer.rs on the SURVEY.md 8(c) worked example.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "broadphase/layer.hpp"

using namespace broadphase;

#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #cond, __LINE__); return 1; } \
    } while (0)

int main() {
    using B3 = Bounds<3>;
    const B3 system_bounds{{0, 0, 0}, {16, 16, 16}};
    std::vector<std::pair<B3, uint32_t>> objects = {
        {B3{{1, 1, 1}, {2, 2, 2}}, 0},      {B3{{1.5f, 1.5f, 1.5f}, {3, 3, 3}}, 1}, {B3{{0, 0, 0}, {16, 16, 16}}, 2},
        {B3{{7, 7, 7}, {9, 9, 9}}, 3},      {B3{{20, 0, 0}, {21, 1, 1}}, 4},  // outside: skipped silently
    };
    auto layer = LayerBuilder::create().with_min_depth(0).with_index_capacity(64).build<Index64_3D, uint32_t>();
    layer.clear();
    layer.extend(system_bounds, objects);
    CHECK(layer.len() == 25 && !layer.sorted());
    auto tree = layer.iter();
    CHECK(tree[0].first == 0x4ull && tree[0].second == 0);          // id 0, depth 4, first cell
    CHECK(tree[16].first == 0x0ull && tree[16].second == 2);        // whole-system box: Index::default()
    CHECK(tree[24].first == 0x3800000000000003ull && tree[24].second == 3);

    auto snapshot = layer;  // Clone
    CHECK(snapshot == layer);
    layer.par_sort();
    CHECK(layer.sorted() && snapshot != layer);
    tree = layer.iter();
    for (size_t i = 1; i < tree.size(); ++i) CHECK(tree[i - 1] <= tree[i]);  // tests/test_layer.rs:42-54

    const uint32_t expect[4][2] = {{0, 1}, {0, 2}, {1, 2}, {3, 2}};
    auto pairs = layer.par_scan();
    CHECK(pairs.size() == 4);
    for (size_t i = 0; i < 4; ++i) CHECK(pairs[i].first == expect[i][0] && pairs[i].second == expect[i][1]);
    auto serial = layer.scan();
    CHECK(serial.size() == 4);

    size_t calls = 0;
    auto filtered = layer.scan_filtered([&](uint32_t a, uint32_t b) { ++calls; return b == 2 && a != 3; });
    CHECK(calls == 32);  // the filter sees every raw pair before dedup (layer.rs:455-459)
    CHECK(filtered.size() == 2 && filtered[0].first == 0 && filtered[1].first == 1);

    // test_box (layer.rs:294-313): expected ids from the oracle's restatement of test_impl
    auto near_origin = layer.test_box(system_bounds, B3{{1.f, 1.f, 1.f}, {2.f, 2.f, 2.f}});
    CHECK(near_origin.size() == 3 && near_origin[0] == 0 && near_origin[1] == 1 && near_origin[2] == 2);
    auto far_corner = layer.test_box(system_bounds, B3{{10.f, 10.f, 10.f}, {12.f, 12.f, 12.f}}, 2);
    CHECK(far_corner.size() == 2 && far_corner[0] == 2 && far_corner[1] == 3);
    const float ray_o[3] = {0.5f, 8.0f, 8.0f}, ray_d[3] = {1.0f, 0.0f, 0.0f};
    auto along = layer.test_ray(system_bounds, ray_o, ray_d, -INFINITY, INFINITY);
    CHECK(along.size() == 2 && along[0] == 2 && along[1] == 3);   // the whole-system box and the one around (8, 8, 8)
    const float boxes_by_id[4][6] = {{1, 1, 1, 2, 2, 2}, {1.5f, 1.5f, 1.5f, 3, 3, 3}, {0, 0, 0, 16, 16, 16}, {7, 7, 7, 9, 9, 9}};
    float hit_dist = -1.f;
    uint32_t hit_id = 99;
    CHECK(layer.pick_ray_aabb(system_bounds, ray_o, ray_d, 100.f, &boxes_by_id[0][0], 4, &hit_dist, &hit_id));
    CHECK(hit_id == 2 && hit_dist == 0.f);                          // the origin is inside the whole-system box
    CHECK(layer.par_scan().size() == 4);  // the scan is unaffected by the shared result buffers

    // merge: static layer into a dynamic one (layer.rs:123-138)
    auto dynamic = LayerBuilder::create().with_min_depth(2).build<Index64_3D>();
    dynamic.extend(system_bounds, {{B3{{2.5f, 2.5f, 2.5f}, {2.75f, 2.75f, 2.75f}}, 9}});
    dynamic.merge(layer);
    CHECK(dynamic.min_depth() == 0 && !dynamic.sorted() && dynamic.len() > 25 && dynamic.iter().size() == dynamic.len());
    auto merged = dynamic.par_scan();
    bool has_9_1 = false, has_9_2 = false;
    for (size_t i = 0; i < merged.size(); ++i) {
        has_9_1 |= merged[i].first == 9 && merged[i].second == 1;
        has_9_2 |= merged[i].first == 9 && merged[i].second == 2;
    }
    CHECK(has_9_1 && has_9_2);

    // 2-D index type through the same facade
    auto flat = LayerBuilder::create().with_min_depth(4).build<Index32_2D>();
    flat.extend(Bounds<2>{{-1, -1}, {1280, 1280}}, {{Bounds<2>{{10, 10}, {20, 20}}, 1}, {Bounds<2>{{15, 15}, {30, 30}}, 2}, {Bounds<2>{{600, 600}, {610, 610}}, 3}});
    auto p2 = flat.par_scan();
    CHECK(p2.size() == 1 && ((p2[0].first == 2 && p2[0].second == 1) || (p2[0].first == 1 && p2[0].second == 2)));

    // ---- the example's frame loop, call for call (examples/main.rs:364-374 layer setup, :410-418 clear + extend over the
    // circles' AABBs, :420 par_sort, :459 par_scan, :461-479 narrow phase over the candidate pairs).  The Rust shim
    // (rust/src/lib.rs) has this exact shape; it cannot be compiled here, this twin can.
    {
        struct Ball { float x, y, r; };
        std::vector<Ball> balls;
        uint32_t rng = 12345u;
        auto frand = [&]() { rng = rng * 1664525u + 1013904223u; return (float)(rng >> 8) / 16777216.0f; };
        const float W = 1280.f, H = 720.f;
        for (int i = 0; i < 2500; ++i) {                                   // BALL_COUNT_MAX (examples/main.rs:304)
            const float r = std::exp(0.5f + 1.5f * frand());               // :314
            balls.push_back(Ball{r + (W - 2 * r) * frand(), r + (H - 2 * r) * frand(), r});
        }
        const Bounds<2> bounds2{{-1.f, -1.f}, {W, W}};                       // system bounds :163-169
        auto frame_layer = LayerBuilder::create().with_min_depth(4).build<Index32_2D, uint32_t>();   // :364-374
        size_t hits_last = 0;
        for (int frame = 0; frame < 3; ++frame) {
            for (auto& b : balls) b.y += 0.25f * (float)frame;            // the scene moves between frames
            std::vector<std::pair<Bounds<2>, uint32_t>> objs;
            for (uint32_t i = 0; i < balls.size(); ++i)
                objs.push_back({Bounds<2>{{balls[i].x - balls[i].r, balls[i].y - balls[i].r}, {balls[i].x + balls[i].r, balls[i].y + balls[i].r}}, i});
            frame_layer.clear();                                           // :410
            frame_layer.extend(bounds2, objs);                             // :412-418
            frame_layer.par_sort();                                        // :420
            auto cand = frame_layer.par_scan();                            // :459
            size_t hits = 0;
            std::vector<std::pair<uint32_t, uint32_t>> narrow;
            for (size_t k = 0; k < cand.size(); ++k) {                     // narrow phase :461-479
                const Ball &a = balls[cand[k].first], &b = balls[cand[k].second];
                const float dx = b.x - a.x, dy = b.y - a.y;
                if (!(std::sqrt(dx * dx + dy * dy) > a.r + b.r)) { ++hits; narrow.push_back({std::min(cand[k].first, cand[k].second), std::max(cand[k].first, cand[k].second)}); }
            }
            // the broadphase may only add false positives: every touching pair of circles must be among the candidates
            size_t brute = 0;
            for (uint32_t i = 0; i < balls.size(); ++i)
                for (uint32_t j = i + 1; j < balls.size(); ++j) {
                    const float dx = balls[j].x - balls[i].x, dy = balls[j].y - balls[i].y;
                    if (!(std::sqrt(dx * dx + dy * dy) > balls[i].r + balls[j].r)) ++brute;
                }
            CHECK(hits == brute && cand.size() >= hits);
            for (size_t k = 1; k < cand.size(); ++k) CHECK(cand[k - 1] < cand[k]);   // sorted, duplicate-free (tests/test_layer.rs:42-54)
            hits_last = hits;
        }
        std::printf("example frame loop ok: %zu colliding pairs in the last frame\n", hits_last);
    }

    std::printf("facade ok: %zu pairs, %zu merged pairs\n", pairs.size(), merged.size());
    return 0;
}
