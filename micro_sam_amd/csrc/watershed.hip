// Seeded watershed of the decoder-based instance segmentation (reference micro_sam/instance_segmentation.py:1083-1168:
// torch_em.util.segmentation.watershed_from_center_and_boundary_distances -> skimage.segmentation.watershed(boundary_distances,
// markers=label(seeds), mask=foreground)).  HOST code: the reference runs this step on the CPU as well (a priority flood is a serial
// algorithm; the maps it consumes are one 4 MB download).  scikit-image is absent from the container, so the published algorithm of
// skimage/segmentation/_watershed_cy.pyx (watershed_raveled, compactness 0, no watershed line) is restated:
//   * all marker pixels enter a min-heap ordered by (height, age) in raster order, age = running push counter (FIFO among equal heights);
//   * pop the lowest element; every unlabelled 4-neighbour inside the mask, visited in the order up / left / right / down, takes the
//     popped pixel's label AT PUSH TIME and is pushed with its own height and the next age.
// PARITY UNPINNED against the library itself (not importable here); the test checks the flood's defining properties.
#include <cstdint>
#include <queue>
#include <vector>

void msam_set_error(const char* msg);

namespace {
struct Elem { float value; int64_t age; int32_t index; };
struct Cmp { bool operator()(const Elem& a, const Elem& b) const { return a.value > b.value || (a.value == b.value && a.age > b.age); } };
}  // namespace

// image fp32 [H, W] (heights), markers int32 [H, W] (0 = unlabelled), mask uint8 [H, W] (0 = outside; NULL = everywhere), out int32 [H, W]
// (host pointers).  Marker pixels outside the mask are dropped, as scikit-image does.
extern "C" int msam_host_seeded_watershed(const float* image, const int32_t* markers, const uint8_t* mask, int32_t H, int32_t W, int32_t* out) {
    if (!image || !markers || !out || H <= 0 || W <= 0) { msam_set_error("msam_host_seeded_watershed: bad argument"); return 1; }
    const int64_t n = (int64_t)H * W;
    std::priority_queue<Elem, std::vector<Elem>, Cmp> heap;
    int64_t age = 0;
    for (int64_t i = 0; i < n; ++i) {
        const bool inside = !mask || mask[i];
        out[i] = inside ? markers[i] : 0;
        if (out[i] != 0) heap.push(Elem{image[i], age++, (int32_t)i});       // age 0.. in raster order (skimage: all seeds at age 0, heap
    }                                                                        // insertion order = raster order decides among equals)
    const int dy[4] = {-1, 0, 0, 1}, dx[4] = {0, -1, 1, 0};
    while (!heap.empty()) {
        const Elem e = heap.top();
        heap.pop();
        const int y = e.index / W, x = e.index - y * W;
        for (int k = 0; k < 4; ++k) {
            const int yy = y + dy[k], xx = x + dx[k];
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const int64_t j = (int64_t)yy * W + xx;
            if ((mask && !mask[j]) || out[j] != 0) continue;
            out[j] = out[e.index];
            heap.push(Elem{image[j], age++, (int32_t)j});
        }
    }
    return 0;
}
