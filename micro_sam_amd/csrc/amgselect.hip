// AutomaticMaskGenerator.generate(output_mode="instance_segmentation") for the single-crop device state, as ONE call that
// enqueues 15 kernels and no torch operator (reference: AMGBase._postprocess_batch micro_sam/instance_segmentation.py:99-144 -
// predicted-IoU / stability / crop-edge filters + box NMS - and util.mask_data_to_segmentation util.py:1773-1848 - paint by
// descending area, connected components, drop the largest component (with_background) and the small ones, relabel
// consecutively).  Integer / compare work on <= 4096 candidates and one label image: latency-bound, no GEMM shape.
//
//   amg_select_sort_kernel   filters -> valid flags; stable descending sort of the scores (bitonic sort of 64-bit
//                            (inverted ordered score, index) keys in LDS, one workgroup) -> order, sorted boxes / flags
//   nms_mask / nms_sweep64   (segment.hip) greedy NMS on the sorted boxes
//   amg_area_sort_kernel     kept flags back to candidate order, stable descending sort by area -> paint order + count
//   paint / cc_*             (segment.hip) label image from the bit masks, union-find components, component sizes
//   relabel_*                per-block root statistics -> one-workgroup scan (+ largest-component decision) -> new ids at the
//                            roots -> gather
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);
extern "C" int msam_box_nms_valid(const float* boxes_sorted, const int32_t* valid_sorted, int32_t K, float iou_threshold,
                                  uint64_t* mask_scratch, int32_t* keep_flags, void* stream);
extern "C" int msam_paint_label_image_dev(const uint32_t* bits, const int32_t* order, const int32_t* k_dev, int32_t H, int32_t W,
                                          int32_t* label, void* stream);
extern "C" int msam_label_components_async(const int32_t* seg, int32_t H, int32_t W, int32_t* roots, int32_t* changed_flag,
                                           int32_t passes, void* stream);
extern "C" int msam_component_sizes(const int32_t* roots, int32_t n, int32_t* sizes, int32_t* bg_count, void* stream);

namespace {

typedef unsigned long long u64;
constexpr int NMAX = 4096, NT = 1024;

MSAM_DEVINL uint32_t f2ord(float f) {            // order-preserving float -> uint32
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ascending bitonic sort of 4096 64-bit keys in LDS by 1024 threads
MSAM_DEVINL void bitonic4096(u64* keys, int tid) {
    for (int k = 2; k <= NMAX; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                const int t = tid + rep * NT;
                const int i = ((t / j) * 2 * j) + (t % j), p = i + j;
                const bool asc = (i & k) == 0;
                const u64 a = keys[i], b = keys[p];
                if ((a > b) == asc) { keys[i] = b; keys[p] = a; }
            }
            __syncthreads();
        }
    }
}

struct SelArgs {
    const float* iou; const float* stab; const int* boxes; int N;
    float iou_thr, stab_thr; int x0, y0, x1, y1, W, H;
    int* order; float* boxes_sorted; int* valid_sorted;
};

__global__ __launch_bounds__(NT) void amg_select_sort_kernel(SelArgs a) {
    __shared__ u64 keys[NMAX];
    const int tid = threadIdx.x;
    const float crop[4] = {(float)a.x0, (float)a.y0, (float)a.x1, (float)a.y1};
    const float orig[4] = {0.f, 0.f, (float)a.W, (float)a.H};
    for (int i = tid; i < NMAX; i += NT) {
        u64 key = ~0ull;
        if (i < a.N) {
            bool valid = true;
            if (a.iou_thr > 0.f) valid = valid && (a.iou[i] > a.iou_thr);
            if (a.stab_thr > 0.f) valid = valid && (a.stab[i] >= a.stab_thr);            // NaN (0 / 0: empty mask) -> false
            bool near = false;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float b = (float)(a.boxes[i * 4 + c] + ((c & 1) ? a.y0 : a.x0));
                near = near || (fabsf(b - crop[c]) <= 20.f && !(fabsf(b - orig[c]) <= 20.f));   // is_box_near_crop_edge, atol 20
            }
            valid = valid && !near;
            const float s = valid ? a.iou[i] : -__builtin_inff();
            key = ((u64)(~f2ord(s)) << 32) | (u64)(uint32_t)i | (valid ? 0ull : 0ull);
            // the valid flag travels in bit 31 of the low word (N <= 4096 needs 12 bits)
            if (valid) key |= 1ull << 31;
        }
        keys[i] = key;
    }
    __syncthreads();
    bitonic4096(keys, tid);
    for (int i = tid; i < a.N; i += NT) {
        const uint32_t lo = (uint32_t)keys[i];
        const int idx = (int)(lo & 0x7fffffffu);
        a.order[i] = idx;
        a.valid_sorted[i] = (int)(lo >> 31);
#pragma unroll
        for (int c = 0; c < 4; ++c) a.boxes_sorted[i * 4 + c] = (float)a.boxes[idx * 4 + c];
    }
}

// keep flags (sorted order) -> candidate order; stable area-descending order of the selected masks + their count
__global__ __launch_bounds__(NT) void amg_area_sort_kernel(const int* __restrict__ order, const int* __restrict__ keep_sorted,
                                                           const int* __restrict__ area, int N, int min_size,
                                                           int* __restrict__ order2, int* __restrict__ k_dev) {
    __shared__ u64 keys[NMAX];
    __shared__ int count;
    const int tid = threadIdx.x;
    if (tid == 0) count = 0;
    for (int i = tid; i < NMAX; i += NT) keys[i] = ~0ull;
    __syncthreads();
    for (int i = tid; i < N; i += NT) {
        const int idx = order[i];
        const int ar = area[idx];
        const bool sel = keep_sorted[i] != 0 && (min_size <= 0 || ar >= min_size);
        // descending area, ties by ascending candidate index; unselected masks sink to the end
        keys[idx] = sel ? (((u64)(uint32_t)(0x7fffffff - ar) << 32) | (u64)(uint32_t)idx) : ((0xfffffffeull << 32) | (u64)(uint32_t)idx);
        if (sel) atomicAdd(&count, 1);
    }
    __syncthreads();
    bitonic4096(keys, tid);
    for (int i = tid; i < N; i += NT) order2[i] = (int)(uint32_t)keys[i];
    if (tid == 0) *k_dev = count;
}

// ---- relabel: roots int32 [n] per PIXEL = key of its component's root (-1 = background; common.h bm_key: block-major order, the
// reference's numbering); the pixel bm_pix(q) of a root key q has roots[..] == q; sizes[q] at root keys.  The kernels below walk the
// KEY space in ascending order, so components are numbered by ascending root key.
constexpr int RB = 1024;           // pixels per block (256 threads x 4)

__global__ __launch_bounds__(256) void relabel_stats_kernel(const int* __restrict__ roots, const int* __restrict__ sizes, int n, int H, int W,
                                                            int min_size, int* __restrict__ blk_cnt, u64* __restrict__ blk_max) {
    __shared__ int scnt[4];
    __shared__ u64 smax[4];
    const int base = blockIdx.x * RB + threadIdx.x * 4;
    int cnt = 0; u64 mx = 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = base + k;                                                              // a key
        if (p < n && roots[bm_pix(p, H, W)] == p) {
            const int sz = sizes[p];
            if (sz >= min_size) ++cnt;
            const u64 key = ((u64)(uint32_t)sz << 32) | (u64)(0xffffffffu - (uint32_t)p);     // larger size, then smaller index
            mx = key > mx ? key : mx;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o);
        const u64 other = ((u64)(uint32_t)__shfl_xor((int)(mx >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)mx, o);
        mx = other > mx ? other : mx;
    }
    if ((threadIdx.x & 63) == 0) { scnt[threadIdx.x >> 6] = cnt; smax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        blk_cnt[blockIdx.x] = scnt[0] + scnt[1] + scnt[2] + scnt[3];
        u64 m = smax[0];
        for (int w = 1; w < 4; ++w) m = smax[w] > m ? smax[w] : m;
        blk_max[blockIdx.x] = m;
    }
}

// one workgroup: largest component / background decision, exclusive scan of the block counts
__global__ __launch_bounds__(NT) void relabel_scan_kernel(int* __restrict__ blk_cnt, const u64* __restrict__ blk_max, int nb,
                                                          const int* __restrict__ bg_count, int with_background, int min_size,
                                                          int* __restrict__ blk_off, int* __restrict__ drop_idx) {
    __shared__ u64 red[NT];
    __shared__ int part[NT];
    __shared__ int carry;
    const int tid = threadIdx.x;
    u64 m = 0ull;
    for (int b = tid; b < nb; b += NT) m = blk_max[b] > m ? blk_max[b] : m;
    red[tid] = m;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (tid < s && red[tid + s] > red[tid]) red[tid] = red[tid + s];
        __syncthreads();
    }
    int drop = -1;
    {
        const u64 best = red[0];
        const int best_size = (int)(best >> 32), best_idx = (int)(0xffffffffu - (uint32_t)best);
        // np.unique reports label 0 (the background) as well: the largest id is dropped, label 0 wins ties (smallest id)
        if (with_background && best != 0ull && best_size > *bg_count) drop = best_idx;
        if (tid == 0) {
            *drop_idx = drop;
            if (drop >= 0 && best_size >= min_size) blk_cnt[drop / RB] -= 1;
            carry = 0;
        }
    }
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += NT) {
        const int b = b0 + tid;
        const int v = b < nb ? blk_cnt[b] : 0;
        part[tid] = v;
        __syncthreads();
        for (int s = 1; s < NT; s <<= 1) {
            const int add = tid >= s ? part[tid - s] : 0;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        if (b < nb) blk_off[b] = carry + part[tid] - v;
        __syncthreads();
        if (tid == NT - 1) carry += part[tid];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void relabel_assign_kernel(const int* __restrict__ roots, const int* __restrict__ sizes, int n, int H, int W,
                                                             int min_size, const int* __restrict__ blk_off,
                                                             const int* __restrict__ drop_idx, int* __restrict__ newid) {
    __shared__ int wsum[4];
    const int base = blockIdx.x * RB + threadIdx.x * 4;
    const int drop = *drop_idx;
    int f[4], c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = base + k;
        f[k] = (p < n && roots[bm_pix(p, H, W)] == p && sizes[p] >= min_size && p != drop) ? 1 : 0;
        c += f[k];
    }
    int incl = c;                                   // inclusive scan over the 64 lanes of the wave
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if ((threadIdx.x & 63) >= o) incl += t; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
    int rank = blk_off[blockIdx.x] + woff + incl - c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = base + k;
        if (p < n) newid[p] = f[k] ? ++rank : 0;
    }
}

__global__ __launch_bounds__(256) void relabel_gather_kernel(const int* __restrict__ roots, const int* __restrict__ newid, int n,
                                                             int* __restrict__ labels) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int r = roots[p];
    labels[p] = r >= 0 ? newid[r] : 0;
}

inline int64_t al256(int64_t x) { return (x + 255) & ~255LL; }

}  // namespace

extern "C" int64_t msam_amg_generate_workspace_bytes(int32_t N, int32_t H, int32_t W) {
    if (N <= 0 || N > NMAX || H <= 0 || W <= 0) return 0;
    const int64_t n = (int64_t)H * W, nb = (n + RB - 1) / RB, nblk = (N + 63) / 64;
    return al256(4LL * N) * 4 + al256(16LL * N) + al256(8LL * N * nblk) + 256 + al256(4 * n) * 4 + al256(4 * nb) * 2 + al256(8 * nb) + 256;
}

extern "C" int msam_amg_generate_labels(const float* iou, const float* stability, const int32_t* boxes, const int32_t* area,
                                        const uint32_t* bits, int32_t N, int32_t H, int32_t W, const int32_t* crop_box,
                                        float pred_iou_thresh, float stability_score_thresh, float box_nms_thresh,
                                        int32_t min_object_size, int32_t with_background, int32_t* labels, int32_t* flag,
                                        void* workspace, int64_t workspace_bytes, void* stream) {
    if (!iou || !stability || !boxes || !area || !bits || !crop_box || !labels || !flag || !workspace || N <= 0 || N > NMAX ||
        H <= 0 || W <= 0) {
        msam_set_error("msam_amg_generate_labels: bad argument (1 <= N <= 4096 candidates)");
        return 1;
    }
    if (workspace_bytes < msam_amg_generate_workspace_bytes(N, H, W)) { msam_set_error("msam_amg_generate_labels: workspace too small"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    const int n = H * W, nb = (n + RB - 1) / RB, nblk = (N + 63) / 64;
    char* p = (char*)workspace;
    auto take = [&](int64_t bytes) { char* r = p; p += al256(bytes); return r; };
    int* order = (int*)take(4LL * N); int* valid_sorted = (int*)take(4LL * N); int* keep_sorted = (int*)take(4LL * N);
    int* order2 = (int*)take(4LL * N);
    float* boxes_sorted = (float*)take(16LL * N);
    uint64_t* scratch = (uint64_t*)take(8LL * N * nblk);
    int* small = (int*)take(256);                    // [0] k_dev, [1] bg count, [2] drop index
    int* painted = (int*)take(4LL * n); int* roots = (int*)take(4LL * n); int* sizes = (int*)take(4LL * n); int* newid = (int*)take(4LL * n);
    int* blk_cnt = (int*)take(4LL * nb); int* blk_off = (int*)take(4LL * nb);
    u64* blk_max = (u64*)take(8LL * nb);
    int e;
    SelArgs a{iou, stability, boxes, N, pred_iou_thresh, stability_score_thresh, crop_box[0], crop_box[1], crop_box[2], crop_box[3],
              W, H, order, boxes_sorted, valid_sorted};
    hipLaunchKernelGGL(amg_select_sort_kernel, dim3(1), dim3(NT), 0, s, a);
    if ((e = msam_check_launch("amg_select_sort"))) return e;
    if ((e = msam_box_nms_valid(boxes_sorted, valid_sorted, N, box_nms_thresh, scratch, keep_sorted, s))) return e;
    hipLaunchKernelGGL(amg_area_sort_kernel, dim3(1), dim3(NT), 0, s, order, keep_sorted, area, N, min_object_size, order2, small);
    if ((e = msam_check_launch("amg_area_sort"))) return e;
    if ((e = msam_paint_label_image_dev(bits, order2, small, H, W, painted, s))) return e;
    if ((e = msam_label_components_async(painted, H, W, roots, flag, 2, s))) return e;
    if ((e = msam_component_sizes(roots, n, sizes, small + 1, s))) return e;
    hipLaunchKernelGGL(relabel_stats_kernel, dim3(nb), dim3(256), 0, s, roots, sizes, n, H, W, min_object_size, blk_cnt, blk_max);
    hipLaunchKernelGGL(relabel_scan_kernel, dim3(1), dim3(NT), 0, s, blk_cnt, blk_max, nb, small + 1, with_background,
                       min_object_size, blk_off, small + 2);
    hipLaunchKernelGGL(relabel_assign_kernel, dim3(nb), dim3(256), 0, s, roots, sizes, n, H, W, min_object_size, blk_off, small + 2, newid);
    hipLaunchKernelGGL(relabel_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, roots, newid, n, labels);
    return msam_check_launch("msam_amg_generate_labels");
}

// util.mask_data_to_segmentation(label_masks=True, merge_exclusively=False) for K masks that are already selected (reference
// micro_sam/util.py:1773-1848): paint in the given order (later masks overwrite), connected components, drop components below
// min_object_size and - with_background - the largest one counting label 0, consecutive relabel: the tail of
// msam_amg_generate_labels as its own entry point (7 kernels, no host synchronisation).  bits uint32 [*, ceil(H/32), W]; order int32 [K]
// = mask indices in paint order (the caller sorts by area, stable, descending); K may also come from device memory (k_dev != NULL).
extern "C" int64_t msam_labels_from_masks_workspace_bytes(int32_t H, int32_t W) {
    if (H <= 0 || W <= 0) return 0;
    const int64_t n = (int64_t)H * W, nb = (n + RB - 1) / RB;
    return 256 + al256(4 * n) * 4 + al256(4 * nb) * 2 + al256(8 * nb) + 256;
}

extern "C" int msam_labels_from_masks(const uint32_t* bits, const int32_t* order, int32_t K, const int32_t* k_dev, int32_t H, int32_t W,
                                      int32_t min_object_size, int32_t with_background, int32_t* labels, int32_t* flag,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
    if (!bits || !order || !labels || !flag || !workspace || H <= 0 || W <= 0 || (K < 0 && !k_dev)) {
        msam_set_error("msam_labels_from_masks: bad argument");
        return 1;
    }
    if (workspace_bytes < msam_labels_from_masks_workspace_bytes(H, W)) { msam_set_error("msam_labels_from_masks: workspace too small"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    const int n = H * W, nb = (n + RB - 1) / RB;
    char* p = (char*)workspace;
    auto take = [&](int64_t bytes) { char* r = p; p += al256(bytes); return r; };
    int* small = (int*)take(256);                    // [1] bg count, [2] drop index
    int* painted = (int*)take(4LL * n); int* roots = (int*)take(4LL * n); int* sizes = (int*)take(4LL * n); int* newid = (int*)take(4LL * n);
    int* blk_cnt = (int*)take(4LL * nb); int* blk_off = (int*)take(4LL * nb);
    u64* blk_max = (u64*)take(8LL * nb);
    int e;
    if (k_dev) { if ((e = msam_paint_label_image_dev(bits, order, k_dev, H, W, painted, s))) return e; }
    else if ((e = msam_paint_label_image(bits, order, K, H, W, painted, s))) return e;
    if ((e = msam_label_components_async(painted, H, W, roots, flag, 2, s))) return e;
    if ((e = msam_component_sizes(roots, n, sizes, small + 1, s))) return e;
    hipLaunchKernelGGL(relabel_stats_kernel, dim3(nb), dim3(256), 0, s, roots, sizes, n, H, W, min_object_size, blk_cnt, blk_max);
    hipLaunchKernelGGL(relabel_scan_kernel, dim3(1), dim3(NT), 0, s, blk_cnt, blk_max, nb, small + 1, with_background,
                       min_object_size, blk_off, small + 2);
    hipLaunchKernelGGL(relabel_assign_kernel, dim3(nb), dim3(256), 0, s, roots, sizes, n, H, W, min_object_size, blk_off, small + 2, newid);
    hipLaunchKernelGGL(relabel_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, roots, newid, n, labels);
    return msam_check_launch("msam_labels_from_masks");
}
