// Device side of util.mask_data_to_segmentation (micro_sam/util.py:1773-1848): paint the kept masks into a label image
// straight from their bit masks, then connected-component labelling (4-connectivity, components = regions of equal
// non-zero value, i.e. elf.parallel.label on the painted image).  HBM-bound integer work; no GEMM shape anywhere.
//
//   paint : the reference paints masks in area-descending order, later (smaller) masks overwrite earlier ones, so
//           label(y,x) = rank+1 of the LAST mask in paint order that covers the pixel.  One thread per (32-row word,
//           column): walks the masks from last to first, assigns the still-unassigned bits, stops when all 32 are done.
//   label : label-equivalence union-find over block-major KEYS (common.h bm_key): hook (smaller key becomes the root) +
//           pointer-jumping compression, iterated until no hook fires.  The root key of a component is its first pixel in the
//           reference's labelling order (elf.parallel.label over 512 x 512 blocks), so ascending root keys == its numbering.
//   overlap : contingency table between consecutive slices of a label volume (the nifty.ground_truth.overlap behind
//           elf.tracking compute_edges_from_overlap, called by merge_instance_segmentation_3d,
//           micro_sam/multi_dimensional_segmentation.py:357): a scatter-add into an open-addressing hash table.
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);

namespace {

__global__ __launch_bounds__(256) void paint_kernel(const uint32_t* __restrict__ bits, const int* __restrict__ order, int K,
                                                    const int* __restrict__ k_dev, int H, int W, int* __restrict__ label) {
    const int x = blockIdx.x * 256 + threadIdx.x, yw = blockIdx.y;
    if (x >= W) return;
    if (k_dev) K = *k_dev;                     // number of masks decided on the device (no host round trip)
    const int wpc = (H + 31) >> 5;
    const int nb = min(32, H - yw * 32);
    uint32_t un = nb < 32 ? ((1u << nb) - 1u) : 0xffffffffu;
    for (int r = K - 1; r >= 0 && un; --r) {
        const uint32_t w = bits[((long)order[r] * wpc + yw) * W + x];
        uint32_t hit = w & un;
        un &= ~w;
        while (hit) {
            const int b = __ffs(hit) - 1;
            hit &= hit - 1;
            label[(long)(yw * 32 + b) * W + x] = r + 1;
        }
    }
    while (un) {
        const int b = __ffs(un) - 1;
        un &= un - 1;
        label[(long)(yw * 32 + b) * W + x] = 0;
    }
}

// L[p] (pixel-indexed) holds the KEY (common.h bm_key: block-major order = the reference's component numbering) of the parent of
// pixel p's tree node, -1 for background; the node of key q lives at pixel bm_pix(q).  Keys order the unions (smaller key = root),
// so a converged root key is the component's first pixel in block-major order.
__global__ __launch_bounds__(256) void cc_init_kernel(const int* __restrict__ seg, int H, int W, int* __restrict__ L) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < H * W) L[i] = seg[i] != 0 ? bm_key(i, H, W) : -1;
}

__device__ __forceinline__ int cc_find(const int* L, int i, int H, int W) {       // root KEY of the tree that pixel i is in
    int r = L[i];
    while (true) { const int p = L[bm_pix(r, H, W)]; if (p == r) break; r = p; }
    return r;
}

// hook: for every foreground pixel join with the right and the lower neighbour of equal value (covers every 4-edge once)
__global__ __launch_bounds__(256) void cc_hook_kernel(const int* __restrict__ seg, int H, int W, int* L, int* __restrict__ changed) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int v = seg[i];
    if (v == 0) return;
    const int y = i / W, x = i - y * W;
    int any = 0;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int j = d == 0 ? (x + 1 < W ? i + 1 : -1) : (y + 1 < H ? i + W : -1);
        if (j < 0 || seg[j] != v) continue;
        int a = cc_find(L, i, H, W), b = cc_find(L, j, H, W);
        while (a != b) {                       // union by smaller key (lock-free)
            if (a < b) { const int t = a; a = b; b = t; }      // a > b
            const int old = atomicMin(&L[bm_pix(a, H, W)], b);
            if (old == a) { any = 1; break; }
            a = old;                            // someone hooked a elsewhere: continue from there
            any = 1;
        }
    }
    if (any) *changed = 1;
}

__global__ __launch_bounds__(256) void cc_compress_kernel(int H, int W, int* L) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < H * W && L[i] >= 0) L[i] = cc_find(L, i, H, W);
}

// Component sizes keyed by root KEY: sizes[root key] += 1 for every foreground pixel.  A plain scatter-add puts hundreds
// of thousands of atomics on the few addresses of the big components; here every workgroup first aggregates its 4096
// pixels in an LDS open-addressing table (runs of equal roots are pre-summed per thread), then flushes one global
// atomic per distinct root.
__global__ __launch_bounds__(256) void cc_sizes_kernel(const int* __restrict__ roots, int n, int* __restrict__ sizes,
                                                       int* __restrict__ bg_count) {
    constexpr int TAB = 512, PER = 16;
    __shared__ int keys[TAB], vals[TAB];
    __shared__ int bg;
    for (int i = threadIdx.x; i < TAB; i += 256) { keys[i] = -1; vals[i] = 0; }
    if (threadIdx.x == 0) bg = 0;
    __syncthreads();
    const int base = (blockIdx.x * 256 + threadIdx.x) * PER;
    int cur = -2, cnt = 0, nbg = 0;
    auto flush = [&](int key, int c) {
        if (key < 0 || c == 0) return;
        unsigned h = ((unsigned)key * 2654435761u) >> 23;          // 9 bits
        for (int probe = 0; probe < 16; ++probe, h = (h + 1) & (TAB - 1)) {
            const int prev = atomicCAS(&keys[h], -1, key);
            if (prev == -1 || prev == key) { atomicAdd(&vals[h], c); return; }
        }
        atomicAdd(&sizes[key], c);                                 // table crowded: straight to global memory
    };
    for (int k = 0; k < PER; ++k) {
        const int i = base + k;
        if (i >= n) break;
        const int r = roots[i];
        if (r < 0) { ++nbg; continue; }
        if (r == cur) ++cnt;
        else { flush(cur, cnt); cur = r; cnt = 1; }
    }
    flush(cur, cnt);
    if (nbg) atomicAdd(&bg, nbg);
    __syncthreads();
    for (int i = threadIdx.x; i < TAB; i += 256)
        if (keys[i] >= 0) atomicAdd(&sizes[keys[i]], vals[i]);
    if (threadIdx.x == 0 && bg) atomicAdd(bg_count, bg);
}

// ---- greedy box NMS (torchvision.ops.nms semantics) on score-sorted boxes: suppression bit matrix + serial sweep
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int K, float thr,
                                                      unsigned long long* __restrict__ mask) {
    const int rb = blockIdx.y, cb = blockIdx.x, t = threadIdx.x;
    const int nblk = (K + 63) >> 6;
    __shared__ float cbx[64][4];
    const int cj = cb * 64 + t;
    if (cj < K) { cbx[t][0] = boxes[cj * 4]; cbx[t][1] = boxes[cj * 4 + 1]; cbx[t][2] = boxes[cj * 4 + 2]; cbx[t][3] = boxes[cj * 4 + 3]; }
    __syncthreads();
    const int i = rb * 64 + t;
    if (i >= K) return;
    unsigned long long bitsv = 0ull;
    if (cb >= rb) {                                         // only j > i can be suppressed by i
        const float x1 = boxes[i * 4], y1 = boxes[i * 4 + 1], x2 = boxes[i * 4 + 2], y2 = boxes[i * 4 + 3];
        const float ai = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
        const int ncol = min(64, K - cb * 64);
        for (int c = (cb == rb ? t + 1 : 0); c < ncol; ++c) {
            const float w = fmaxf(0.f, __fsub_rn(fminf(x2, cbx[c][2]), fmaxf(x1, cbx[c][0])));
            const float h = fmaxf(0.f, __fsub_rn(fminf(y2, cbx[c][3]), fmaxf(y1, cbx[c][1])));
            const float inter = __fmul_rn(w, h);
            const float aj = __fmul_rn(__fsub_rn(cbx[c][2], cbx[c][0]), __fsub_rn(cbx[c][3], cbx[c][1]));
            const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(ai, aj), inter));
            if (iou > thr) bitsv |= 1ull << c;
        }
    }
    mask[(long)i * nblk + cb] = bitsv;
}

// one wave: lane l owns words l, l+64, ... of the running "removed" set (K <= 64*64*4 words handled by the stride loop)
__global__ __launch_bounds__(64) void nms_sweep_kernel(const unsigned long long* __restrict__ mask, int K,
                                                       const int* __restrict__ valid, int* __restrict__ keep) {
    const int nblk = (K + 63) >> 6, lane = threadIdx.x;
    extern __shared__ unsigned long long remv[];
    for (int w = lane; w < nblk; w += 64) {
        unsigned long long r = 0ull;
        if (valid)                                   // boxes filtered out beforehand start as "removed"
            for (int b = 0; b < 64 && w * 64 + b < K; ++b) r |= (unsigned long long)(valid[w * 64 + b] == 0) << b;
        remv[w] = r;
    }
    __syncthreads();
    for (int i = 0; i < K; ++i) {
        const bool removed = (remv[i >> 6] >> (i & 63)) & 1ull;          // uniform across the wave
        if (lane == 0) keep[i] = removed ? 0 : 1;
        if (!removed)
            for (int w = lane; w < nblk; w += 64) remv[w] |= mask[(long)i * nblk + w];
        __syncthreads();
    }
}

// K <= 4096 (nblk <= 64): block-wise sweep.  The 64 x 64 diagonal blocks of the suppression matrix sit in LDS, every wave
// resolves the 64 boxes of a block redundantly in registers (no barrier for the keep word), and the rows of the block are
// OR-ed into the running "removed" set by all 256 threads with unconditional loads issued before the resolve.
__global__ __launch_bounds__(256) void nms_sweep64_kernel(const unsigned long long* __restrict__ mask, int K,
                                                          const int* __restrict__ valid, int* __restrict__ keep) {
    typedef unsigned long long u64;
    const int nblk = (K + 63) >> 6, tid = threadIdx.x, lane = tid & 63, grp = tid >> 6;
    extern __shared__ u64 sm[];
    u64* remv = sm;                    // [64]
    u64* diag = sm + 64;               // [nblk * 64]
    u64* part = diag + nblk * 64;      // [4][64]
    if (tid < 64) {
        u64 r = 0ull;
        if (tid < nblk) {
            for (int b = 0; b < 64; ++b) {
                const int i = tid * 64 + b;
                if (i >= K || (valid && valid[i] == 0)) r |= 1ull << b;     // filtered / padding boxes start as "removed"
            }
        }
        remv[tid] = r;
    }
    for (int i = tid; i < nblk * 64; i += 256) diag[i] = i < K ? mask[(long)i * nblk + (i >> 6)] : 0ull;
    __syncthreads();
    for (int b = 0; b < nblk; ++b) {
        u64 rows[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int r = b * 64 + grp * 16 + j;
            rows[j] = (lane < nblk && lane > b && r < K) ? mask[(long)r * nblk + lane] : 0ull;
        }
        u64 cur = remv[b];
        const u64 d = diag[b * 64 + lane];
        const unsigned dlo = (unsigned)d, dhi = (unsigned)(d >> 32);
        u64 keepw = 0ull;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (!((cur >> i) & 1ull)) {
                keepw |= 1ull << i;
                cur |= ((u64)(unsigned)__builtin_amdgcn_readlane((int)dhi, i) << 32) | (unsigned)__builtin_amdgcn_readlane((int)dlo, i);
            }
        }
        if (grp == 0 && b * 64 + lane < K) keep[b * 64 + lane] = (int)((keepw >> lane) & 1ull);
        u64 acc = 0ull;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if ((keepw >> (grp * 16 + j)) & 1ull) acc |= rows[j];
        part[grp * 64 + lane] = acc;
        __syncthreads();
        if (grp == 0) remv[lane] |= part[lane] | part[64 + lane] | part[128 + lane] | part[192 + lane];
        __syncthreads();
    }
}

// ---- mask NMS (micro_sam/util.py:1589-1668 _batched_mask_nms: IoU or intersection-over-min matrix of full-size masks +
// greedy suppression).  The reference multiplies the flattened masks (masks_flat @ masks_flat.T);
// on bit masks the same intersection is a popcount of AND over the overlap window of the two boxes - integer exact, no
// GEMM.  One workgroup per pair (i < j) of SCORE-SORTED masks writes bit j of row i of the suppression matrix that the
// box-NMS sweep kernels consume.
__global__ __launch_bounds__(256) void mask_nms_matrix_kernel(const uint32_t* __restrict__ bits, const int* __restrict__ order,
                                                              const float* __restrict__ boxes, const int* __restrict__ area,
                                                              int K, int H, int W, float thr, int iomin,
                                                              unsigned long long* __restrict__ mask) {
    const int i = blockIdx.y, j = blockIdx.x;
    if (j <= i || j >= K) return;
    const int a = order[i], b = order[j];
    // _overlap_matrix: the boxes share a region of positive area (float arithmetic on the xyxy boxes as given)
    const float x1 = fmaxf(boxes[a * 4], boxes[b * 4]), y1 = fmaxf(boxes[a * 4 + 1], boxes[b * 4 + 1]);
    const float x2 = fminf(boxes[a * 4 + 2], boxes[b * 4 + 2]), y2 = fminf(boxes[a * 4 + 3], boxes[b * 4 + 3]);
    if (!(fmaxf(x2 - x1, 0.f) * fmaxf(y2 - y1, 0.f) > 0.f)) return;
    // intersection over the window of the two boxes (everything outside it is 0 in at least one mask when the boxes are the
    // masks' bounding boxes; the window is widened by one pixel against inclusive / exclusive box conventions)
    const int wpc = (H + 31) >> 5;
    const int cx0 = max(0, (int)floorf(x1) - 1), cx1 = min(W, (int)ceilf(x2) + 2);
    const int wy0 = max(0, ((int)floorf(y1) - 1) >> 5), wy1 = min(wpc, (((int)ceilf(y2) + 1) >> 5) + 1);
    const int ncol = cx1 - cx0, nrow = wy1 - wy0;
    int cnt = 0;
    for (int t = threadIdx.x; t < ncol * nrow; t += 256) {
        const int wy = wy0 + t / ncol, x = cx0 + t % ncol;
        cnt += __popc(bits[((long)a * wpc + wy) * W + x] & bits[((long)b * wpc + wy) * W + x]);
    }
    __shared__ int red[4];
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float inter = (float)(red[0] + red[1] + red[2] + red[3]);
        const float aa = (float)area[a], ab = (float)area[b];
        const float v = iomin ? __fdiv_rn(inter, __fadd_rn(fminf(aa, ab), 1e-6f)) : __fdiv_rn(inter, __fsub_rn(__fadd_rn(aa, ab), inter));
        if (v > thr) atomicOr(&mask[(long)i * ((K + 63) >> 6) + (j >> 6)], 1ull << (j & 63));
    }
}


// ---- slice-to-slice overlap table.  Pair (a, b) = (labels[z][p], labels[z + 1][p]) with a != 0; key = a << 32 | b.
// Every thread walks PER consecutive pixels and pre-sums runs of equal pairs, a workgroup aggregates its pairs in an LDS table,
// then flushes one global update per distinct pair: objects are hundreds to thousands of pixels, so the global table sees a few
// atomics per (object pair, workgroup) instead of one per pixel.
typedef unsigned long long u64;
constexpr u64 OV_EMPTY = ~0ull;
__device__ __forceinline__ unsigned ov_hash(u64 k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; return (unsigned)k; }

__device__ __forceinline__ void ov_global_add(u64* __restrict__ keys, int* __restrict__ counts, unsigned mask, u64 key, int c,
                                              int* __restrict__ overflow) {
    unsigned h = ov_hash(key) & mask;
    for (unsigned probe = 0; probe <= mask; ++probe, h = (h + 1) & mask) {
        const u64 prev = atomicCAS(&keys[h], OV_EMPTY, key);
        if (prev == OV_EMPTY || prev == key) { atomicAdd(&counts[h], c); return; }
        if (probe >= 4096) break;
    }
    atomicExch(overflow, 1);
}

__global__ __launch_bounds__(256) void overlap_count_kernel(const int* __restrict__ labels, long slice, long pairs_px,
                                                            u64* __restrict__ keys, int* __restrict__ counts, unsigned mask,
                                                            int* __restrict__ overflow) {
    constexpr int TAB = 1024, PER = 16;
    __shared__ u64 lk[TAB];
    __shared__ int lv[TAB];
    for (int i = threadIdx.x; i < TAB; i += 256) { lk[i] = OV_EMPTY; lv[i] = 0; }
    __syncthreads();
    auto flush = [&](u64 key, int c) {
        if (c == 0) return;
        unsigned h = ov_hash(key) & (TAB - 1);
        for (int probe = 0; probe < 24; ++probe, h = (h + 1) & (TAB - 1)) {
            const u64 prev = atomicCAS(&lk[h], OV_EMPTY, key);
            if (prev == OV_EMPTY || prev == key) { atomicAdd(&lv[h], c); return; }
        }
        ov_global_add(keys, counts, mask, key, c, overflow);          // LDS table crowded
    };
    const long base = ((long)blockIdx.x * 256 + threadIdx.x) * PER;
    u64 cur = OV_EMPTY; int cnt = 0;
    for (int k = 0; k < PER; ++k) {
        const long i = base + k;                                     // pixel of the stack [Z - 1, H * W]: slice pairs are contiguous
        if (i >= pairs_px) break;
        const int a = labels[i], b = labels[i + slice];
        if (a == 0) { flush(cur, cnt); cur = OV_EMPTY; cnt = 0; continue; }
        const u64 key = ((u64)(uint32_t)a << 32) | (u64)(uint32_t)b;
        if (key == cur) ++cnt;
        else { flush(cur, cnt); cur = key; cnt = 1; }
    }
    if (cur != OV_EMPTY) flush(cur, cnt);
    __syncthreads();
    for (int i = threadIdx.x; i < TAB; i += 256)
        if (lk[i] != OV_EMPTY) ov_global_add(keys, counts, mask, lk[i], lv[i], overflow);
}

// table -> (source, target, count) triples in arbitrary order; n_out[0] counts them (the caller sorts)
__global__ __launch_bounds__(256) void overlap_compact_kernel(const u64* __restrict__ keys, const int* __restrict__ counts, unsigned cap,
                                                              int* __restrict__ out, int max_out, int* __restrict__ n_out) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap || keys[i] == OV_EMPTY) return;
    const int slot = atomicAdd(n_out, 1);
    if (slot >= max_out) return;
    out[3 * slot] = (int)(keys[i] >> 32); out[3 * slot + 1] = (int)(uint32_t)keys[i]; out[3 * slot + 2] = counts[i];
}

}  // namespace

extern "C" int msam_box_nms_valid(const float* boxes_sorted, const int32_t* valid_sorted, int32_t K, float iou_threshold,
                                  uint64_t* mask_scratch, int32_t* keep_flags, void* stream);

extern "C" int msam_box_nms(const float* boxes_sorted, int32_t K, float iou_threshold, uint64_t* mask_scratch,
                            int32_t* keep_flags, void* stream) {
    return msam_box_nms_valid(boxes_sorted, nullptr, K, iou_threshold, mask_scratch, keep_flags, stream);
}

extern "C" int msam_box_nms_valid(const float* boxes_sorted, const int32_t* valid_sorted, int32_t K, float iou_threshold,
                                  uint64_t* mask_scratch, int32_t* keep_flags, void* stream) {
    if (K < 0 || (K > 0 && (!boxes_sorted || !mask_scratch || !keep_flags))) { msam_set_error("msam_box_nms: bad arguments"); return 1; }
    if (K == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (K + 63) / 64;
    if (nblk * 8 > 60000) { msam_set_error("msam_box_nms: too many boxes"); return 1; }
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nblk, nblk), dim3(64), 0, s, boxes_sorted, K, iou_threshold,
                       (unsigned long long*)mask_scratch);
    if (nblk <= 64)
        hipLaunchKernelGGL(nms_sweep64_kernel, dim3(1), dim3(256), (64 + nblk * 64 + 256) * 8, s,
                           (const unsigned long long*)mask_scratch, K, valid_sorted, keep_flags);
    else
        hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), nblk * 8, s, (const unsigned long long*)mask_scratch, K,
                           valid_sorted, keep_flags);
    return msam_check_launch("msam_box_nms");
}

extern "C" int msam_paint_label_image(const uint32_t* bits, const int32_t* order, int32_t K, int32_t H, int32_t W,
                                      int32_t* label, void* stream) {
    if (!label || H <= 0 || W <= 0 || K < 0 || (K > 0 && (!bits || !order))) {
        msam_set_error("msam_paint_label_image: bad arguments");
        return 1;
    }
    hipLaunchKernelGGL(paint_kernel, dim3((W + 255) / 256, (H + 31) / 32), dim3(256), 0, (hipStream_t)stream, bits, order, K,
                       (const int*)nullptr, H, W, label);
    return msam_check_launch("msam_paint_label_image");
}

extern "C" int msam_paint_label_image_dev(const uint32_t* bits, const int32_t* order, const int32_t* k_dev, int32_t H,
                                          int32_t W, int32_t* label, void* stream) {
    if (!label || !bits || !order || !k_dev || H <= 0 || W <= 0) { msam_set_error("msam_paint_label_image_dev: bad arguments"); return 1; }
    hipLaunchKernelGGL(paint_kernel, dim3((W + 255) / 256, (H + 31) / 32), dim3(256), 0, (hipStream_t)stream, bits, order, 0,
                       k_dev, H, W, label);
    return msam_check_launch("msam_paint_label_image_dev");
}

extern "C" int msam_component_sizes(const int32_t* roots, int32_t n, int32_t* sizes, int32_t* bg_count, void* stream) {
    if (!roots || !sizes || !bg_count || n <= 0) { msam_set_error("msam_component_sizes: bad arguments"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(sizes, 0, (size_t)n * sizeof(int), s) != hipSuccess || hipMemsetAsync(bg_count, 0, sizeof(int), s) != hipSuccess) {
        msam_set_error("msam_component_sizes: memset failed");
        return 2;
    }
    hipLaunchKernelGGL(cc_sizes_kernel, dim3((n + 4095) / 4096), dim3(256), 0, s, roots, n, sizes, bg_count);
    return msam_check_launch("msam_component_sizes");
}

// Fixed number of union passes without host synchronisation; changed_flag holds the flag of the LAST pass (0 = converged).
extern "C" int msam_label_components_async(const int32_t* seg, int32_t H, int32_t W, int32_t* roots, int32_t* changed_flag,
                                           int32_t passes, void* stream) {
    if (!seg || !roots || !changed_flag || H <= 0 || W <= 0 || passes <= 0) { msam_set_error("msam_label_components_async: bad arguments"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    const int n = H * W, grid = (n + 255) / 256;
    hipLaunchKernelGGL(cc_init_kernel, dim3(grid), dim3(256), 0, s, seg, H, W, roots);
    for (int it = 0; it < passes; ++it) {
        if (hipMemsetAsync(changed_flag, 0, sizeof(int), s) != hipSuccess) { msam_set_error("msam_label_components_async: memset"); return 2; }
        hipLaunchKernelGGL(cc_hook_kernel, dim3(grid), dim3(256), 0, s, seg, H, W, roots, changed_flag);
        hipLaunchKernelGGL(cc_compress_kernel, dim3(grid), dim3(256), 0, s, H, W, roots);
    }
    return msam_check_launch("msam_label_components_async");
}

extern "C" int msam_label_components(const int32_t* seg, int32_t H, int32_t W, int32_t* roots, int32_t* changed_flag,
                                     int32_t max_iters, int32_t* iters_done, void* stream) {
    if (!seg || !roots || !changed_flag || H <= 0 || W <= 0) { msam_set_error("msam_label_components: bad arguments"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    const int n = H * W, grid = (n + 255) / 256;
    hipLaunchKernelGGL(cc_init_kernel, dim3(grid), dim3(256), 0, s, seg, H, W, roots);
    int it = 0;
    // the union loop inside cc_hook_kernel already merges whole trees, so one hook pass + compression is complete;
    // further passes only confirm convergence (changed == 0) - bounded by max_iters
    for (; it < (max_iters > 0 ? max_iters : 8); ++it) {
        if (hipMemsetAsync(changed_flag, 0, sizeof(int), s) != hipSuccess) { msam_set_error("msam_label_components: memset"); return 2; }
        hipLaunchKernelGGL(cc_hook_kernel, dim3(grid), dim3(256), 0, s, seg, H, W, roots, changed_flag);
        hipLaunchKernelGGL(cc_compress_kernel, dim3(grid), dim3(256), 0, s, H, W, roots);
        int h = 0;
        if (hipMemcpyAsync(&h, changed_flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { msam_set_error("msam_label_components: readback"); return 2; }
        if (!h) { ++it; break; }
    }
    if (iters_done) *iters_done = it;
    return msam_check_launch("msam_label_components");
}

// Greedy mask NMS: bits uint32 [K, ceil(H/32), W], order int32 [K] (descending score), boxes fp32 [K,4] xyxy and area int32 [K]
// indexed by MASK (not by sorted position); keep_flags int32 [K] in sorted order.  mask_scratch: K * ceil(K/64) uint64.
extern "C" int msam_mask_nms(const uint32_t* bits, const int32_t* order, const float* boxes, const int32_t* area, int32_t K,
                             int32_t H, int32_t W, float thresh, int32_t intersection_over_min, uint64_t* mask_scratch,
                             int32_t* keep_flags, void* stream) {
    if (K < 0 || (K > 0 && (!bits || !order || !boxes || !area || !mask_scratch || !keep_flags)) || H <= 0 || W <= 0) {
        msam_set_error("msam_mask_nms: bad arguments");
        return 1;
    }
    if (K == 0) return 0;
    if (K > 32768) { msam_set_error("msam_mask_nms: at most 32768 masks"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (K + 63) / 64;
    if (hipMemsetAsync(mask_scratch, 0, (size_t)K * nblk * 8, s) != hipSuccess) { msam_set_error("msam_mask_nms: memset failed"); return 2; }
    hipLaunchKernelGGL(mask_nms_matrix_kernel, dim3(K, K), dim3(256), 0, s, bits, order, boxes, area, K, H, W, thresh,
                       intersection_over_min, (unsigned long long*)mask_scratch);
    if (nblk <= 64)
        hipLaunchKernelGGL(nms_sweep64_kernel, dim3(1), dim3(256), (64 + nblk * 64 + 256) * 8, s,
                           (const unsigned long long*)mask_scratch, K, (const int*)nullptr, keep_flags);
    else
        hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), nblk * 8, s, (const unsigned long long*)mask_scratch, K,
                           (const int*)nullptr, keep_flags);
    return msam_check_launch("msam_mask_nms");
}

// Overlap table between consecutive slices (reference: elf.tracking.tracking_utilities.compute_edges_from_overlap ->
// nifty.ground_truth.overlap per slice pair; micro_sam/multi_dimensional_segmentation.py:357).  labels: int32 [Z, H, W], ids consecutive
// across z as _segment_slices leaves them.  table_keys: uint64 [capacity], table_counts: int32 [capacity] (capacity a power of two, at
// least 4 x the number of distinct (object, object-or-background) pairs; both initialised here).  edges: int32 [max_edges, 3] =
// (source id in slice z, target id in slice z + 1 (0 = background), overlapping pixels), unordered; n_edges: int32 [2] = {number of
// edges found (may exceed max_edges: only max_edges are written), overflow flag (table too small)}.
extern "C" int msam_slice_overlaps(const int32_t* labels, int32_t Z, int32_t H, int32_t W, uint64_t* table_keys, int32_t* table_counts,
                                   int32_t capacity, int32_t* edges, int32_t max_edges, int32_t* n_edges, void* stream) {
    if (!labels || !table_keys || !table_counts || !edges || !n_edges || Z < 1 || H <= 0 || W <= 0 || capacity < 1024 ||
        (capacity & (capacity - 1)) || max_edges <= 0) {
        msam_set_error("msam_slice_overlaps: bad arguments (capacity: a power of two >= 1024)");
        return 1;
    }
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(table_keys, 0xff, (size_t)capacity * 8, s) != hipSuccess || hipMemsetAsync(table_counts, 0, (size_t)capacity * 4, s) != hipSuccess ||
        hipMemsetAsync(n_edges, 0, 8, s) != hipSuccess) { msam_set_error("msam_slice_overlaps: memset failed"); return 2; }
    if (Z >= 2) {
        const long slice = (long)H * W, pairs_px = slice * (Z - 1);
        const long per_wg = 256L * 16;
        hipLaunchKernelGGL(overlap_count_kernel, dim3((unsigned)((pairs_px + per_wg - 1) / per_wg)), dim3(256), 0, s, labels, slice, pairs_px,
                           (u64*)table_keys, table_counts, (unsigned)(capacity - 1), n_edges + 1);
        hipLaunchKernelGGL(overlap_compact_kernel, dim3((capacity + 255) / 256), dim3(256), 0, s, (const u64*)table_keys, table_counts,
                           (unsigned)capacity, edges, max_edges, n_edges);
    }
    return msam_check_launch("msam_slice_overlaps");
}
