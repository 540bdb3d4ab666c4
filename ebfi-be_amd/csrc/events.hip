// Event-list -> per-bin polarity count images, bit-exact with the reference's events_to_stack
// (dataloader/encodings.py:307-350, :243-268, :77-99).  Integer-valued fp32 counts.
//
// The reference walks B time bins on the host; each bin is the slice [beg, end) found with its own
// binary search (exact hit at the probed end/mid, otherwise l / r; end = r + 1, so neighbouring
// bins can share events) and is accumulated with index_put_(accumulate=True).  Two side effects are
// part of the observable result and are reproduced here in closed form:
//   * out-of-range events are zeroed IN PLACE (x = y = 0, weight 0) by the positive-image pass, so
//     the negative-image pass of the same bin already sees them at pixel (0,0): a negative
//     out-of-range event counts there;
//   * the zeroing persists, so in every LATER bin that shares the event it counts at (0,0) too.
// Device plan: kernel 1 (one thread per bin) replays the float64 bin-edge arithmetic without FMA
// contraction and the exact search; kernel 2 (one thread per event) finds the bins that contain the
// event and adds p*p with fp32 atomics (sums of small integers: order-independent, exact).
#include "common.hpp"

using namespace ebfi;

namespace {

__device__ int64_t ref_bsearch(const double *__restrict__ t, int64_t l, int64_t r, double x, bool left) {
    while (l <= r) {
        if (t[l] == x) return l;
        if (t[r] == x) return r;
        const int64_t mid = l + (r - l) / 2;
        const double mv = t[mid];
        if (mv == x) return mid;
        else if (mv < x) l = mid + 1;
        else r = mid - 1;
    }
    return left ? l : r;
}

// bounds[2*b] = beg, bounds[2*b+1] = end; bounds[2*bins] = 1 when the stack must stay all-zero
__global__ void events_bounds(const double *__restrict__ ts, int64_t n, int bins, int64_t *__restrict__ bounds) {
    const int bi = blockIdx.x * blockDim.x + threadIdx.x;
    if (bi > bins) return;
    // ts.sum() == 0 for the sorted, non-negative normalised stamps the pipeline produces <=> both ends are 0
    const bool empty = (n <= 3) || (ts[0] == 0.0 && ts[n - 1] == 0.0);
    if (bi == bins) {
        bounds[2 * bins] = empty ? 1 : 0;
        return;
    }
    if (empty) {
        bounds[2 * bi] = 0;
        bounds[2 * bi + 1] = 0;
        return;
    }
    const double dt = __dadd_rn(__dsub_rn(ts[n - 1], ts[0]), 1e-6);
    const double delta_t = __ddiv_rn(dt, (double)bins);
    const double tstart = __dadd_rn(ts[0], __dmul_rn(delta_t, (double)bi));
    const double tend = __dadd_rn(tstart, delta_t);
    bounds[2 * bi] = ref_bsearch(ts, 0, n - 1, tstart, true);
    bounds[2 * bi + 1] = ref_bsearch(ts, 0, n - 1, tend, false) + 1;
}

__global__ void events_accumulate(const double *__restrict__ xs, const double *__restrict__ ys,
                                  const float *__restrict__ ps, int64_t n, int bins, int H, int W,
                                  const int64_t *__restrict__ bounds, float *__restrict__ out) {
    extern __shared__ int64_t sb[];
    for (int i = threadIdx.x; i < 2 * bins + 1; i += blockDim.x) sb[i] = bounds[i];
    __syncthreads();
    if (sb[2 * bins] != 0) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = xs[i], y = ys[i];
    const float p = ps[i];
    const float w = p * p;
    const bool oob = (x >= (double)W) || (x < 0.0) || (y >= (double)H) || (y < 0.0);
    const int64_t px = oob ? 0 : (int64_t)x, py = oob ? 0 : (int64_t)y;   // .long(): truncation
    const int64_t plane = (int64_t)H * W;
    bool zeroed = false;   // has an earlier bin already moved this event to (0,0)?
    for (int b = 0; b < bins; ++b) {
        if (i < sb[2 * b] || i >= sb[2 * b + 1]) continue;
        float *pos = out + (int64_t)b * plane + py * W + px;
        float *neg = pos + (int64_t)bins * plane;
        const bool pos_counts = !oob || zeroed;   // first bin: the positive pass masks it out
        if (p > 0.f && pos_counts) atomicAdd(pos, w);
        if (p < 0.f) atomicAdd(neg, w);           // negative pass always sees valid coordinates
        zeroed = oob;
    }
}

}  // namespace

extern "C" size_t ebfi_events_workspace(int bins) { return bins > 0 ? (size_t)(2 * bins + 1) * sizeof(int64_t) : 0; }

extern "C" int ebfi_events_to_stack(const double *xs, const double *ys, const double *ts, const float *ps, int64_t n,
                                    int bins, int H, int W, float *out, void *workspace, size_t workspace_bytes,
                                    void *stream) {
    if (bins <= 0 || H <= 0 || W <= 0 || n < 0) return fail(EBFI_ERR_ARG, "events_to_stack: bad sizes");
    if (!out) return fail(EBFI_ERR_ARG, "events_to_stack: null output");
    if (n > 0 && (!xs || !ys || !ts || !ps)) return fail(EBFI_ERR_ARG, "events_to_stack: null event array");
    if (bins > 2048) return fail(EBFI_ERR_UNSUPPORTED, "events_to_stack: more than 2048 bins");
    if (!workspace || workspace_bytes < ebfi_events_workspace(bins))
        return fail(EBFI_ERR_WORKSPACE, "events_to_stack: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(out, 0, (size_t)2 * bins * H * W * sizeof(float), st) != hipSuccess)
        return fail(EBFI_ERR_LAUNCH, "events_to_stack: memset failed");
    if (n <= 3) return EBFI_OK;
    int64_t *bounds = static_cast<int64_t *>(workspace);
    {
        ProfScope ps_("events_bounds", st);
        hipLaunchKernelGGL(events_bounds, dim3((unsigned)ceil_div(bins + 1, 64)), dim3(64), 0, st, ts, n, bins, bounds);
    }
    if (int rc = check_launch("events_bounds")) return rc;
    {
        ProfScope ps_("events_accumulate", st);
        hipLaunchKernelGGL(events_accumulate, dim3((unsigned)ceil_div(n, 256)), dim3(256),
                           (size_t)(2 * bins + 1) * sizeof(int64_t), st, xs, ys, ps, n, bins, H, W, bounds, out);
    }
    return check_launch("events_accumulate");
}
