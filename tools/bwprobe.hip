// How much read bandwidth does ONE workgroup per CU get, as a function of the waves that issue loads and the 16-byte loads each
// lane keeps in flight?  (The conv kernels stage through 4 producer waves per CU.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// PLANES = 0: a wave's load covers 1 KB contiguous.  PLANES = 1: the conv producers' pattern -- 8 lanes x 16 B = one 128-byte
// line in each of 8 planes that lie `plane` bytes apart.
// LB = bytes per lane and load: 16 (the quad stagers), 8 (round 5: the fp16-plane FAC kernels and the planar / piece loads of
// conv_wgrad_f16_tr*: 4 pixels x 2 bytes per lane, 512 contiguous bytes per wave instruction) or 4 (dword stagers).  The PMC
// calibration (tools/pmc_calibrate.py) reads FETCH_SIZE against the known byte count for each.
template <int NL, int LB>
__global__ __launch_bounds__(1024) void probe_narrow(const char *__restrict__ src, size_t bytes_per_wg, unsigned *__restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char *base = src + (size_t)blockIdx.x * bytes_per_wg;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, (unsigned)bytes_per_wg, 0x00020000);
    unsigned acc = 0;
    constexpr unsigned WB = 64u * LB;                     // bytes per wave instruction
    const unsigned step = (unsigned)nw * NL * WB;
    for (unsigned o = 0; o + step <= bytes_per_wg; o += step) {
        unsigned v[NL][LB / 4];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const unsigned off = o + (unsigned)(wave * NL + k) * WB + (unsigned)lane * LB;
            if constexpr (LB == 8) {
                const auto q = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
                v[k][0] = q[0]; v[k][1] = q[1];
            } else {
                v[k][0] = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < NL; ++k)
#pragma unroll
            for (int j = 0; j < LB / 4; ++j) acc += v[k][j];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int NL, int PLANES>
__global__ __launch_bounds__(1024) void probe(const char *__restrict__ src, size_t bytes_per_wg, unsigned plane, unsigned *__restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char *base = src + (size_t)blockIdx.x * bytes_per_wg;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, (unsigned)bytes_per_wg, 0x00020000);
    unsigned acc = 0;
    // the workgroup's slice is walked in steps of nw * NL KB
    const unsigned step = (unsigned)nw * NL * 1024u;
    for (unsigned o = 0; o + step <= bytes_per_wg; o += step) {
        u32x4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            unsigned off;
            if (PLANES) {
                // 8 planes inside the slice: plane p = lane & 7 at p * (bytes_per_wg / 8), 128 contiguous bytes per plane and load
                const unsigned within = (o / 8u) + (unsigned)(wave * NL + k) * 128u + (unsigned)(lane >> 3) * 16u;
                off = (unsigned)(lane & 7) * (unsigned)(bytes_per_wg / 8) + within;
            } else {
                off = o + (unsigned)(wave * NL + k) * 1024u + (unsigned)lane * 16u;
            }
            v[k] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int NL, int PLANES>
static void run(const char *src, size_t total, int nwaves, unsigned *sink) {
    const int G = 256;
    const size_t per = total / G;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<NL, PLANES>), dim3(G), dim3(64 * nwaves), 0, 0, src, per, 0u, sink);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe<NL, PLANES>), dim3(G), dim3(64 * nwaves), 0, 0, src, per, 0u, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("  %s  waves %2d  loads in flight/lane %2d : %7.1f us  %6.2f TB/s\n", PLANES ? "planes" : "linear", nwaves, NL, best * 1e3, total / (best * 1e-3) / 1e12);
}

template <int NL, int LB>
static void run_narrow(const char *src, size_t total, int nwaves, unsigned *sink) {
    const int G = 256;
    const size_t per = total / G;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe_narrow<NL, LB>), dim3(G), dim3(64 * nwaves), 0, 0, src, per, sink);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe_narrow<NL, LB>), dim3(G), dim3(64 * nwaves), 0, 0, src, per, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("  %2d-byte lanes  waves %2d  loads in flight/lane %2d : %7.1f us  %6.2f TB/s\n", LB, nwaves, NL, best * 1e3, total / (best * 1e-3) / 1e12);
}

int main(int argc, char **argv) {
    const size_t total = (argc > 1 ? (size_t)atoi(argv[1]) : 64) << 20;
    char *src; unsigned *sink;
    CK(hipMalloc(&src, total)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 1, total));
    printf("read %zu MB with 256 workgroups (one per CU)\n", total >> 20);
    for (int nw : {4, 8, 16}) {
        run<4, 0>(src, total, nw, sink);
        run<8, 0>(src, total, nw, sink);
        run<16, 0>(src, total, nw, sink);
        run<16, 1>(src, total, nw, sink);
        if (nw <= 8) run<32, 0>(src, total, nw, sink);
    }
    for (int nw : {4, 8}) {
        run_narrow<16, 8>(src, total, nw, sink);
        run_narrow<16, 4>(src, total, nw, sink);
    }
    return 0;
}
