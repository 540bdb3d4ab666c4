// kbench: standalone A/B harness for the conv kernels of csrc/conv2d.hip (no Python, no torch).
//
// Built by tools/build_kbench.sh into tools/bin/kbench (git-ignored, travels to the GPU box with the snapshot):
//     hipcc --offload-arch=gfx950 -O3 -DEBFI_KBENCH tools/kbench.hip ebfi-be_amd/csrc/runtime.hip -o tools/bin/kbench
// It #includes conv2d.hip so that the kernels in its anonymous namespace can be launched directly, times every variant
// with hipEvent pairs over interleaved rounds in ONE process (cdna_hip_programming.md rule 24), checks each variant's
// output against the exact-fp32 kernel of the library, and -- with -DEBFI_KBENCH -- reads the in-kernel s_memtime
// stamps (phase breakdown per workgroup).  Development tool only: nothing in the product path depends on it.
//
//   kbench fwd   [Cin Cout H W B]     forward variants (default 64 64 128 128 8)
//   kbench wgrad [Cin Cout H W B]     weight-gradient variants
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../ebfi-be_amd/csrc/conv2d.hip"

#define CK(x)                                                                                         \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) {                                                                       \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));         \
            exit(2);                                                                                  \
        }                                                                                             \
    } while (0)

static float *dev_random(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float u = (float)((s >> 8) & 0xffff) / 65536.f;
        s = s * 1664525u + 1013904223u;
        const float v = (float)((s >> 8) & 0xffff) / 65536.f;
        h[i] = scale * (u + v - 1.f) * 2.4494897f;   // ~unit variance
    }
    float *d;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

static double max_rel(const float *a_dev, const float *b_dev, size_t n) {
    std::vector<float> a(n), b(n);
    CK(hipMemcpy(a.data(), a_dev, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), b_dev, n * 4, hipMemcpyDeviceToHost));
    double md = 0, mb = 0;
    for (size_t i = 0; i < n; ++i) {
        md = std::max(md, (double)std::fabs(a[i] - b[i]));
        mb = std::max(mb, (double)std::fabs(b[i]));
    }
    return md / std::max(mb, 1e-30);
}

struct Variant {
    std::string name;
    std::function<int()> run;
    std::vector<float> us;
};

static void time_variants(std::vector<Variant> &vs, int rounds, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (auto &v : vs)
        for (int i = 0; i < 3; ++i)
            if (v.run()) { fprintf(stderr, "%s: launch failed: %s\n", v.name.c_str(), ebfi_last_error()); exit(3); }
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r)
        for (auto &v : vs) {
            CK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < reps; ++i) v.run();
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.us.push_back(1e3f * ms / reps);
        }
    for (auto &v : vs) {
        std::sort(v.us.begin(), v.us.end());
        printf("  %-34s median %8.2f us   min %8.2f us\n", v.name.c_str(), v.us[v.us.size() / 2], v.us[0]);
    }
}

#ifdef EBFI_KBENCH
static void report_stamps(unsigned long long *d_stamps, size_t nwg, const char *what) {
    std::vector<unsigned long long> h(nwg * 2 * KB_NSTAMP);
    CK(hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost));
    printf("  phase stamps of %s (cycles of s_memtime at 100 MHz, mean over %zu workgroups; wave 0 | wave 4; each wave's own stamps)\n", what, nwg);
    for (int i = 1; i < KB_NSTAMP; ++i) {
        double d[2] = {0, 0}, s[2] = {0, 0};
        size_t cnt[2] = {0, 0};
        for (size_t w = 0; w < nwg; ++w)
            for (int k = 0; k < 2; ++k) {
                const unsigned long long *row = &h[(w * 2 + k) * KB_NSTAMP];
                if (!row[i] || !row[0]) continue;
                ++cnt[k];
                int prev = i - 1;
                while (prev > 0 && !row[prev]) --prev;
                d[k] += (double)(row[i] - row[prev]);
                s[k] += (double)(row[i] - row[0]);
            }
        if (!cnt[0] && !cnt[1]) continue;
        printf("    stamp %2d:", i);
        for (int k = 0; k < 2; ++k) {
            if (cnt[k]) printf("  +%7.0f (since start %8.0f)", d[k] / cnt[k], s[k] / cnt[k]);
            else printf("  %8s %24s", "", "");
            if (k == 0) printf(" |");
        }
        printf("\n");
    }
}
#endif

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "fwd";
    const int Cin = argc > 2 ? atoi(argv[2]) : 64, Cout = argc > 3 ? atoi(argv[3]) : 64;
    const int H = argc > 4 ? atoi(argv[4]) : 128, W = argc > 5 ? atoi(argv[5]) : 128, B = argc > 6 ? atoi(argv[6]) : 8;
    const size_t nx = (size_t)B * Cin * H * W, ny = (size_t)B * Cout * H * W, nw = (size_t)Cout * Cin * 9;
    float *x = dev_random(nx, 1, 1.f), *w = dev_random(nw, 2, 1.f / std::sqrt((float)Cin * 9)), *bias = dev_random(Cout, 3, 0.1f);
    float *g = dev_random(ny, 4, 1.f);
    float *y_ref, *y, *gx, *gx_ref, *gw, *gw_ref, *gb;
    CK(hipMalloc(&y_ref, ny * 4));
    CK(hipMalloc(&y, ny * 4));
    CK(hipMalloc(&gx, nx * 4));
    CK(hipMalloc(&gx_ref, nx * 4));
    CK(hipMalloc(&gw, nw * 4));
    CK(hipMalloc(&gw_ref, nw * 4));
    CK(hipMalloc(&gb, Cout * 4));
    const size_t wsb = ebfi_conv2d_bf16_workspace(Cin, Cout, 3);
    void *ws;
    CK(hipMalloc(&ws, wsb));
    const size_t wgb = ebfi_conv2d_backward_weight_workspace(B, Cin, H, W, Cout, 3, 1, 1, EBFI_F32);
    void *wgs;
    CK(hipMalloc(&wgs, wgb));
    printf("kbench %s: Cin %d Cout %d %dx%d B %d   (algorithmic %.2f GFLOP, x3 = %.2f)\n", mode.c_str(), Cin, Cout, H, W, B,
           2e-9 * B * H * W * (double)Cin * Cout * 9, 6e-9 * B * H * W * (double)Cin * Cout * 9);
#ifdef EBFI_KBENCH
    unsigned long long *d_stamps = nullptr;
    const size_t nwg_max = 65536;
    CK(hipMalloc(&d_stamps, nwg_max * 2 * KB_NSTAMP * 8));
    CK(hipMemset(d_stamps, 0, nwg_max * 2 * KB_NSTAMP * 8));
    unsigned long long *null_stamps = nullptr;
#endif
    if (mode == "fwd") {
        if (ebfi_conv2d_forward(x, w, bias, y_ref, B, Cin, H, W, Cout, 3, 1, 1, 1, 0.01f, EBFI_F32, nullptr)) { fprintf(stderr, "%s\n", ebfi_last_error()); return 3; }
        std::vector<Variant> vs;
        vs.push_back({"fwd fp32 exact (library)", [&] { return ebfi_conv2d_forward(x, w, bias, y, B, Cin, H, W, Cout, 3, 1, 1, 1, 0.01f, EBFI_F32, nullptr); }, {}});
        auto with_vec = [&](const char *v, std::function<int()> f) { return [=] { setenv("EBFI_CONV_VEC", v, 1); int rc = f(); unsetenv("EBFI_CONV_VEC"); return rc; }; };
        auto fwd = [&] { return ebfi_conv2d_forward_bf16x3(x, w, bias, y, B, Cin, H, W, Cout, 3, 1, 1, 1, 0.01f, ws, wsb, nullptr); };
        auto dg_act = [&] { return ebfi_conv2d_backward_data_bf16x3(g, y_ref, w, gx, B, Cin, H, W, Cout, 3, 1, 1, 1, 0.01f, ws, wsb, nullptr); };
        auto dg = [&] { return ebfi_conv2d_backward_data_bf16x3(g, nullptr, w, gx, B, Cin, H, W, Cout, 3, 1, 1, 0, 0.f, ws, wsb, nullptr); };
        vs.push_back({"fwd x3 one tile per workgroup", [&] { setenv("EBFI_CONV_NOPERSIST", "1", 1); int rc = fwd(); unsetenv("EBFI_CONV_NOPERSIST"); return rc; }, {}});
        vs.push_back({"fwd x3 persistent over tiles", with_vec("1", fwd), {}});
        vs.push_back({"fwd x3 quad ld / dword st", with_vec("4", fwd), {}});
        vs.push_back({"dgrad x3 act' folded, dword", with_vec("1", dg_act), {}});
        vs.push_back({"dgrad x3 act' folded, quad", with_vec("4", dg_act), {}});
        vs.push_back({"dgrad x3 plain, quad", with_vec("4", dg), {}});
        time_variants(vs, 7, 20);
        ebfi_conv2d_forward_bf16x3(x, w, bias, y, B, Cin, H, W, Cout, 3, 1, 1, 1, 0.01f, ws, wsb, nullptr);
        CK(hipDeviceSynchronize());
        printf("  bf16x3 forward vs exact fp32: max rel %.3e\n", max_rel(y, y_ref, ny));
        ebfi_conv2d_backward_data(g, y_ref, w, gx_ref, B, Cin, H, W, Cout, 3, 1, 1, 1, 0.01f, EBFI_F32, nullptr);
        ebfi_conv2d_backward_data_bf16x3(g, y_ref, w, gx, B, Cin, H, W, Cout, 3, 1, 1, 1, 0.01f, ws, wsb, nullptr);
        CK(hipDeviceSynchronize());
        printf("  bf16x3 dgrad (act' folded) vs exact fp32: max rel %.3e\n", max_rel(gx, gx_ref, nx));
#ifdef EBFI_KBENCH
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_kb_stamps), &d_stamps, sizeof(d_stamps)));
        ebfi_conv2d_forward_bf16x3(x, w, bias, y, B, Cin, H, W, Cout, 3, 1, 1, 1, 0.01f, ws, wsb, nullptr);
        CK(hipDeviceSynchronize());
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_kb_stamps), &null_stamps, sizeof(null_stamps)));
        const size_t nwg = (size_t)B * ((H + 7) / 8) * ((W + 63) / 64) * ((Cout + 63) / 64);
        report_stamps(d_stamps, std::min(nwg, nwg_max), "conv_fwd_bf16x3_db forward");
#endif
    } else if (mode == "wgrad") {
        if (ebfi_conv2d_backward_weight(x, g, y_ref, gw_ref, gb, B, Cin, H, W, Cout, 3, 1, 1, 0, 0.f, wgs, wgb, EBFI_F32, nullptr)) { fprintf(stderr, "%s\n", ebfi_last_error()); return 3; }
        std::vector<Variant> vs;
        vs.push_back({"wgrad fp32 exact (library)", [&] { return ebfi_conv2d_backward_weight(x, g, y_ref, gw, gb, B, Cin, H, W, Cout, 3, 1, 1, 0, 0.f, wgs, wgb, EBFI_F32, nullptr); }, {}});
        vs.push_back({"wgrad x3 512 thr x 64 ch, 1 WG/CU", [&] { setenv("EBFI_WGRAD_BIGWG", "1", 1); int rc = ebfi_conv2d_backward_weight(x, g, y_ref, gw, gb, B, Cin, H, W, Cout, 3, 1, 1, 0, 0.f, wgs, wgb, EBFI_F32_BF16X3MMA, nullptr); unsetenv("EBFI_WGRAD_BIGWG"); return rc; }, {}});
        vs.push_back({"wgrad x3 256 thr x 32 ch, 2 WG/CU", [&] { return ebfi_conv2d_backward_weight(x, g, y_ref, gw, gb, B, Cin, H, W, Cout, 3, 1, 1, 0, 0.f, wgs, wgb, EBFI_F32_BF16X3MMA, nullptr); }, {}});
        vs.push_back({"wgrad bf16x3 act' folded + side out", [&] { return ebfi_conv2d_backward_weight_ex(x, g, y_ref, gw, gb, y, B, Cin, H, W, Cout, 3, 1, 1, 1, 0.01f, wgs, wgb, EBFI_F32_BF16X3MMA, nullptr); }, {}});
        time_variants(vs, 7, 20);
        ebfi_conv2d_backward_weight(x, g, y_ref, gw, gb, B, Cin, H, W, Cout, 3, 1, 1, 0, 0.f, wgs, wgb, EBFI_F32_BF16X3MMA, nullptr);
        CK(hipDeviceSynchronize());
        printf("  bf16x3 wgrad vs exact fp32: max rel %.3e\n", max_rel(gw, gw_ref, nw));
    } else if (mode == "f16") {
        // the fp16 backward kernels of the training step: data gradient (conv_fwd_f16_ws on a transposed image: here simply Cout ->
        // Cin channels with arbitrary weights, the timing does not depend on them) and the pixel-major weight gradient
        const int K16 = (Cout + 15) / 16 * 16;
        const size_t img_bytes = (size_t)9 * Cin * K16 * 2;
        std::vector<_Float16> himg(img_bytes / 2);
        for (size_t i = 0; i < himg.size(); ++i) himg[i] = (_Float16)(0.01f * (float)((int)(i * 2654435761u >> 24) - 128));
        void *img;
        CK(hipMalloc(&img, img_bytes));
        CK(hipMemcpy(img, himg.data(), img_bytes, hipMemcpyHostToDevice));
        // KBENCH_COLD=n: rotate over n independent operand sets (n x 100 MB at the default shape) so that no launch finds its
        // operands in the 256 MB Infinity Cache -- what the kernels see inside a training step
        const int ncold = getenv("KBENCH_COLD") ? std::max(1, atoi(getenv("KBENCH_COLD"))) : 1;
        std::vector<float *> xs{x}, gs{g}, gxs{gx};
        for (int i = 1; i < ncold; ++i) {
            xs.push_back(dev_random(nx, 10 + i, 1.f));
            gs.push_back(dev_random(ny, 40 + i, 1.f));
            float *t;
            CK(hipMalloc(&t, nx * 4));
            gxs.push_back(t);
        }
        int rot_d = 0, rot_w = 0;
        auto dg = [&] { const int i = rot_d++ % ncold; return ebfi_conv2d_packed_f16(gs[i], img, img_bytes, nullptr, gxs[i], B, Cout, H, W, Cin, 3, 1, 1, 0, 0.f, nullptr, nullptr, 0, 0.f, nullptr, nullptr, nullptr); };
        auto wg = [&] { const int i = rot_w++ % ncold; return ebfi_conv2d_backward_weight_f16g(xs[i], gs[i], nullptr, gw, gb, nullptr, B, Cin, H, W, Cout, 3, 1, 1, 0, 0.f, nullptr, nullptr, wgs, wgb, nullptr); };
        printf("  operand sets in rotation: %d\n", ncold);
        if (dg()) { fprintf(stderr, "%s\n", ebfi_last_error()); return 3; }
        if (wg()) { fprintf(stderr, "%s\n", ebfi_last_error()); return 3; }
        std::vector<Variant> vs;
        vs.push_back({"dgrad f16 (conv_fwd_f16_ws)", dg, {}});
        vs.push_back({"wgrad f16 tr + reduce", wg, {}});
        // the two are independent given grad_out: on two streams one kernel's ramp / tail can overlap the other's steady state
        hipStream_t s2;
        hipEvent_t ev_fork, ev_join;
        CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        CK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        auto wg_s2 = [&] { const int i = rot_w++ % ncold; return ebfi_conv2d_backward_weight_f16g(xs[i], gs[i], nullptr, gw, gb, nullptr, B, Cin, H, W, Cout, 3, 1, 1, 0, 0.f, nullptr, nullptr, wgs, wgb, s2); };
        vs.push_back({"dgrad + wgrad, one stream", [&] { int rc = dg(); return rc ? rc : wg(); }, {}});
        vs.push_back({"dgrad | wgrad, two streams", [&] {
            CK(hipEventRecord(ev_fork, nullptr));
            CK(hipStreamWaitEvent(s2, ev_fork, 0));
            int rc = wg_s2();
            if (rc) return rc;
            rc = dg();
            CK(hipEventRecord(ev_join, s2));
            CK(hipStreamWaitEvent(nullptr, ev_join, 0));
            return rc; }, {}});
        time_variants(vs, 7, 20);
#ifdef EBFI_KBENCH
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_kb_stamps), &d_stamps, sizeof(d_stamps)));
        dg();
        CK(hipDeviceSynchronize());
        report_stamps(d_stamps, std::min((size_t)256, nwg_max), "conv_fwd_f16_ws as data gradient (wave 0 = consumer | wave 4 = producer)");
        CK(hipMemset(d_stamps, 0, nwg_max * 2 * KB_NSTAMP * 8));
        wg();
        CK(hipDeviceSynchronize());
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_kb_stamps), &null_stamps, sizeof(null_stamps)));
        report_stamps(d_stamps, std::min((size_t)256, nwg_max), "conv_wgrad_f16_tr (wave 0 = consumer | wave 4 = producer)");
#endif
    } else if (mode == "f16img") {
        // the weight gradient on fp16 operand IMAGES (c16 layout; random halves: the timing does not depend on the values), with
        // the in-kernel stamps; KBENCH_COLD=n rotates over n operand sets
        const int ncold = getenv("KBENCH_COLD") ? std::max(1, atoi(getenv("KBENCH_COLD"))) : 1;
        auto dev_halves = [&](size_t n, unsigned seed) {
            std::vector<_Float16> h(n);
            unsigned s = seed * 2654435761u + 99u;
            for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (_Float16)(((float)((s >> 9) & 0xfff) / 2048.f - 1.f) * 2.f); }
            void *d;
            CK(hipMalloc(&d, n * 2));
            CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
            return d;
        };
        std::vector<void *> xs, gs;
        for (int i = 0; i < ncold; ++i) { xs.push_back(dev_halves(nx, 10 + i)); gs.push_back(dev_halves(ny, 40 + i)); }
        std::vector<float> hslot(128, 0.f);
        hslot[0] = 1.f; hslot[64] = 1.f;
        float *slots;
        CK(hipMalloc(&slots, 128 * 4));
        CK(hipMemcpy(slots, hslot.data(), 128 * 4, hipMemcpyHostToDevice));
        int rot = 0;
        auto wg = [&] { const int i = rot++ % ncold; return ebfi_conv2d_backward_weight_f16c(xs[i], gs[i], 0, gw, gb, B, Cin, H, W, Cout, 1, slots, slots + 64, wgs, wgb, nullptr); };
        printf("  operand sets in rotation: %d\n", ncold);
        if (wg()) { fprintf(stderr, "%s\n", ebfi_last_error()); return 3; }
        std::vector<Variant> vs;
        vs.push_back({"wgrad f16 images + reduce", wg, {}});
        time_variants(vs, 7, 20);
#ifdef EBFI_KBENCH
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_kb_stamps), &d_stamps, sizeof(d_stamps)));
        CK(hipMemset(d_stamps, 0, nwg_max * 2 * KB_NSTAMP * 8));
        wg();
        CK(hipDeviceSynchronize());
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_kb_stamps), &null_stamps, sizeof(null_stamps)));
        report_stamps(d_stamps, std::min((size_t)256, nwg_max), "conv_wgrad_f16_tr<IN16> (wave 0 = consumer | wave 4 = producer)");
#endif
    } else {
        fprintf(stderr, "usage: kbench fwd|wgrad|f16|f16img [Cin Cout H W B]\n");
        return 1;
    }
    return 0;
}
