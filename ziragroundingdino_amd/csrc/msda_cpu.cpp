// msda_cpu.cpp -- host-memory twins of the two MSDA entry points (C ABI zira_msda_{fwd,bwd}_cpu_f32).
//
// The reference's CPU entry points are stubs that raise (csrc/MsDeformAttn/ms_deform_attn_cpu.cpp:17-41); its
// module falls back to a per-level grid_sample for CPU tensors (ms_deform_attn.py:326-348).  These twins give a
// binding the same op on host pointers -- the arithmetic of the device kernels' reference
// (ms_deform_im2col_cuda.cuh:237-299 forward, :87-159 inside :301-403 backward; bilinear corners :33-84), organised
// per sample: a `Tap` holds a sample's four corner rows and weights once, every channel reuses it.  Product code:
// it shares nothing with the test oracle (oracle/msda_oracle.c).  Heads are independent in all outputs, so the
// (batch, head) pairs are dealt to a few std::threads; sums run in fp32 in the order the reference's one-thread-per-
// channel kernel would produce them for a single sample (grad_value accumulates sample after sample, no atomics).
#include <stdint.h>

#include <cmath>
#include <thread>
#include <vector>

#include "zira_msda.h"

namespace {

struct Tap {
    bool live;           // inside (-1, H) x (-1, W): else the sample contributes nothing (cuh:288)
    const float *row[4]; // value rows of the corners (nullptr: outside the map, reads as 0)
    long off[4];         // the same as element offsets into value / grad_value, -1 outside the map
    float w[4];          // bilinear weights hh*hw, hh*lw, lh*hw, lh*lw (cuh:80-83)
    float lw, lh, hw, hh;
};

inline Tap make_tap(const float *vhead, long pix_stride, int H, int W, float x, float y)
{
    Tap t;
    const float h_im = y * (float)H - 0.5f, w_im = x * (float)W - 0.5f;   // (cuh:285-286)
    t.live = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
    for (int c = 0; c < 4; ++c) { t.row[c] = nullptr; t.off[c] = -1; t.w[c] = 0.f; }
    t.lw = t.lh = t.hw = t.hh = 0.f;
    if (!t.live) return t;
    const int hl = (int)std::floor(h_im), wl = (int)std::floor(w_im);
    t.lh = h_im - (float)hl;
    t.lw = w_im - (float)wl;
    t.hh = 1.f - t.lh;
    t.hw = 1.f - t.lw;
    const float ws[4] = {t.hh * t.hw, t.hh * t.lw, t.lh * t.hw, t.lh * t.lw};
    for (int c = 0; c < 4; ++c) {
        const int yy = hl + (c >> 1), xx = wl + (c & 1);
        t.w[c] = ws[c];
        if (yy >= 0 && yy <= H - 1 && xx >= 0 && xx <= W - 1) {
            t.off[c] = ((long)yy * W + xx) * pix_stride;
            t.row[c] = vhead + t.off[c];
        }
    }
    return t;
}

template <typename F>
void for_each_head(int B, int M, F f)
{
    const int heads = B * M;
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 1 : (nt > 16 ? 16 : nt);
    if ((int)nt > heads) nt = (unsigned)heads;
    if (nt <= 1) {
        for (int h = 0; h < heads; ++h) f(h / M, h % M);
        return;
    }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([=] { for (int h = (int)t; h < heads; h += (int)nt) f(h / M, h % M); });
    for (auto &th : pool) th.join();
}

bool dims_ok(int B, int S, int M, int D, int L, int Q, int P)
{
    return B > 0 && S > 0 && M > 0 && D > 0 && L > 0 && Q > 0 && P > 0;
}

}  // namespace

extern "C" {

int zira_msda_fwd_cpu_f32(const float *value, const int64_t *shapes, const int64_t *start, const float *loc,
                          const float *attn, int B, int S, int M, int D, int L, int Q, int P, float *out)
{
    if (!value || !shapes || !start || !loc || !attn || !out || !dims_ok(B, S, M, D, L, Q, P)) return ZIRA_MSDA_EINVAL;
    const long ps = (long)M * D;
    for_each_head(B, M, [&](int b, int m) {
        for (int q = 0; q < Q; ++q) {
            const long item = ((long)b * Q + q) * M + m;
            float *o = out + item * D;
            for (int c = 0; c < D; ++c) o[c] = 0.f;
            for (int l = 0; l < L; ++l) {
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
                const float *vhead = value + ((long)b * S + start[l]) * ps + (long)m * D;
                for (int p = 0; p < P; ++p) {
                    const long s = (item * L + l) * P + p;
                    const Tap t = make_tap(vhead, ps, H, W, loc[2 * s], loc[2 * s + 1]);
                    if (!t.live) continue;
                    const float a = attn[s];
                    for (int c = 0; c < D; ++c) {
                        float v = 0.f;   // (cuh:80-83: w1 v1 + w2 v2 + w3 v3 + w4 v4, corners outside the map read as 0)
                        for (int k = 0; k < 4; ++k) v += t.w[k] * (t.row[k] ? t.row[k][c] : 0.f);
                        o[c] += a * v;   // (cuh:290)
                    }
                }
            }
        }
    });
    return 0;
}

int zira_msda_bwd_cpu_f32(const float *grad_out, const float *value, const int64_t *shapes, const int64_t *start,
                          const float *loc, const float *attn, int B, int S, int M, int D, int L, int Q, int P,
                          float *grad_value, float *grad_loc, float *grad_attn)
{
    if (!grad_out || !value || !shapes || !start || !loc || !attn || !grad_value || !grad_loc || !grad_attn ||
        !dims_ok(B, S, M, D, L, Q, P))
        return ZIRA_MSDA_EINVAL;
    const long ps = (long)M * D;
    for_each_head(B, M, [&](int b, int m) {
        for (long px = 0; px < S; ++px) {   // this head's slice of grad_value starts at zero (ms_deform_attn_cuda.cu:122)
            float *g = grad_value + ((long)b * S + px) * ps + (long)m * D;
            for (int c = 0; c < D; ++c) g[c] = 0.f;
        }
        for (int q = 0; q < Q; ++q) {
            const long item = ((long)b * Q + q) * M + m;
            const float *go = grad_out + item * D;
            for (int l = 0; l < L; ++l) {
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
                const long lbase = ((long)b * S + start[l]) * ps + (long)m * D;
                for (int p = 0; p < P; ++p) {
                    const long s = (item * L + l) * P + p;
                    const Tap t = make_tap(value + lbase, ps, H, W, loc[2 * s], loc[2 * s + 1]);
                    float ga = 0.f, gx = 0.f, gy = 0.f;
                    if (t.live) {
                        const float a = attn[s];
                        for (int c = 0; c < D; ++c) {
                            const float tg = go[c] * a;   // top_grad * attn_weight (cuh:117)
                            float v[4];
                            for (int k = 0; k < 4; ++k) {
                                v[k] = t.row[k] ? t.row[k][c] : 0.f;
                                if (t.off[k] >= 0) grad_value[lbase + t.off[k] + c] += t.w[k] * tg;   // (cuh:118-153)
                            }
                            ga += go[c] * (t.w[0] * v[0] + t.w[1] * v[1] + t.w[2] * v[2] + t.w[3] * v[3]);   // (cuh:155-156)
                            const float dw = -t.hh * v[0] + t.hh * v[1] - t.lh * v[2] + t.lh * v[3];            // (cuh:123-151)
                            const float dh = -t.hw * v[0] - t.lw * v[1] + t.hw * v[2] + t.lw * v[3];
                            gx += (float)W * dw * tg;                                                            // (cuh:157-158)
                            gy += (float)H * dh * tg;
                        }
                    }
                    grad_attn[s] = ga;
                    grad_loc[2 * s] = gx;
                    grad_loc[2 * s + 1] = gy;
                }
            }
        }
    });
    return 0;
}

}  // extern "C"
