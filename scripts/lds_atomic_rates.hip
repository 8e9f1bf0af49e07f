// Developer micro-benchmark: LDS atomic-add rates on gfx950 (what msda_bwd_accum's fixed-point accumulators rest on).
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics scripts/lds_atomic_rates.hip -o build_ab/lds_atomic_rates
// Eight 8-lane groups per wave, each adding to a random 32-element row of an LDS tile, 16 adds per lane and iteration.
// Measured (round 2, 16 waves per CU): ds_add_f32 193 cycles per wave instruction, ds_add_f64 16, ds_add_u32 4.6,
// ds_add_u64 7.1 when the lanes of a group add to consecutive words (16 with a 32-byte lane stride); a ds_add_u64 with
// 1/8 of its lanes enabled costs the same as a full one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T, int STRIDED, int ACTIVE>
__global__ __launch_bounds__(1024) void rate(const unsigned *__restrict__ rows, float *out, int iters, int nrows)
{
    extern __shared__ unsigned long long lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (unsigned i = threadIdx.x; i < (unsigned)nrows * 32; i += blockDim.x) lds[i] = T(0);
    __syncthreads();
    const unsigned g = lane >> 3, j = lane & 7;
    const unsigned *rp = rows + (blockIdx.x * 16 + wave) * 4096;
    for (int it = 0; it < iters; ++it) {
        unsigned r[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = rp[((it * 4 + c) * 8 + g) & 4095];
        if (g < ACTIVE) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    atomicAdd(&lds[r[c] * 32 + (STRIDED ? j * 4 + k : k * 8 + j)], T(it & 7));
        }
    }
    __syncthreads();
    double s = 0;
    for (unsigned i = threadIdx.x; i < (unsigned)nrows * 32; i += blockDim.x) s += (double)lds[i];
    if (s == 12345.678) out[0] = (float)s;
}

template <typename T, int STRIDED, int ACTIVE>
void run(const char *what, const unsigned *d, float *o, int threads)
{
    const int nrows = 256, iters = 300, blocks = 512;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        rate<T, STRIDED, ACTIVE><<<blocks, threads, nrows * 32 * sizeof(T)>>>(d, o, iters, nrows);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double wi = (double)blocks * (threads / 64) * iters * 16;
    printf("%-44s %4d threads: %7.2f CU-cycles per wave instruction (2.4 GHz, 256 CUs)\n", what, threads, best * 1e-3 * 2.4e9 * 256 / wi);
}

int main()
{
    std::vector<unsigned> h(512 * 16 * 4096);
    unsigned x = 12345;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (x >> 8) % 256; }
    unsigned *d;
    float *o;
    (void)hipMalloc(&d, h.size() * 4);
    (void)hipMalloc(&o, 4);
    (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int threads : {512, 1024}) {
        run<float, 0, 8>("ds_add_f32, consecutive lanes", d, o, threads);
        run<double, 0, 8>("ds_add_f64, consecutive lanes", d, o, threads);
        run<unsigned, 0, 8>("ds_add_u32, consecutive lanes", d, o, threads);
        run<unsigned long long, 0, 8>("ds_add_u64, consecutive lanes", d, o, threads);
        run<unsigned long long, 1, 8>("ds_add_u64, 32-byte lane stride", d, o, threads);
        run<unsigned long long, 0, 1>("ds_add_u64, 1 of 8 groups enabled", d, o, threads);
    }
    return 0;
}
