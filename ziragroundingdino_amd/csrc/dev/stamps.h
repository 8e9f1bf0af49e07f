// dev/stamps.h -- developer-only per-block phase timers for the sparse-backward kernels (scripts/tile_stamps.py).
// The shipped library is built WITHOUT -DZIRA_DEV_STAMPS: every macro below is then empty and nothing of this file
// reaches the binary.  A developer build (scripts/build_variant.sh stamps -DZIRA_DEV_STAMPS=1) gets two device arrays
// and two extern "C" readers; results stay correct, only the timing is disturbed (each stamp drains the memory queue).
#ifndef ZIRA_DEV_STAMPS_H_
#define ZIRA_DEV_STAMPS_H_

#ifndef ZIRA_DEV_STAMPS
#define ZIRA_DEV_STAMPS 0
#endif

#if ZIRA_DEV_STAMPS
#include <hip/hip_runtime.h>
__device__ unsigned long long zira_tile_stamps[16 * 2048];
__device__ unsigned long long zira_plan_stamps[16 * 2048];
#define TSTAMP_DECL unsigned long long ts_t = wall_clock64(), ts_acc[16] = {0}
#define TSTAMP(i)                                                   \
    do {                                                            \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        const unsigned long long ts_n = wall_clock64();             \
        ts_acc[i] += ts_n - ts_t;                                   \
        ts_t = ts_n;                                                \
    } while (0)
#define TSTAMP_COUNT(i) ts_acc[i] += 1
#define TSTAMP_FLUSH_TO(arr)                                                                     \
    do {                                                                                         \
        if (threadIdx.x == 0 && blockIdx.x < 2048)                                               \
            for (int ts_i = 0; ts_i < 16; ++ts_i) arr[blockIdx.x * 16 + ts_i] = ts_acc[ts_i];    \
    } while (0)
#define PSTAMP_FLUSH TSTAMP_FLUSH_TO(zira_plan_stamps)
#define TSTAMP_FLUSH TSTAMP_FLUSH_TO(zira_tile_stamps)
#define ZIRA_DEV_STAMP_READERS                                                                                      \
    extern "C" int zira_dev_read_tile_stamps(unsigned long long *host, int n)                                       \
    {                                                                                                               \
        return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(zira_tile_stamps), sizeof(unsigned long long) * n);        \
    }                                                                                                               \
    extern "C" int zira_dev_read_plan_stamps(unsigned long long *host, int n)                                       \
    {                                                                                                               \
        return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(zira_plan_stamps), sizeof(unsigned long long) * n);        \
    }
#else
#define TSTAMP_DECL
#define TSTAMP(i)
#define TSTAMP_COUNT(i)
#define TSTAMP_FLUSH
#define PSTAMP_FLUSH
#define ZIRA_DEV_STAMP_READERS
#endif

#endif
