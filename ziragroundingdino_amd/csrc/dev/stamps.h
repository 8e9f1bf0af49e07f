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

// -DZIRA_DEV_BTIMES=1 (independent of the phase stamps above, which drain the memory queue at every stamp): one start / mark /
// end time per block and where it ran; the only disturbance is one wait + four stores at the end of a block.
#ifndef ZIRA_DEV_BTIMES
#define ZIRA_DEV_BTIMES 0
#endif
#if ZIRA_DEV_BTIMES
#include <hip/hip_runtime.h>
// absolute block times (100 MHz wall clock): start, a mark, end, kind | XCC id << 8 | HW_ID << 16 -- scripts/tile_timeline.py
__device__ unsigned long long zira_block_times[8 * 8192];
#define BTIME_DECL unsigned long long bt_0 = wall_clock64(), bt_1 = 0, bt_x = 0, bt_y = 0
#define BTIME_NOTE(x, y) do { bt_x += (x); bt_y += (y); } while (0)
#define BTIME_MARK bt_1 = wall_clock64()
#define BTIME_FLUSH(kind)                                                                                         \
    do {                                                                                                          \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                               \
        if (threadIdx.x == 0 && blockIdx.x < 8192) {                                                              \
            zira_block_times[blockIdx.x * 8 + 0] = bt_0;                                                          \
            zira_block_times[blockIdx.x * 8 + 1] = bt_1;                                                          \
            zira_block_times[blockIdx.x * 8 + 2] = wall_clock64();                                                \
            zira_block_times[blockIdx.x * 8 + 4] = bt_x;                                                          \
            zira_block_times[blockIdx.x * 8 + 5] = bt_y;                                                          \
            zira_block_times[blockIdx.x * 8 + 3] = (unsigned long long)(kind) |                                   \
                ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 8) |                            \
                ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 16);                            \
        }                                                                                                         \
    } while (0)
#define ZIRA_DEV_BTIME_READER                                                                                       \
    extern "C" int zira_dev_read_block_times(unsigned long long *host, int n)                                       \
    {                                                                                                               \
        return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(zira_block_times), sizeof(unsigned long long) * n);        \
    }
#else
#define BTIME_DECL
#define BTIME_MARK
#define BTIME_NOTE(x, y)
#define BTIME_FLUSH(kind)
#define ZIRA_DEV_BTIME_READER
#endif

#endif
