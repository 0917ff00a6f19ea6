// Timing-experiment definitions of the MCMC_EXP_* hooks of incremental_kernels.hip.  NOT part of
// libmcmc_hip.so: tools/exp_inc_variants.sh force-includes this header (`-include`) into builds
// of cobaya_amd/csrc/_exp/lib_<name>.so, selected by -D flags:
//   -DEXP_STEP_WAVES=n / -DEXP_DRAG_WAVES=n / -DEXP_MIX_WAVES=n   waves per SIMD of a kernel family
//   -DEXP_PIPE=n              pairs fetched ahead in the trial loop of step_inc_kernel
//   -DEXP_KEEP=1              the step's (v, u) pairs stay in registers (no second LDS read)
//   -DEXP_NO_ROTATE           no wave-priority rotation;  -DEXP_ROTATE_SHIFT=k  its period
//   -DEXP_BOUNDS_REGS         per-dimension bounds in registers at every dq (else: inc_bounds_in_lds)
//   -DEXP_FLOAT_BOUNDS        single-precision copies of LDS-resident bounds in registers at every dq
//   -DEXP_BLOCK_TIMES         start / end clock and hardware placement of every workgroup
//                             (read back by tools/block_times.py)
#pragma once
#include <hip/hip_runtime.h>

#define MCMC_EXP_FAMILY_STEP 0
#define MCMC_EXP_FAMILY_DRAG 1
#define MCMC_EXP_FAMILY_MIX 2
#ifndef EXP_STEP_WAVES
#define EXP_STEP_WAVES 0
#endif
#ifndef EXP_DRAG_WAVES
#define EXP_DRAG_WAVES 0
#endif
#ifndef EXP_MIX_WAVES
#define EXP_MIX_WAVES 0
#endif
#define MCMC_EXP_WAVES(family, tuned)                                                     \
    ((MCMC_EXP_FAMILY_##family == 0 && EXP_STEP_WAVES) ? EXP_STEP_WAVES                    \
     : (MCMC_EXP_FAMILY_##family == 1 && EXP_DRAG_WAVES) ? EXP_DRAG_WAVES                  \
     : (MCMC_EXP_FAMILY_##family == 2 && EXP_MIX_WAVES) ? EXP_MIX_WAVES : (tuned))
#ifdef EXP_PIPE
#define MCMC_EXP_PIPE(tuned) (EXP_PIPE)
#else
#define MCMC_EXP_PIPE(tuned) (tuned)
#endif
#ifdef EXP_ROTATE_SHIFT
#define MCMC_EXP_ROTATE_SHIFT(tuned) (EXP_ROTATE_SHIFT)
#else
#define MCMC_EXP_ROTATE_SHIFT(tuned) (tuned)
#endif
#ifdef EXP_KEEP
#define MCMC_EXP_KEEP(tuned) (EXP_KEEP != 0)
#else
#define MCMC_EXP_KEEP(tuned) (tuned)
#endif
#ifdef EXP_BOUNDS_REGS
#define MCMC_EXP_BOUNDS_LDS(tuned) (false)
#endif
#ifdef EXP_FLOAT_BOUNDS
#define MCMC_EXP_FLOAT_BOUNDS(tuned) (true)
#endif
#ifdef EXP_NO_ROTATE
#define MCMC_EXP_ROTATE(on) (false)
#else
#define MCMC_EXP_ROTATE(on) (on)
#endif

#ifdef EXP_BLOCK_TIMES
__device__ unsigned long long g_block_times[2 * 4096];
__device__ unsigned int g_wave_place[2 * 4 * 4096];   // per wave: HW_ID, XCC_ID
extern "C" __attribute__((visibility("default"))) int mcmc_hip_debug_block_times(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_block_times), sizeof(g_block_times));
}
extern "C" __attribute__((visibility("default"))) int mcmc_hip_debug_wave_place(unsigned int* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_place), sizeof(g_wave_place));
}
#define MCMC_EXP_BLOCK_BEGIN()                                                                \
    do {                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_block_times[2 * blockIdx.x] = wall_clock64(); \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) {                                   \
            g_wave_place[2 * (4 * blockIdx.x + (threadIdx.x >> 6))] =                         \
                __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));                       \
            g_wave_place[2 * (4 * blockIdx.x + (threadIdx.x >> 6)) + 1] =                     \
                __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));                      \
        }                                                                                     \
    } while (0)
#define MCMC_EXP_BLOCK_END()                                                                  \
    do {                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x < 4096)                                            \
            g_block_times[2 * blockIdx.x + 1] = wall_clock64();                               \
    } while (0)
#else
#define MCMC_EXP_BLOCK_BEGIN() ((void)0)
#define MCMC_EXP_BLOCK_END() ((void)0)
#endif
//   -DEXP_FLOAT_LDS=0|1       (round 6) single-precision bounds from LDS + kept pairs in the two-wave kernels of MODE 1 / 2
#ifdef EXP_FLOAT_LDS
#define MCMC_EXP_FLOAT_LDS(tuned) (EXP_FLOAT_LDS != 0)
#endif
#ifdef EXP_FLOAT_LDS_KEEP
#define MCMC_EXP_FLOAT_LDS_KEEP(tuned) (EXP_FLOAT_LDS_KEEP != 0)
#endif
#ifdef EXP_FLOAT_VEC
#define MCMC_EXP_FLOAT_VEC(tuned) (EXP_FLOAT_VEC != 0)
#endif
//   -DEXP_DUO_DEPK=k          (round 6) incremental_duo.hip: the element of a plane whose result the next plane's reads wait for
//   -DEXP_DUO_KEEPV=1         ... the v plane of a step kept in registers from the trial to the commit
#ifdef EXP_DUO_DEPK
#define MCMC_DUO_DEPK(tuned) (EXP_DUO_DEPK)
#endif
#ifdef EXP_DUO_KEEPV
#define MCMC_DUO_KEEPV(tuned) (EXP_DUO_KEEPV != 0)
#endif
