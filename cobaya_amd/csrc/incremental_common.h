// Shared device helpers of the incremental kernels (incremental_kernels.hip, incremental_any.hip): DPP quad permutes and sums, lane-mask helpers, the priority rotation,
// laundered LDS pointers, the size of a column chunk.  gfx950 only.
#pragma once
#include "det_math.h"
#include "kernels.h"

// Experiment hooks.  The shipped build uses the tuned values: MCMC_EXP_* are identities / no-ops.
// Timing experiments (occupancy sweeps, read-ahead depth, per-workgroup clocks) force-include
// _exp/inc_experiment.h (`-include`, tools/exp_inc_variants.sh), which defines them instead;
// nothing of that is compiled into libmcmc_hip.so.
#ifndef MCMC_EXP_WAVES
#define MCMC_EXP_WAVES(family, tuned) (tuned)    // waves per SIMD of a kernel family
#define MCMC_EXP_PIPE(tuned) (tuned)             // pairs fetched ahead in the trial loop
#define MCMC_EXP_BLOCK_BEGIN() ((void)0)         // per-workgroup clock and placement records
#define MCMC_EXP_BLOCK_END() ((void)0)
#define MCMC_EXP_ROTATE_SHIFT(tuned) (tuned)     // log2 of the shader clocks per priority turn
#define MCMC_EXP_ROTATE(on) (on)                 // rotate the wave priorities at all
#define MCMC_EXP_KEEP(tuned) (tuned)             // keep a step's (v, u) pairs in registers
#endif
#ifndef MCMC_EXP_FLOAT_BOUNDS
#define MCMC_EXP_FLOAT_BOUNDS(tuned) (tuned)     // single-precision copies of LDS-resident bounds in registers
#endif
#ifndef MCMC_EXP_BOUNDS_LDS
#define MCMC_EXP_BOUNDS_LDS(tuned) (tuned)       // per-dimension bounds in LDS (else in registers)
#endif

namespace mcmc {
namespace {

// ---------------------------------------------------------------- DPP quad helpers
template <int CTRL>
__device__ __forceinline__ double quad_perm(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// (p0 + p1) + (p2 + p3) in every lane of the quad; lane c holds p_c
__device__ __forceinline__ double quad_sum(double p)
{
    const double q = p + quad_perm<0xB1>(p);   // [1,0,3,2]: p0+p1 | p0+p1 | p2+p3 | p2+p3
    return q + quad_perm<0x4E>(q);             // [2,3,0,1]
}

// max over the four lanes of a quad (unsigned)
__device__ __forceinline__ unsigned quad_max_u32(unsigned v)
{
    unsigned o = (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);
    v = v > o ? v : o;
    o = (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);
    return v > o ? v : o;
}

// (lanes(cond) -- the wave's lane mask of a condition -- and the mask-taking selects sel(m, a, b)
// are in det_math.h)
// The issue arbiter of a SIMD serves its resident waves by priority, then by AGE.  Left alone, the
// oldest of the waves that share a SIMD for a whole launch finishes first and the youngest
// runs the last part of it alone, with nothing to cover its latencies (step kernel at d = 30:
// the workgroups of one launch end between 0.68 and 1.20 ms, tools/block_times.py).  The
// kernels therefore rotate their priority over the hardware wave slots -- the waves of a SIMD
// hold distinct slots -- so that they advance together: 1.22 -> 1.04 ms.  The turn is taken
// from the shader clock (a new level every 2^17 cycles, about 60 us), not from the wave's own
// progress: the waves of a SIMD then hold distinct levels at every moment however far apart
// they have drifted (with turns counted in steps the two oldest slots still finished 13 % early).
__device__ __forceinline__ int hw_wave_slot()
{
    return (int)(__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 15u);   // HW_ID.WAVE_ID
}
template <int NW>   // NW: the waves that share a SIMD (the kernel's occupancy)
__device__ __forceinline__ void rotate_priority(int slot)
{
    if (!MCMC_EXP_ROTATE(true)) return;
    // (a rotation over NW levels: over four levels two waves would not get equal turns; kernels
    // held to three waves are left alone -- measured: rotating them loses 2-8 %)
    if (NW != 2 && NW != 4) return;
    const int turn = (int)(__builtin_amdgcn_s_memtime() >> MCMC_EXP_ROTATE_SHIFT(17));
    switch (NW == 4 ? ((slot + turn) & 3) : ((slot + turn) & 1)) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// lane mask -> the mask of the lanes whose quad is held completely
__device__ __forceinline__ unsigned long long quad_all_mask(unsigned long long m)
{
    m &= m >> 1;
    m &= m >> 2;
    m &= 0x1111111111111111ull;
    return m * 15ull;
}
// ... and the same as a lane predicate
__device__ __forceinline__ bool quad_all(unsigned long long m)
{
    return __builtin_amdgcn_inverse_ballot_w64(quad_all_mask(m));
}

// A pointer into LDS that the optimiser has to take as new (so that it re-reads what it read
// before instead of keeping it in registers) and that stays an LDS pointer: the 32-bit LDS offset
// goes through the empty asm, not the generic pointer -- a laundered generic pointer makes every
// read a flat_load_dwordx4 (64-bit address, both memory counters).
typedef double __attribute__((ext_vector_type(2))) pair_t;   // (v_i, u_i): .x, .y
typedef const pair_t __attribute__((address_space(3))) * lds_pairs;
typedef const double __attribute__((address_space(3))) * lds_doubles;
// read-only data at wave-uniform addresses, read through the scalar cache
typedef const double __attribute__((address_space(4))) * cdoubles;
__device__ __forceinline__ unsigned lds_offset(const void* p) { return (unsigned)(unsigned long long)p; }
__device__ __forceinline__ lds_pairs relaunder(const double2* p)
{
    unsigned off = lds_offset(p);
    asm volatile("" : "+v"(off));
    return (lds_pairs)(unsigned long long)off;
}
__device__ __forceinline__ lds_pairs relaunder_after(lds_pairs p, double anchor)
{
    unsigned off = (unsigned)(unsigned long long)p;
    asm volatile("" : "+v"(off) : "v"(anchor));
    return (lds_pairs)(unsigned long long)off;
}
__device__ __forceinline__ lds_doubles relaunder(const double* p)
{
    unsigned off = lds_offset(p);
    asm volatile("" : "+v"(off));
    return (lds_doubles)(unsigned long long)off;
}

// The (r, Ea) variates of an OCTET of steps, staged in LDS (round 4).  Lane class c of a walker draws
// the Philox block of the step pair 4 * octet + c (PairRng) and leaves its two pairs in the wave's
// corner of a workgroup array sRE[kStagedPairs] = [wave][step of the octet][walker of the wave];
// a step then reads its pair with ONE ds_read_b128 -- the same 16 bytes in the four lanes of a
// walker.  (Until round 4 the pair came by a quad broadcast behind an eight-way switch on the step
// index: 4 DPP moves and ~12 scalar instructions per step, and 8 VGPRs alive across the octet;
// profiles/r04_instruction_diet.txt.)  A wave reads what it wrote itself: LDS operations of one
// wave complete in order, no barrier.
// Row stride: 17 pairs = 272 bytes (round 5).  With 16 (256 B) the four lane classes of a walker
// wrote rows 2c, 2c + 1 at the same 16 bytes of the 128-byte bank window of ds_write_b128 (served
// in groups of 8 contiguous lanes = two walkers x four classes, bank = (a / 4) mod 32): a 4-way
// conflict on every fill, 24 extra LDS cycles per store -- SQ_LDS_BANK_CONFLICT 1.73e7 -> 4.68e7
// per launch at d = 30 when the staging arrived.  With 272 the rows 2c start
// 32 c bytes apart modulo 128, so the eight lanes of a group cover the window exactly once.  The
// reads (one ds_read_b128 per step, the same 16 bytes in the four lanes of a walker, 16
// consecutive pairs per wave) are conflict-free with either stride.
#ifndef MCMC_STAGED_ROW
#define MCMC_STAGED_ROW 17
#endif
constexpr int kStagedRow = MCMC_STAGED_ROW;        // pairs per row (16 walkers + 1 of padding)
constexpr int kStagedPairs = 4 * 8 * kStagedRow;   // 8.5 KB per workgroup of four waves
struct StagedVariates {
    unsigned base, off;
    __device__ __forceinline__ void init(const pair_t* sRE, int wave, int lane)
    {
        base = lds_offset(sRE + (wave * 8 * kStagedRow + (lane >> 2)));
        off = base;
    }
    // after PairRng::run for the octet that holds step S (the step about to be taken)
    __device__ __forceinline__ void fill(pair_t* sRE, int wave, int lane, int c, const PairRng& pr,
                                         unsigned long long S)
    {
        pair_t* const mine = sRE + ((wave * 8 + 2 * c) * kStagedRow + (lane >> 2));
        mine[0] = pair_t{pr.r[0], pr.Ea[0]};
        mine[kStagedRow] = pair_t{pr.r[1], pr.Ea[1]};
        off = base + (unsigned)(S & 7ull) * (16u * kStagedRow);
    }
    __device__ __forceinline__ void fetch(double& r, double& Ea) const
    {
        const pair_t re = *(lds_pairs)(unsigned long long)off;
        r = re.x;
        Ea = re.y;
    }
    __device__ __forceinline__ void next() { off += 16u * kStagedRow; }
};

// ---------------------------------------------------------------- periodic parameters
// (step_inc_kernel<.., PER>; prior.py:658-676 in incremental mode, oracle: step_core_inc with
// `carry_periodic`)
constexpr int kMaxPeriodic = kIncMaxPeriodic;   // periodic parameters step_inc_kernel serves (kernels.h)

// the largest double below a finite x
__device__ __forceinline__ double pred_double(double x)
{
    const long long b = __double_as_longlong(x);
    if (x > 0.0) return __longlong_as_double(b - 1);
    if (x < 0.0) return __longlong_as_double(b + 1);
    return -4.9406564584124654e-324;
}

// (div_by -- the correctly rounded quotient from the correctly rounded reciprocal -- is in det_math.h)

// LDS behind the column chunks of a launch with np periodic parameters: the wrap moves
// [64 walkers][np] and the columns of L^-1 of the periodic dimensions [np][4 dq]
__host__ __device__ constexpr size_t inc_periodic_lds(int dq, int np)
{
    return sizeof(double) * (size_t)np * (64 + 4 * (size_t)dq);
}

// columns of one LDS chunk: a multiple of 4 (the variates come in fours); 14 KiB of pairs, 32 KiB
// from dq = 14 on (kernels of at most two waves per SIMD, i.e. two workgroups per CU: the
// workgroup barrier between chunks comes half as often)
// with_w: the chunk also holds the doubles of the carried log-prior's stream (24 bytes per
// dimension and column instead of 16: MODE 2 of step_inc_kernel)
// per: room for kMaxPeriodic periodic parameters (sPer, the wrap moves, the columns of L^-1)
__host__ __device__ constexpr int inc_chunk(int dq, bool with_w = false, bool per = false)
{
    if (per) {
        // four waves per SIMD (40 KB per workgroup) up to dq = 8, two (80 KB) above: what the
        // staged variates (8.5 KB), the logarithm table (2 KB), the bounds in LDS (64 dq bytes from
        // dq = 13 on), sPer (128 dq bytes) and inc_periodic_lds(dq, 8) leave, in bytes per
        // dimension and column: 16, or 24 with the carried log-prior's stream
        const int left = (dq <= 8 ? 40 : 80) * 1024 - 10752 - 64 * dq - 128 * dq -
                         (int)inc_periodic_lds(dq, kMaxPeriodic);
        int c = (left / (2 * (with_w ? 24 : 16)) / (4 * dq)) & ~3;
        return c < 4 ? 4 : (c > 64 ? 64 : c);
    }
    if (with_w) {
        int c = ((dq >= 9 ? 1364 : 600) / (4 * dq)) & ~3;
        return c < 8 ? 8 : (c > 64 ? 64 : c);
    }
    // (kernels at four waves per SIMD -- four workgroups per CU -- have 40 KB of LDS each: 28 KB
    // of pairs beside the 8.5 KB of staged variates and the 2 KB logarithm table)
    int c = ((dq >= 14 ? 2048 : 896) / (4 * dq)) & ~3;
    return c < 4 ? 4 : (c > 64 ? 64 : c);
}

}  // namespace
}  // namespace mcmc
